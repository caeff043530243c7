#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Runs only in the build container, where /root/reference exists (it never travels to the GPU
box; the .npz files written here do).  The reference is imported unmodified with two stub
modules for packages that are absent here and unused on this path:
``wandb`` (mtm_model.py:31) and ``gym`` (learner.py:4, a type annotation).

What is captured, and how:
  * the reference's ``Learner`` methods are called unmodified on an instance built with
    ``object.__new__`` (its __init__ needs an env + checkpoint file);
  * their local variables (expect_return, p, sample_actions, decode, ...) are read from the
    returning frame via ``sys.setprofile`` -- no reference source is edited or copied;
  * the candidate noise is made explicit by replacing ``torch.normal`` for the duration of
    the call with ``eps*std+mean`` on a stored ``eps`` -- verified below to be bit-identical
    to what ``torch.normal`` itself returns for the same generator state.

Weights/statistics/histories come from ``m3pc_amd.synth`` (deterministic recipes), so only
inputs that are not recipe-derived and the expected outputs are stored.

Usage:  python tests/golden/make_golden.py [g1 g2 g3 g4 ...]
"""
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
for _name in ("wandb", "gym"):
    sys.modules.setdefault(_name, types.ModuleType(_name))
sys.modules["gym"].Env = object
sys.path.insert(0, "/root/reference")

from research.omtm.models.mtm_model import omtmConfig  # noqa: E402
from research.omtm.tokenizers.base import TokenizerManager  # noqa: E402
from research.omtm.tokenizers.continuous import ContinuousTokenizer  # noqa: E402
from research.omtm.datasets.base import DataStatistics  # noqa: E402
from research.finetune_omtm import masks as ft_masks  # noqa: E402
from research.finetune_omtm.model import TwinQ  # noqa: E402
from research.finetune_omtm.learner import Learner  # noqa: E402
from research.zeroshot_omtm import masks as zs_masks  # noqa: E402
from research.zeroshot_omtm.learner import Learner as ZSLearner  # noqa: E402

from m3pc_amd import synth  # noqa: E402

torch.set_num_threads(8)
META = dict(torch_version=torch.__version__, reference="wkh923/m3pc @ /root/reference (2025-03-21)",
            reference_torch_pin="pytorch==1.12.1 (README.md:24)")


# ----------------------------------------------------------------------------------------------
class FrameTap:
    """Collect f_locals of selected reference functions when they return."""

    def __init__(self, names):
        self.names = set(names)
        self.locals = {}

    def __call__(self, frame, event, arg):
        if event == "return" and frame.f_code.co_name in self.names and "learner.py" in frame.f_code.co_filename:
            self.locals[frame.f_code.co_name] = dict(frame.f_locals)

    def __enter__(self):
        sys.setprofile(self)
        return self

    def __exit__(self, *a):
        sys.setprofile(None)


class ExplicitNormal:
    """torch.normal(mean, std) -> eps*std+mean for a provided eps (see module docstring)."""

    def __init__(self, eps):
        self.eps = eps

    def __enter__(self):
        self.orig = torch.normal
        eps = self.eps

        def fake(mean, std, *a, **k):
            assert mean.shape == eps.shape, (mean.shape, eps.shape)
            return eps * std + mean

        torch.normal = fake

    def __exit__(self, *a):
        torch.normal = self.orig


def check_normal_identity():
    g = torch.Generator().manual_seed(5)
    loc = torch.randn(1, 8, 1, 3, generator=g)
    std = torch.rand(1, 8, 1, 3, generator=g) + 0.1
    shape = (16, 1, 8, 1, 3)
    torch.manual_seed(9)
    a = torch.normal(loc.expand(shape), std.expand(shape))
    torch.manual_seed(9)
    e = torch.randn(shape)
    assert torch.equal(a, e * std + loc), "torch.normal != randn*std+loc on this build"


def build_reference(dims: synth.Dims, cfg_kw, seed=0, zeroshot=False):
    sd = synth.make_state_dict(dims, seed)
    mc = omtmConfig(norm="none", n_embd=dims.n_embd, n_enc_layer=dims.n_enc_layer, n_dec_layer=dims.n_dec_layer,
                    n_head=dims.n_head, dropout=0.1)
    model = mc.create(dims.data_shapes, dims.traj_length, {k: False for k in synth.KEYS}).eval()
    missing = model.load_state_dict(sd, strict=True)
    stats = synth.make_tokenizer_stats(dims, seed)
    toks = {}
    for k in synth.KEYS:  # config.yaml:16-24 order: states, actions, returns, rewards (order is irrelevant)
        s = stats[k]
        toks[k] = ContinuousTokenizer(s["mean"], s["std"], DataStatistics(s["mean"], s["std"], s["min"], s["max"]),
                                      normalize=(k != "actions"))
    tm = TokenizerManager(toks)
    qsd, om, os_ = synth.make_critic(dims, seed)
    qf = TwinQ(dims.state_dim, dims.action_dim, om, os_).eval()
    qf.load_state_dict(qsd, strict=True)
    L = object.__new__(ZSLearner if zeroshot else Learner)
    L.cfg = types.SimpleNamespace(traj_length=dims.traj_length, device="cpu", **cfg_kw)
    L.tokenizer_manager = tm
    L.mtm = model
    L.iql = types.SimpleNamespace(qf=qf)
    return L


def hist_with_len(dims, seed, path_length):
    h = synth.make_history(dims, seed)
    h["path_length"] = path_length
    return h


def hook_io(module, store, name):
    def fn(mod, inp, out):
        store.setdefault(name + "_in", []).append(inp[0].detach().clone())
        store.setdefault(name + "_out", []).append(out.detach().clone())
    return module.register_forward_hook(fn)


def npf(x):
    return np.ascontiguousarray(x.detach().cpu().numpy()) if torch.is_tensor(x) else np.asarray(x)


def run_guiding(L, dims, mode, path_length, rtg, eps=None, seed_sel=77, noise_seed=None, taps=None, hist_seed=0):
    """Call action_sample -> <mode> on the reference and return the captured locals."""
    hist = hist_with_len(dims, hist_seed, path_length)
    fname = {"rtg": "rtg_guiding", "critic": "critic_lambda_guiding", "noise": "noise_adding_lambda"}[mode]
    L.cfg.plan_guidance = fname
    handles = []
    if taps is not None:
        handles = [hook_io(L.mtm.encoder, taps, "enc"), hook_io(L.mtm.decoder, taps, "dec")]
    with FrameTap([fname, "action_sample"]) as tap:
        if mode == "noise":
            torch.manual_seed(noise_seed)
            out = L.action_sample(hist, plan=True, eval=True, rtg=rtg)
        else:
            torch.manual_seed(seed_sel)
            with ExplicitNormal(eps):
                out = L.action_sample(hist, plan=True, eval=True, rtg=rtg)
    for h in handles:
        h.remove()
    loc = tap.locals[fname]
    loc["_window"] = tap.locals["action_sample"]["torch_zero_trajectory"]
    loc["_horizon"] = tap.locals["action_sample"]["horizon"]
    loc["_eval_out"] = out
    return loc


# ----------------------------------------------------------------------------------------------
def g1():
    """Tiny config, every tensor stored."""
    dims = synth.Dims(11, 3, 8, n_embd=64, n_head=2)
    N, H = 16, 4
    out = dict(meta=str(META), dims=np.array([11, 3, 8, 64, 2, N, H]))
    eps = synth.make_eps(N, dims, seed=1)
    out["eps"] = npf(eps)
    for mode, temp in (("rtg", 0.01), ("critic", 1.0), ("noise", 1.0)):
        L = build_reference(dims, dict(action_samples=N, horizon=H, discount=0.99, temperature=temp, lmbda=0.6,
                                       plan_guidance=""))
        for pl in (0, 3, 100, 998):
            taps = {}
            loc = run_guiding(L, dims, mode, pl, rtg=3.0, eps=eps, noise_seed=123, taps=taps)
            pre = f"{mode}_pl{pl}_"
            for k, v in loc["_window"].items():
                out[pre + "win_" + k] = npf(v)
            out[pre + "horizon"] = np.array(loc["_horizon"])
            ad = loc["action_dist"]
            out[pre + "loc"] = npf(ad.loc)
            out[pre + "std"] = npf(ad.std)
            out[pre + "sample_actions"] = npf(loc["sample_actions"])
            for k in ("states", "rewards", "returns"):
                out[pre + "dec_" + k] = npf(loc["decode"][k])
            out[pre + "expect_return"] = npf(loc["expect_return"])  # after the max shift
            out[pre + "p"] = npf(loc["p"])
            out[pre + "eval_action"] = npf(loc["eval_action"])
            out[pre + "sample_idx"] = npf(loc["sample_idx"])
            out[pre + "sample_action"] = npf(loc["sample_action"])
            if pl == 100:
                for nm in ("enc_in", "enc_out", "dec_in", "dec_out"):
                    out[pre + nm + "_pass1"] = npf(taps[nm][0])
                    out[pre + nm + "_pass2"] = npf(taps[nm][1])
    # no-plan path (mtm_sampling, learner.py:103-115)
    L = build_reference(dims, dict(action_samples=N, horizon=H, discount=0.99, temperature=1.0, lmbda=0.6,
                                   plan_guidance="rtg_guiding"))
    for pl in (0, 100):
        hist = hist_with_len(dims, 0, pl)
        eps1 = synth.make_eps(1, dims, seed=3)[0]
        with FrameTap(["mtm_sampling"]) as tap, ExplicitNormal(eps1):
            ev = L.action_sample(hist, plan=False, eval=True, rtg=3.0)
        out[f"noplan_pl{pl}_eval_action"] = npf(ev)
        out[f"noplan_pl{pl}_sample_action"] = npf(tap.locals["mtm_sampling"]["sample_action"])
        out[f"noplan_pl{pl}_eps"] = npf(eps1)
    # explore rtg (rtg=None branch, learner.py:375-385)
    hist = hist_with_len(dims, 0, 50)
    with FrameTap(["action_sample"]) as tap, ExplicitNormal(synth.make_eps(1, dims, seed=3)[0]):
        L.action_sample(hist, percentage=0.8, plan=False, eval=False, rtg=None)
    out["explore_returns"] = npf(tap.locals["action_sample"]["torch_zero_trajectory"]["returns"])
    np.savez_compressed(os.path.join(HERE, "g1_tiny.npz"), **out)
    print("g1 written", len(out), "arrays")


FULL = {
    # name: (env, mode, N, H, T, temperature, candidate blocks to run (None = all at once))
    "c1": ("hopper", "rtg", 64, 8, 16, 0.01, None),
    "c2": ("hopper", "rtg", 1024, 16, 32, 0.01, None),
    "c2s": ("hopper", "rtg", 1024, 16, 16, 0.01, None),
    "c3": ("walker2d", "critic", 4096, 16, 32, 1.0, None),
    "c4": ("halfcheetah", "rtg", 16384, 32, 64, 0.01, [(0, 512), (8192, 8704), (15872, 16384)]),
}


def g2(which=None):
    """Full-size configs (d=512): recipe weights, outputs only."""
    for name, (env, mode, N, H, T, temp, blocks) in FULL.items():
        if which and name not in which:
            continue
        t0 = time.time()
        S, A = synth.ENV_DIMS[env]
        dims = synth.Dims(S, A, T)
        eps = synth.make_eps(N, dims, seed=1)
        out = dict(meta=str(META), cfg=np.array([S, A, T, H, N]), mode=mode, temperature=temp)
        if blocks is None:
            L = build_reference(dims, dict(action_samples=N, horizon=H, discount=0.99, temperature=temp, lmbda=0.6,
                                           plan_guidance=""))
            loc = run_guiding(L, dims, mode, 500, rtg=3.0, eps=eps)
            er = loc["expect_return"]
            out["expect_return_shifted"] = npf(er)
            out["argmax"] = np.array(int(torch.argmax(er)))
            out["top32"] = npf(torch.topk(er, min(32, N)).indices)
            out["p"] = npf(loc["p"])
            out["eval_action"] = npf(loc["eval_action"])
            out["sample_idx"] = npf(loc["sample_idx"])
            out["sample_action"] = npf(loc["sample_action"])
            out["loc"] = npf(loc["action_dist"].loc)
            out["std"] = npf(loc["action_dist"].std)
            rows = np.linspace(0, N - 1, 8).astype(int)
            out["rows"] = rows
            out["sample_actions_rows"] = npf(loc["sample_actions"][rows])
            for k in ("rewards", "returns") + (("states",) if mode == "critic" else ()):
                out["dec_" + k + "_rows"] = npf(loc["decode"][k][rows][:, T - H:])
        else:
            # candidate blocks of the big config: each block is an independent reference call with
            # cfg.action_samples = block size and the matching slice of the full eps tensor.
            ers = []
            for (b0, b1) in blocks:
                L = build_reference(dims, dict(action_samples=b1 - b0, horizon=H, discount=0.99, temperature=temp,
                                               lmbda=0.6, plan_guidance=""))
                loc = run_guiding(L, dims, mode, 500, rtg=3.0, eps=eps[b0:b1])
                # the reference shifts expect_return in place by the block's own max (learner.py:318);
                # stored as such, next to the decoded rewards/returns the scores were built from.
                dec = loc["decode"]
                ers.append((b0, b1, loc["expect_return"].clone(), dec["rewards"][:, T - H:, 0].clone(),
                            dec["returns"][:, T - H:, 0].clone()))
                out["loc"] = npf(loc["action_dist"].loc)
                out["std"] = npf(loc["action_dist"].std)
            out["blocks"] = np.array([(b0, b1) for b0, b1, *_ in ers])
            out["expect_return_shifted_blocks"] = np.stack([npf(e) for _, _, e, _, _ in ers])
            out["dec_rewards_blocks"] = np.stack([npf(r) for *_, r, _ in ers])
            out["dec_returns_blocks"] = np.stack([npf(g) for *_, g in ers])
        np.savez_compressed(os.path.join(HERE, f"g2_{name}.npz"), **out)
        print(f"g2 {name} written in {time.time() - t0:.1f}s")


G5_CASES = [(env, mode, wseed, hseed, pl) for env, mode in (("hopper", "rtg"), ("walker2d", "critic"))
            for wseed in (0, 1, 2) for hseed, pl in ((10, 200), (11, 237), (12, 31), (13, 998))]


def g5():
    """Arg-max pins for the bf16 + fp32-re-score path: 2 envs x 3 weight seeds x 4 windows at
    N=256, H=16, T=32, full-size model, plus one exact tie (two candidates with the same noise)."""
    N, H, T = 256, 16, 32
    out = dict(meta=str(META), cases=np.array([f"{e}:{m}:{w}:{hs}:{pl}" for e, m, w, hs, pl in G5_CASES]), cfg=np.array([N, H, T]))
    t0 = time.time()
    for ci, (env, mode, wseed, hseed, pl) in enumerate(G5_CASES):
        S, A = synth.ENV_DIMS[env]
        dims = synth.Dims(S, A, T)
        temp = 0.01 if mode == "rtg" else 1.0
        L = build_reference(dims, dict(action_samples=N, horizon=H, discount=0.99, temperature=temp, lmbda=0.6, plan_guidance=""),
                            seed=wseed)
        eps = synth.make_eps(N, dims, seed=100 + ci)
        loc = run_guiding(L, dims, mode, pl, rtg=3.0, eps=eps, hist_seed=hseed)
        er = loc["expect_return"]
        out[f"er_{ci}"] = npf(er)
        out[f"argmax_{ci}"] = np.array(int(torch.argmax(er)))
        out[f"eval_action_{ci}"] = npf(loc["eval_action"])
        out[f"horizon_{ci}"] = np.array(int(loc["_horizon"]))
        if ci == 0:  # the tie: copy the winner's noise onto a later candidate -> two equal maxima, torch.argmax takes the first
            am = int(torch.argmax(er))
            j = (am + 57) % N
            eps2 = eps.clone()
            eps2[j] = eps2[am]
            loc2 = run_guiding(L, dims, mode, pl, rtg=3.0, eps=eps2, hist_seed=hseed)
            er2 = loc2["expect_return"]
            assert float(er2[j]) == float(er2[am]) == float(er2.max())
            out["tie_pair"] = np.array([am, j])
            out["tie_er"] = npf(er2)
            out["tie_argmax"] = np.array(int(torch.argmax(er2)))
            out["tie_eval_action"] = npf(loc2["eval_action"])
        print(f"g5 case {ci} {env} w{wseed} h{hseed} pl{pl}: argmax {int(out[f'argmax_{ci}'])} ({time.time() - t0:.0f}s)", flush=True)
    np.savez_compressed(os.path.join(HERE, "g5_argmax.npz"), **out)
    print("g5 written")


def g3():
    """Zero-shot goal reaching, two-pass piid + single-pass id, E=4 windows."""
    dims = synth.Dims(11, 3, 8)
    wp = np.loadtxt("/root/reference/research/zeroshot_omtm/waypoint_gen/hopper-wiggle-f2.txt")
    assert wp.shape == (1000, 11)
    raw = wp.astype(np.float32).copy()  # the reference's way-point data file as read (a data fixture for the hold test)
    # index_jump hold (zeroshot learner.py:530-539, config_hopper.yaml index_jump: 4)
    jump, father = 4, 4
    while father < 999:
        for i in range(jump):
            wp[father - 1 - i] = wp[father]
        father += jump + 1
    out = dict(meta=str(META), waypoints_held=wp.astype(np.float32), waypoints_raw=raw)
    L = build_reference(dims, dict(action_samples=1, horizon=4, discount=0.99, temperature=1.0, lmbda=0.6,
                                   plan_guidance="", index_jump=4), zeroshot=True)
    base = synth.make_history(dims, 0)
    pls = [0, 2, 37, 997]
    out["path_lengths"] = np.array(pls)
    for pl in pls:
        hist = dict(base)
        obs = wp.astype(np.float32).copy()
        obs[: pl + 1] = base["observations"][: pl + 1]  # "observed" states up to the current step
        hist["observations"] = obs
        hist["path_length"] = pl
        for fn in ("action_piid_sample", "action_id_sample"):
            with FrameTap([fn]) as tap:
                ev = getattr(L, fn)(hist, percentage=1.0, plan=False, eval=True, rtg=2.5)
            loc = tap.locals[fn]
            pre = f"{fn}_pl{pl}_"
            out[pre + "eval_action"] = npf(ev)
            out[pre + "loc"] = npf(loc["action_dist"].loc)
            out[pre + "std"] = npf(loc["action_dist"].std)
            out[pre + "horizon"] = np.array(loc["horizon"])
            if fn == "action_piid_sample":
                out[pre + "state_inference"] = npf(loc["state_inference"])
                out[pre + "win_states_after"] = npf(loc["torch_zero_trajectory"]["states"])
            else:
                out[pre + "win_states"] = npf(loc["torch_zero_trajectory"]["states"])
        # goal_mask "piid_allout" (unseen.py:146-148): action_piid_list_sample leaves its result in learner.action_list and
        # returns None (zeroshot learner.py:263-370); shot pops it (559-568).  Always the distribution's mean, eval or not.
        L.action_list = []
        assert L.action_piid_list_sample(hist, percentage=1.0, plan=False, eval=True, rtg=2.5) is None
        assert len(L.action_list) == 1
        out[f"action_piid_list_sample_pl{pl}_action0"] = npf(L.action_list[0])
        L.action_list = []
        L.action_piid_list_sample(hist, percentage=0.7, plan=False, eval=False, rtg=None)  # return-to-go from the statistics
        out[f"action_piid_list_sample_pl{pl}_explore_action0"] = npf(L.action_list[0])
        out[f"obs_pl{pl}"] = obs
    np.savez_compressed(os.path.join(HERE, "g3_zeroshot.npz"), **out)
    print("g3 written")


def g4():
    """Mask known-answer tests: rows = states, actions, rewards, returns."""
    out = dict(meta=str(META))
    for (T, idx) in ((8, 4), (16, 0), (32, 16), (8, 0), (8, 7)):
        for nm, fn in (("rcbc", ft_masks.create_rcbc_mask), ("fd", ft_masks.create_fd_mask),
                       ("pi", zs_masks.create_pi_mask), ("fid", zs_masks.create_fid_mask),
                       ("gid", zs_masks.create_gid_mask)):
            m = fn(T, "cpu", idx)
            out[f"{nm}_T{T}_i{idx}"] = np.stack([m[k].numpy() for k in synth.KEYS]).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "g4_masks.npz"), **out)
    print("g4 written")


if __name__ == "__main__":
    check_normal_identity()
    todo = sys.argv[1:] or ["g1", "g4", "g3", "g2", "g5"]
    for t in todo:
        if t == "g1":
            g1()
        elif t == "g3":
            g3()
        elif t == "g4":
            g4()
        elif t == "g5":
            g5()
        elif t == "g2":
            g2()
        elif t.startswith("g2:"):
            g2(t[3:].split(","))

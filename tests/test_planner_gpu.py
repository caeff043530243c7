"""GPU tests of the host-side mirror of the reference plug-in API (m3pc_amd.planner / mtm / tokenizers):
same call signatures, shapes and outputs as the reference Learner methods, checked against the golden
vectors captured from the reference and against the oracle."""
import os
import types

import numpy as np
import pytest
import torch

from m3pc_amd import capi, synth
from m3pc_amd.masks import create_fd_mask, create_rcbc_mask
from m3pc_amd.mtm import omtmConfig
from m3pc_amd.planner import HipPlanner, attach
from m3pc_amd.tokenizers import ContinuousTokenizer, DataStatistics, TokenizerManager
from oracle import mtm_oracle as O

pytestmark = pytest.mark.gpu
GD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cfg(T, N, H, temp, guidance):
    return types.SimpleNamespace(traj_length=T, action_samples=N, horizon=H, discount=0.99, temperature=temp, lmbda=0.6,
                                 plan_guidance=guidance, device="cuda")


def _planner(dims, cfg, precision="fp32", gen=None):
    qsd, om, os_ = synth.make_critic(dims, 0)
    return HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), qsd, om, os_,
                      n_embd=dims.n_embd, n_head=dims.n_head, precision=precision, generator=gen)


@pytest.mark.parametrize("guidance,mode,temp", [("rtg_guiding", "rtg", 0.01), ("critic_lambda_guiding", "critic", 1.0)])
@pytest.mark.parametrize("pl", [0, 3, 100, 998])
def test_action_sample_matches_reference_golden(guidance, mode, temp, pl):
    """action_sample(history, eval=True, rtg=3.0) end to end (window assembly included) on the tiny config,
    with the candidate noise injected so that the run is comparable to the golden capture."""
    g = np.load(os.path.join(GD, "g1_tiny.npz"))
    dims = synth.Dims(11, 3, 8, n_embd=64, n_head=2)
    p = _planner(dims, _cfg(8, 16, 4, temp, guidance))
    eps = torch.from_numpy(g["eps"]).cuda()
    p._eps = lambda shape: eps  # the reference's draw, made explicit (tests/golden/make_golden.py)
    hist = synth.make_history(dims, 0)
    hist["path_length"] = pl
    ev = p.action_sample(hist, plan=True, eval=True, rtg=3.0)
    pre = f"{mode}_pl{pl}_"
    assert ev.shape == (3,)
    assert np.abs(ev.cpu().numpy() - g[pre + "eval_action"]).max() < 2e-5
    assert int(p.last["argmax"].item()) == int(np.argmax(g[pre + "expect_return"]))
    got = (p.last["expect_return"] - p.last["expect_return"].max()).cpu().numpy()
    assert np.abs(got - g[pre + "expect_return"]).max() <= 2e-5 * max(1.0, float(np.abs(g[pre + "expect_return"]).max()))
    # sample branch: (1, A) row of the candidate set, chosen by multinomial over p
    sa = p.action_sample(hist, plan=True, eval=False, rtg=3.0)
    assert sa.shape == (1, 3)
    acts0 = p.last["sample_actions"][:, 0]
    assert (acts0 == sa).all(dim=1).any()
    p.handle.close()


@pytest.mark.parametrize("pl", [0, 100])
def test_noplan_path_matches_reference_golden(pl):
    g = np.load(os.path.join(GD, "g1_tiny.npz"))
    dims = synth.Dims(11, 3, 8, n_embd=64, n_head=2)
    p = _planner(dims, _cfg(8, 16, 4, 1.0, "rtg_guiding"))
    eps1 = torch.from_numpy(g[f"noplan_pl{pl}_eps"]).cuda()
    p._eps = lambda shape: eps1
    hist = synth.make_history(dims, 0)
    hist["path_length"] = pl
    ev = p.action_sample(hist, plan=False, eval=True, rtg=3.0)
    sa = p.action_sample(hist, plan=False, eval=False, rtg=3.0)
    assert np.abs(ev.cpu().numpy() - g[f"noplan_pl{pl}_eval_action"]).max() < 2e-5
    assert np.abs(sa.cpu().numpy() - g[f"noplan_pl{pl}_sample_action"]).max() < 2e-5
    p.handle.close()


def test_explore_rtg_and_window_assembly():
    dims = synth.Dims(11, 3, 8, n_embd=64, n_head=2)
    p = _planner(dims, _cfg(8, 16, 4, 1.0, "rtg_guiding"))
    hist = synth.make_history(dims, 0)
    ocfg = O.PlanCfg(8, 4, 16, n_head=2)
    for pl in (0, 3, 100, 998):
        hist["path_length"] = pl
        s, a, r, h, rtg = p.assemble_window(hist, rtg=None, percentage=0.8)
        win, hh = O.assemble_window(ocfg, hist, pl, rtg)
        assert h == hh and rtg == O.explore_rtg(O.make_stats(synth.make_tokenizer_stats(dims, 0)), 0.8)
        assert torch.equal(s.cpu(), win["states"][0]) and torch.equal(a.cpu(), win["actions"][0])
        assert torch.equal(r.cpu(), win["rewards"][0])
    p.handle.close()


def test_c2_bf16_with_fp32_rescore_keeps_reference_argmax():
    """BASELINE config 2 through the planner: bf16 candidate pass, fp32 re-score of the top-32:
    arg-max identical to the reference; top candidates' scores fp32-accurate."""
    g = np.load(os.path.join(GD, "g2_c2.npz"))
    dims = synth.Dims(11, 3, 32)
    p = _planner(dims, _cfg(32, 1024, 16, 0.01, "rtg_guiding"), precision="bf16")
    eps = synth.make_eps(1024, dims, 1).cuda()
    p._eps = lambda shape: eps
    hist = synth.make_history(dims, 0)
    hist["path_length"] = 500
    ev = p.action_sample(hist, plan=True, eval=True, rtg=3.0)
    assert int(p.last["argmax"].item()) == int(g["argmax"])
    er = p.last["expect_return"]
    top = p.last["topk"].cpu().numpy()  # the re-scored candidates (all within 2 delta of the bf16 maximum)
    assert int(g["argmax"]) in top and 4 <= top.size <= 64
    got = (er - er.max()).cpu().numpy()[top]
    ref = g["expect_return_shifted"][top]
    assert np.abs(got - ref).max() <= 5e-5 * float(er.abs().max())
    assert np.abs(ev.cpu().numpy() - g["eval_action"]).max() < 5e-3
    # The sampling branch (learner.py:324-325) in bf16 mode: every weight is within the deviation band of the reference's (the
    # un-re-scored candidates' scores carry the bf16 deviation); the INDEX drawn is certified separately (tests/test_race_gpu.py:
    # the reference's stored-seed index is reproduced).
    pb, pr = p.last["p"].cpu().double(), torch.from_numpy(g["p"].reshape(-1)).double()
    band = float(np.exp(0.01 * 4 * p.last["delta"]) - 1.0)
    assert float((pb / pr - 1).abs().max()) <= band, (float((pb / pr - 1).abs().max()), band)
    p.handle.close()


def test_c2_candidate_halves_on_two_streams_are_bit_identical_to_one_stream(monkeypatch):
    """The default splits the 1024 candidates of BASELINE config 2 into two halves on two HIP streams (fused-tail tiles: one
    round per half instead of two, DESIGN.md 4 "overlap"); M3PC_TWO_STREAM=0 (read when the handle is created) runs one chain.
    Candidates are independent and every kernel's result for a row does not depend on the row count, so all scores are equal
    bit for bit -- also with the fused kernels switched off (the GEMM + LayerNorm chains)."""
    dims = synth.Dims(11, 3, 32)
    eps = synth.make_eps(1024, dims, 1).cuda()
    hist = synth.make_history(dims, 0)
    hist["path_length"] = 500

    from hip_util import lab_library
    monkeypatch.setattr(capi, "_lib", lab_library())  # the environment A/B switches exist in the lab build only

    def scores(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        p = _planner(dims, _cfg(32, 1024, 16, 0.01, "rtg_guiding"), precision="bf16")
        for k in env:
            monkeypatch.delenv(k)
        p._eps = lambda shape: eps
        s, a, r, h, rtg = p.assemble_window(hist, rtg=3.0)
        res = p.handle.plan_step(capi.MODE_RTG, s, a, r, eps.reshape(1024, -1, 3), h, rtg, 0.6, 0.99, 1024, 0, 1024,
                                 precision=capi.PREC_BF16)
        out = res["expect_return"].clone(), res["sample_actions"].clone()
        torch.cuda.synchronize()
        p.handle.close()
        return out

    er2, sa2 = scores({})
    er1, sa1 = scores({"M3PC_TWO_STREAM": "0"})
    assert torch.equal(er1, er2) and torch.equal(sa1, sa2)
    # the round-3 fusions against the launches they replaced (the next layer's Q|K|V inside the layer tail, the output heads
    # inside the decoder tail): the same candidates, scores within a fraction of the bf16 deviation of the fp32 scores
    scale = float(er2.abs().max())
    for env in ({"M3PC_NO_QKV_FUSED": "1"}, {"M3PC_NO_HEAD_FUSED": "1"}, {"M3PC_NO_QKV_FUSED": "1", "M3PC_NO_HEAD_FUSED": "1"}):
        er_, sa_ = scores(env)
        assert torch.equal(sa_, sa2)
        d = er_ - er2
        assert float((d - d.median()).abs().max()) <= 1e-2 * scale, (env, float((d - d.median()).abs().max()), scale)


@pytest.mark.parametrize("pl", [0, 2, 37, 997])
def test_zeroshot_goal_reaching_matches_reference_golden(pl):
    """action_piid_sample (two chained forwards) and action_id_sample on windows whose future states are the
    reference's way-points (G3 fixture captured from research/zeroshot_omtm/learner.py)."""
    g = np.load(os.path.join(GD, "g3_zeroshot.npz"))
    dims = synth.Dims(11, 3, 8)
    cfg = types.SimpleNamespace(traj_length=8, action_samples=1, horizon=4, discount=0.99, temperature=1.0, lmbda=0.6,
                                plan_guidance="rtg_guiding", index_jump=4)
    p = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None)
    hist = synth.make_history(dims, 0)
    hist["observations"] = g[f"obs_pl{pl}"]
    hist["path_length"] = pl
    ev = p.action_piid_sample(hist, percentage=1.0, plan=False, eval=True, rtg=2.5)
    assert ev.shape == g[f"action_piid_sample_pl{pl}_eval_action"].shape == (1, 3)  # mean[0, T-h] of a (1,T,1,A) dist
    assert np.abs(ev.cpu().numpy() - g[f"action_piid_sample_pl{pl}_eval_action"]).max() < 2e-5
    si = g[f"action_piid_sample_pl{pl}_state_inference"]
    assert np.abs(p.last["state_inference"].cpu().numpy() - si).max() <= 2e-5 * max(1.0, float(np.abs(si).max()))
    assert np.abs(p.last["window_states"].cpu().numpy() - g[f"action_piid_sample_pl{pl}_win_states_after"][0]).max() <= 1e-4
    ev2 = p.action_id_sample(hist, percentage=1.0, plan=False, eval=True, rtg=2.5)
    assert np.abs(ev2.cpu().numpy() - g[f"action_id_sample_pl{pl}_eval_action"]).max() < 2e-5
    sa = p.action_id_sample(hist, percentage=1.0, plan=False, eval=False, rtg=2.5)
    assert sa.shape == (1, 3) and float(sa.abs().max()) <= 1.0
    p.handle.close()


def test_zeroshot_batched_windows_match_reference_goldens():
    """BASELINE config 5 shape of the path: E goal-reaching windows per launch (here the four G3 windows, which
    have two different horizons) -- each row must equal the reference's single-window result."""
    g = np.load(os.path.join(GD, "g3_zeroshot.npz"))
    dims = synth.Dims(11, 3, 8)
    cfg = types.SimpleNamespace(traj_length=8, action_samples=1, horizon=4, discount=0.99, temperature=1.0, lmbda=0.6,
                                plan_guidance="rtg_guiding", index_jump=4)
    p = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, max_batch=8)
    pls = [0, 2, 37, 997, 37, 2]
    hists = []
    for pl in pls:
        hist = synth.make_history(dims, 0)
        hist["observations"] = g[f"obs_pl{pl}"]
        hist["path_length"] = pl
        hists.append(hist)
    assert len({int(g[f"action_piid_sample_pl{pl}_horizon"]) for pl in pls}) > 1  # mixed horizons in one call
    ev = p.action_piid_sample_batch(hists, percentage=1.0, eval=True, rtg=2.5)
    assert ev.shape == (len(pls), 3)
    for i, pl in enumerate(pls):
        assert np.abs(ev[i].cpu().numpy() - g[f"action_piid_sample_pl{pl}_eval_action"][0]).max() < 2e-5
        si = g[f"action_piid_sample_pl{pl}_state_inference"][0]
        assert np.abs(p.last["state_inference"][i].cpu().numpy() - si).max() <= 2e-5 * max(1.0, float(np.abs(si).max()))
    sa = p.action_piid_sample_batch(hists, percentage=1.0, eval=False, rtg=2.5)
    assert sa.shape == (len(pls), 3) and float(sa.abs().max()) <= 1.0
    p.handle.close()


def test_planner_from_reference_style_checkpoints(tmp_path):
    """`{"model": state_dict}` / `{"qf": ...}` files (train.py:1208-1216, model.py:310-320) give the
    same planner as handing the tensors over directly."""
    from m3pc_amd import checkpoint
    dims = synth.Dims(11, 3, 8, n_embd=64, n_head=2)
    sd = synth.make_state_dict(dims, 0)
    st = synth.make_tokenizer_stats(dims, 0)
    qsd, om, os_ = synth.make_critic(dims, 0)
    torch.save({"model": sd, "optimizer": {}, "step": 3, "eval_max": {}}, tmp_path / "m.pt")
    torch.save({"qf": qsd, "vf": {}, "actor": {}, "total_it": 1}, tmp_path / "iql_3.pt")
    cfg = types.SimpleNamespace(traj_length=8, action_samples=16, horizon=4, discount=0.99, temperature=1.0, lmbda=0.6,
                                plan_guidance="critic_lambda_guiding")
    a = checkpoint.planner_from_checkpoints(cfg, str(tmp_path / "m.pt"), st, str(tmp_path / "iql_3.pt"), om, os_, n_head=2)
    b = HipPlanner(cfg, sd, st, qsd, om, os_, n_embd=64, n_head=2)
    hist = synth.make_history(dims, 0)
    hist["path_length"] = 500
    for pl in (a, b):
        pl.generator = torch.Generator(device="cuda").manual_seed(5)
    ea = a.action_sample(hist, plan=True, eval=True, rtg=3.0)
    eb = b.action_sample(hist, plan=True, eval=True, rtg=3.0)
    assert torch.equal(ea, eb)
    assert torch.equal(a.last["expect_return"], b.last["expect_return"])
    a.handle.close()
    b.handle.close()


def test_omtm_and_tokenizer_mirror_compose_like_the_reference():
    """tokenizer_manager.decode(mtm(tokenizer_manager.encode(traj), mask)) -- the reference's idiom
    (learner.py:108-111) -- on the mirror classes."""
    dims = synth.Dims(11, 3, 8, n_embd=64, n_head=2)
    model = omtmConfig(n_embd=64, n_head=2, n_enc_layer=2, n_dec_layer=1, norm="none").create(
        dims.data_shapes, 8, {k: False for k in synth.KEYS}, max_batch=4)
    sd = synth.make_state_dict(dims, 0)
    model.load_state_dict(sd)
    st = synth.make_tokenizer_stats(dims, 0)
    tm = TokenizerManager({k: ContinuousTokenizer(st[k]["mean"], st[k]["std"],
                                                  DataStatistics(st[k]["mean"], st[k]["std"], st[k]["min"], st[k]["max"]),
                                                  normalize=(k != "actions")) for k in synth.KEYS}).bind(model.handle)
    g = torch.Generator().manual_seed(0)
    traj = {"states": torch.randn(2, 8, 11, generator=g), "actions": torch.rand(2, 8, 3, generator=g) * 2 - 1,
            "rewards": torch.randn(2, 8, 1, generator=g), "returns": torch.from_numpy(3.0 * np.ones((2, 8, 1)))}
    ostats = O.make_stats(st)
    for mk, mo in ((create_rcbc_mask, O.rcbc_mask), (create_fd_mask, O.fd_mask)):
        masks = mk(8, "cuda", 4)
        out = tm.decode(model(tm.encode({k: v.cuda() for k, v in traj.items()}), masks))
        ref = O.mtm_forward(sd, O.encode_all(traj, ostats), mo(8, 4), 2)
        assert list(out.keys()) == list(traj.keys())
        for k in ("states", "rewards", "returns"):
            r = O.tok_decode(ref[k], ostats[k])
            assert out[k].shape == r.shape
            assert (out[k].cpu() - r).abs().max() <= 2e-5 * max(1.0, float(r.abs().max()))
        assert (out["actions"].loc.cpu() - ref["actions"][0]).abs().max() < 2e-5
        assert (out["actions"].mean.cpu() - torch.tanh(ref["actions"][0])).abs().max() < 2e-5
    model.handle.close()


def test_attach_rebinds_a_learner_like_object_and_tracks_weight_updates():
    """attach() on an object shaped like the reference Learner (mtm nn.Module-like with parameters,
    tokenizer_manager.tokenizers[k]._data_mean/..., iql.qf)."""
    dims = synth.Dims(11, 3, 8, n_embd=64, n_head=2)
    sd = synth.make_state_dict(dims, 0)

    class FakeModule(torch.nn.Module):
        def __init__(self, sd_):
            super().__init__()
            self.names = list(sd_.keys())
            self.ps = torch.nn.ParameterList([torch.nn.Parameter(v.clone(), requires_grad=False) for v in sd_.values()])

        def state_dict(self, *a, **k):
            return {n: p.data for n, p in zip(self.names, self.ps)}

    mtm = FakeModule(sd)
    mtm.config = omtmConfig(n_embd=64, n_head=2, n_enc_layer=2, n_dec_layer=1)
    st = synth.make_tokenizer_stats(dims, 0)
    toks = {k: types.SimpleNamespace(_data_mean=torch.tensor(st[k]["mean"]), _data_std=torch.tensor(st[k]["std"]),
                                     normalize=(k != "actions"),
                                     stats=DataStatistics(st[k]["mean"], st[k]["std"], st[k]["min"], st[k]["max"]))
            for k in synth.KEYS}
    qsd, om, os_ = synth.make_critic(dims, 0)
    qf = FakeModule(qsd)
    qf.obs_mean, qf.obs_std = om, os_
    learner = types.SimpleNamespace(cfg=_cfg(8, 16, 4, 0.01, "rtg_guiding"), mtm=mtm,
                                    tokenizer_manager=types.SimpleNamespace(tokenizers=toks), iql=types.SimpleNamespace(qf=qf))
    import inspect
    assert inspect.signature(attach).parameters["precision"].default == "bf16"  # (round 6: the drop-in's default is the headline's)
    planner = attach(learner, precision="fp32")  # (this test holds the result to the reference's fp32 golden at 2e-5)
    eps = synth.make_eps(16, dims, 1).cuda()
    planner._eps = lambda shape: eps
    hist = synth.make_history(dims, 0)
    hist["path_length"] = 100
    g = np.load(os.path.join(GD, "g1_tiny.npz"))
    ev = learner.action_sample(hist, plan=True, eval=True, rtg=3.0)
    assert np.abs(ev.cpu().numpy() - g["rtg_pl100_eval_action"]).max() < 2e-5
    er0 = planner.last["expect_return"].clone()
    # a fine-tuning step mutates the parameters in place -> the next call must see the new weights
    with torch.no_grad():
        i = mtm.names.index("output_head_dict.returns.3.bias")
        mtm.ps[i].add_(0.25)
    learner.action_sample(hist, plan=True, eval=True, rtg=3.0)
    shift = (planner.last["expect_return"].cpu().numpy() - planner.last["expect_return"].max().item())
    # a constant added to every predicted return shifts all scores equally: shifted scores unchanged,
    # raw scores moved
    assert np.abs(shift - g["rtg_pl100_expect_return"]).max() <= 1e-3
    assert float((planner.last["expect_return"] - er0).abs().min()) > 1.0, "updated weights were not picked up"
    planner.handle.close()

"""The bf16 plan step with the bound-driven fp32 re-score (m3pc_amd.planner, rescore="bound") against arg-max pins
captured from the real reference (tests/golden/make_golden.py g5: learner.py:271-327 / 211-268 on 2 envs x 3 weight
seeds x 4 windows, N=256, H=16, T=32, full-size model) -- VERDICT r1 item 4.

What must hold: the reported arg-max IS the reference's fp32 arg-max (bit-exact index), the reference's arg-max is
among the re-scored candidates, the window statistics are reported, and an exact tie resolves to the lower index as
torch.argmax does."""
import os
import types

import numpy as np
import pytest
import torch

from m3pc_amd import capi, synth
from m3pc_amd.planner import HipPlanner

pytestmark = pytest.mark.gpu
GD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _planner(dims, N, H, temp, guidance, wseed, **kw):
    cfg = types.SimpleNamespace(traj_length=dims.traj_length, action_samples=N, horizon=H, discount=0.99, temperature=temp,
                                lmbda=0.6, plan_guidance=guidance, device="cuda")
    qsd, om, os_ = synth.make_critic(dims, wseed)
    return HipPlanner(cfg, synth.make_state_dict(dims, wseed), synth.make_tokenizer_stats(dims, wseed), qsd, om, os_,
                      precision="bf16", **kw)


def _cases():
    g = np.load(os.path.join(GD, "g5_argmax.npz"))
    return [str(c) for c in g["cases"]]


@pytest.fixture(scope="module")
def g5():
    return np.load(os.path.join(GD, "g5_argmax.npz"))


def _run(p, dims, hseed, pl, eps):
    p._eps = lambda shape: eps
    hist = synth.make_history(dims, hseed)
    hist["path_length"] = pl
    return p.action_sample(hist, plan=True, eval=True, rtg=3.0)


def test_bound_rescore_keeps_the_reference_argmax(g5):
    N, H, T = (int(v) for v in g5["cfg"])
    planners = {}
    worst_n = 0
    for ci, case in enumerate(_cases()):
        env, mode, wseed, hseed, pl = case.split(":")
        wseed, hseed, pl = int(wseed), int(hseed), int(pl)
        S, A = synth.ENV_DIMS[env]
        dims = synth.Dims(S, A, T)
        key = (env, wseed)
        if key not in planners:
            planners[key] = _planner(dims, N, H, 0.01 if mode == "rtg" else 1.0,
                                     "rtg_guiding" if mode == "rtg" else "critic_lambda_guiding", wseed)
        p = planners[key]
        eps = synth.make_eps(N, dims, 100 + ci).cuda()
        ev = _run(p, dims, hseed, pl, eps)
        ref_er = g5[f"er_{ci}"]
        ref_am = int(g5[f"argmax_{ci}"])
        last = p.last
        assert int(last["argmax"].item()) == ref_am, (case, int(last["argmax"].item()), ref_am)
        top = last["topk"].cpu().numpy()
        assert ref_am in top, (case, "reference arg-max was not re-scored")
        assert 4 <= last["n_rescored"] + last["n_race"] and last["n_rescored"] == top.size
        assert last["delta"] > 0 and last["min_margin_outside"] >= 0
        # every candidate inside the window was re-scored: either the listed prefix covered the window (containment held
        # with the bound's own room) or the whole window set went through the chunked fallback
        assert last["n_rescored"] >= last["n_in_window"]  # (n_in_window: what the first certificate asked for)
        if not last["saturated"]:
            assert last["n_rescored"] <= 64
            assert last["min_margin_outside"] >= 0  # the certificate's threshold clears the best un-re-scored bf16 score
        assert last["deviation"] <= last["delta"]
        # the re-scored scores are fp32-accurate (shifted like the reference's, learner.py:318)
        er = last["expect_return"]
        got = (er - er.max()).cpu().numpy()[top]
        assert np.abs(got - ref_er[top]).max() <= 1e-4 * max(1.0, float(np.abs(ref_er).max())), case
        # softmax-weighted action: every candidate contributes, the un-re-scored ones at bf16 accuracy
        assert np.abs(ev.cpu().numpy() - g5[f"eval_action_{ci}"]).max() < 2e-2, case
        worst_n = max(worst_n, last["n_rescored"])
    print("largest re-score set", worst_n)
    for p in planners.values():
        p.handle.close()


def test_exact_tie_goes_to_the_lower_index(g5):
    N, H, T = (int(v) for v in g5["cfg"])
    env, mode, wseed, hseed, pl = _cases()[0].split(":")
    S, A = synth.ENV_DIMS[env]
    dims = synth.Dims(S, A, T)
    a, b = (int(v) for v in g5["tie_pair"])
    eps = synth.make_eps(N, dims, 100).clone()
    eps[b] = eps[a]
    for kw in (dict(), dict(rescore="topk", rescore_topk=16)):
        p = _planner(dims, N, H, 0.01, "rtg_guiding", int(wseed), **kw)
        _run(p, dims, int(hseed), int(pl), eps.cuda())
        er = p.last["expect_return"]
        assert float(er[a]) == float(er[b]) == float(er.max())  # same noise -> same candidate -> same bits, both re-scored
        assert int(p.last["argmax"].item()) == int(g5["tie_argmax"]) == min(a, b)
        p.handle.close()


def test_fixed_delta_and_topk_modes():
    """rescore_delta pins the window; rescore='topk' is the round-1 fixed-k behaviour; fp32 planners do not re-score."""
    dims = synth.Dims(11, 3, 32)
    eps = synth.make_eps(256, dims, 3).cuda()
    p = _planner(dims, 256, 16, 0.01, "rtg_guiding", 0, rescore_delta=0.0, rescore_min=5)
    _run(p, dims, 0, 400, eps)
    # delta 0: the floor -- race entries included -- plus whatever beats the fp32 best outright
    assert 3 <= p.last["n_rescored"] <= 16 and p.last["n_first"] + p.last["n_race_first"] == 5
    p.handle.close()
    p = _planner(dims, 256, 16, 0.01, "rtg_guiding", 0, rescore="topk", rescore_topk=7)
    _run(p, dims, 0, 400, eps)
    assert p.last["topk"].numel() == 7 and "n_rescored" not in p.last
    p.handle.close()


@pytest.mark.parametrize("delta,rmax", [(1e9, 32), (40.0, 8)])
def test_window_larger_than_the_cap_falls_back_to_the_whole_window_set(delta, rmax):
    """More candidates inside the 2 delta window than rescore_max may list (VERDICT r2 weak 2 / ADVICE: the bound is lost
    silently): the planner re-scores the WHOLE window set in chunks, warns once, reports it, and the arg-max is the fp32 one.
    delta = 1e9: everything is inside (256 > 32); delta = 40: a few dozen are (> 8)."""
    dims = synth.Dims(11, 3, 32)
    eps = synth.make_eps(256, dims, 3).cuda()
    p32 = HipPlanner(types.SimpleNamespace(traj_length=32, action_samples=256, horizon=16, discount=0.99, temperature=0.01, lmbda=0.6,
                                           plan_guidance="rtg_guiding", device="cuda"),
                     synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision="fp32")
    _run(p32, dims, 0, 400, eps)
    am32, er32 = int(p32.last["argmax"].item()), p32.last["expect_return"].clone()
    p32.handle.close()
    p = _planner(dims, 256, 16, 0.01, "rtg_guiding", 0, rescore_delta=delta, rescore_max=rmax)
    with pytest.warns(UserWarning, match="whole window set"):
        _run(p, dims, 0, 400, eps)
    last = p.last
    assert last["saturated"] and last["n_in_window"] > rmax and last["n_rescored"] == last["n_in_window"]
    assert int(last["argmax"].item()) == am32
    top = last["topk"].long()
    assert float((last["expect_return"][top] - er32[top]).abs().max()) <= 5e-5 * float(er32.abs().max())
    if delta > 1e6:  # every candidate re-scored: the whole vector is the fp32 one
        assert last["n_rescored"] == 256
        assert float((last["expect_return"] - er32).abs().max()) <= 5e-5 * float(er32.abs().max())
    _run(p, dims, 0, 400, eps)  # second step: same path, no second warning needed, same answer
    assert int(p.last["argmax"].item()) == am32
    p.handle.close()


def test_a_constant_bf16_offset_larger_than_delta_cannot_move_the_argmax():
    """delta bounds the deviation of (bf16 - fp32) from its COMMON SHIFT, so a shift larger than delta
    could lift an un-re-scored candidate over the re-scored fp32 maximum when the select ran on a vector mixing both scales.
    The select now runs on the merged vector (m3pc_rescore_merge: un-re-scored entries minus the median shift).  Injected:
    +75 on every bf16 score (delta is ~3-6 here)."""
    dims = synth.Dims(11, 3, 32)
    eps = synth.make_eps(256, dims, 3).cuda()
    p = _planner(dims, 256, 16, 0.01, "rtg_guiding", 0)
    ev0 = _run(p, dims, 0, 400, eps).clone()
    am0, er0, d0 = int(p.last["argmax"].item()), p.last["expect_return"].clone(), p.last["delta"]
    p._bf16_offset = 75.0
    assert p._bf16_offset > 5 * d0
    ev1 = _run(p, dims, 0, 400, eps)
    assert int(p.last["argmax"].item()) == am0
    top = p.last["topk"].long()
    assert int(p.last["expect_return"].argmax().item()) in top.tolist()
    # the merged vectors agree: re-scored entries are the same fp32 scores, the others lose the offset with the shift
    assert float((p.last["expect_return"] - er0).abs().max()) <= 1e-3 * float(er0.abs().max()) + 2 * d0
    assert float((ev1 - ev0).abs().max()) <= 2e-2
    assert abs(p.last["shift"] - 75.0) <= 4 * d0 + 10.0  # the estimated shift holds the offset (plus the bf16 scores' own common shift)
    p.handle.close()


def test_delta_grows_when_the_rescored_set_shows_a_larger_deviation():
    """delta is checked every step against the deviation of (bf16 - fp32) over the re-scored set: started from a
    deliberately small value it grows to 1.5 x what the steps see, and the window follows."""
    dims = synth.Dims(11, 3, 32)
    eps = synth.make_eps(256, dims, 3).cuda()
    p = _planner(dims, 256, 16, 0.01, "rtg_guiding", 0)
    _run(p, dims, 0, 400, eps)
    p._delta = 1e-3  # (as if the calibration subset had been unlucky)
    for _ in range(3):
        _run(p, dims, 0, 400, eps)
    assert p.delta_grown >= 1 and p._delta > 0.05
    p.handle.close()


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_non_constant_returns_rows_are_planned_as_the_reference_does(dtype):
    """rtg_guiding consumes whatever trajectory["returns"] holds (learner.py:272-293), not only the constant row
    action_sample builds: a varying returns row goes through the library (m3pc_plan_args::returns) and matches the oracle."""
    from oracle import mtm_oracle as O
    dims = synth.Dims(11, 3, 8, n_embd=64, n_head=2)
    cfg = types.SimpleNamespace(traj_length=8, action_samples=16, horizon=4, discount=0.99, temperature=0.01, lmbda=0.6,
                                plan_guidance="rtg_guiding", device="cuda")
    sd, st = synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0)
    p = HipPlanner(cfg, sd, st, None, n_embd=64, n_head=2)
    hist = synth.make_history(dims, 0)
    ocfg = O.PlanCfg(8, 4, 16, 0.99, 0.01, 0.6, n_head=2)
    win, h = O.assemble_window(ocfg, hist, 100, 3.0)
    win["returns"] = torch.tensor([3.0, 2.5, 2.75, 1.0, 0.5, 4.0, -1.0, 2.0], dtype=dtype).reshape(1, 8, 1)
    eps = synth.make_eps(16, dims, 1)
    ref = O.guiding(sd, O.make_stats(st), ocfg, win, h, 0.6, eps, "rtg")
    p._eps = lambda shape: eps.cuda()
    traj = {k: v.cuda() for k, v in win.items()}
    sa, ev = p.rtg_guiding(traj, h)
    scale = float(ref["expect_return"].abs().max())
    assert float((p.last["expect_return"].cpu() - ref["expect_return"]).abs().max()) <= 5e-5 * scale
    assert int(p.last["argmax"].item()) == ref["argmax"]
    assert float((ev.cpu() - ref["eval_action"]).abs().max()) <= 1e-4
    const = dict(traj)
    const["returns"] = torch.full((1, 8, 1), 3.0, dtype=dtype).cuda()
    p.rtg_guiding(const, h)
    assert float((p.last["expect_return"].cpu() - ref["expect_return"]).abs().max()) > 1e-3 * scale  # the row mattered
    p.handle.close()


def test_fused_heads_ticket_stress():
    """head_f32_fused_kernel (gemm_f32_direct.hip) hands each row tile's partial sums to the LAST of its 16 workgroups through
    an atomic ticket that resets itself, with agent-scope atomic stores / loads and no fence.  Stress: ~1500 few-row
    fp32 scoring passes of random sizes (8 .. 4096 head rows, two heads) alternating between the two chain workspaces on two
    streams that run concurrently; every result must equal, bit for bit, the one the same call produced on an idle device."""
    dims = synth.Dims(11, 3, 32)
    H, CAP = 16, 256  # 256 candidates x 16 scored steps = 4096 rows per head
    hd = capi.Handle(dims.state_dim, dims.action_dim, dims.traj_length, max_candidates=CAP, max_batch=1, max_rescore=CAP)
    hd.load_weights(synth.make_state_dict(dims, 0))
    st = synth.make_tokenizer_stats(dims, 0)
    for k, name in enumerate(synth.KEYS):
        hd.set_tokenizer(k, st[name]["mean"], st[name]["std"], normalize=(name != "actions"))
    hist = synth.make_history(dims, 0)
    s = torch.from_numpy(hist["observations"][100:132]).cuda()
    a = torch.from_numpy(hist["actions"][100:132]).cuda()
    r = torch.from_numpy(hist["rewards"][100:132]).cuda()
    for slot in (0, 1):  # the returns tokens of both slots (PASS 1 leaves them there)
        hd.policy_pass(capi.MODE_RTG, s, a, r, H, 3.0 + slot, slot=slot)
    g = torch.Generator(device="cuda").manual_seed(9)
    cand = torch.rand((CAP, H, dims.action_dim), device="cuda", generator=g) * 2 - 1
    sizes = [1, 2, 3, 8, 17, 31, 64, 100, 129, 200, 255, 256]
    ref = {}
    for slot in (0, 1):
        for n in sizes:
            ref[(slot, n)] = hd.score_actions(capi.MODE_RTG, s, a, r, cand[:n], None, H, 0.6, 0.99, slot=slot).clone()
            torch.cuda.synchronize()
    assert all(torch.isfinite(v).all() for v in ref.values())
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    rng = np.random.default_rng(3)
    got = []
    for it in range(1500):
        slot = it & 1
        n = int(rng.choice(sizes))
        with torch.cuda.stream(streams[slot]):
            got.append((slot, n, hd.score_actions(capi.MODE_RTG, s, a, r, cand[:n], None, H, 0.6, 0.99, slot=slot)))
        if it % 250 == 249:
            torch.cuda.synchronize()
            bad = [(sl, n) for sl, n, v in got if not torch.equal(v, ref[(sl, n)])]
            assert not bad, bad[:8]
            got = []
    hd.close()


_HEADCMP = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, sys.argv[2])
from m3pc_amd import capi, synth
dims = synth.Dims(11, 3, 32)
H, CAP = 16, 256
hd = capi.Handle(dims.state_dim, dims.action_dim, dims.traj_length, max_candidates=CAP, max_batch=1, max_rescore=CAP)
hd.load_weights(synth.make_state_dict(dims, 0))
st = synth.make_tokenizer_stats(dims, 0)
for k, name in enumerate(synth.KEYS):
    hd.set_tokenizer(k, st[name]["mean"], st[name]["std"], normalize=(name != "actions"))
hist = synth.make_history(dims, 0)
s, a, r = (torch.from_numpy(hist[k][100:132]).cuda() for k in ("observations", "actions", "rewards"))
hd.policy_pass(capi.MODE_RTG, s, a, r, H, 3.0, slot=0)
cand = torch.rand((CAP, H, dims.action_dim), device="cuda", generator=torch.Generator(device="cuda").manual_seed(9)) * 2 - 1
np.savez(sys.argv[1], **{str(n): hd.score_actions(capi.MODE_RTG, s, a, r, cand[:n], None, H, 0.6, 0.99, slot=0).cpu().numpy()
                         for n in (1, 8, 17, 64, 129, 256)})
"""


def test_fused_heads_agree_with_the_unfused_chain(tmp_path):
    """The one-launch fp32 scalar heads (head_f32_fused_kernel) against the chain they replaced (GEMM + head_out launches; the lab
    build's M3PC_NO_HEAD_F32_FUSED switch), few-row scoring passes of 1..256 candidates in fresh processes: the two sum the same
    products in different orders, so they agree to fp32 rounding (measured <= 9e-7 of the score scale), not bit for bit."""
    import subprocess
    import sys

    from hip_util import lab_library

    lab_library()  # (built if stale)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lab = os.path.join(root, "m3pc_amd", "libm3pc_hip_lab.so")
    outs = {}
    for tag, extra in (("fused", {}), ("unfused", {"M3PC_NO_HEAD_F32_FUSED": "1"})):
        path = str(tmp_path / f"{tag}.npz")
        env = dict(os.environ, M3PC_LIB=lab, **extra)
        subprocess.run([sys.executable, "-c", _HEADCMP, path, root], env=env, check=True, timeout=300)
        outs[tag] = np.load(path)
    for k in outs["fused"].files:
        a, b = outs["fused"][k], outs["unfused"][k]
        assert np.isfinite(a).all() and np.abs(a - b).max() <= 5e-6 * np.abs(a).max(), (k, float(np.abs(a - b).max()), float(np.abs(a).max()))

"""The multinomial draw of a bf16 plan step, certified: learner.py:324-325 returns
sample_action = a0[torch.multinomial(p, 1)] -- the action every online rollout step executes (replay_buffer.py:206-216) -- and
torch.multinomial(p, 1) is the race arg-max_j p_j / q_j = arg-max_j (tau E_j - log q_j), q ~ Exp(1).

Kernel level: m3pc_topk_race_window / m3pc_rescore_merge_race against the same quantities computed with torch.
Planner level: the bf16 planner's sample index is the fp32 planner's (same variates) and BASELINE config 2's stored-seed
index from the reference (tests/golden/g2_c2.npz: 8) is reproduced in bf16."""
import os
import types

import numpy as np
import pytest
import torch

from m3pc_amd import capi, synth
from m3pc_amd.certificate import RACE_MAX
from m3pc_amd.planner import HipPlanner

pytestmark = pytest.mark.gpu
GD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _handle():
    dims = synth.Dims(11, 3, 8, n_embd=64, n_head=2)
    return capi.Handle(dims.state_dim, dims.action_dim, dims.traj_length, 64, 2, 2, 1, max_candidates=16, max_batch=1, critic_hidden=0)


@pytest.mark.parametrize("n,tau,kmax,rmax", [(1024, 0.01, 128, 32), (625, 1.0, 64, 32), (4096, 1.0, 128, 32), (16384, 0.01, 63, 16),
                                             (16, 0.01, 15, 16), (1, 0.01, 1, 1)])
def test_topk_race_window_lists_against_torch(n, tau, kmax, rmax):
    h = _handle()
    g = torch.Generator(device="cuda").manual_seed(n)
    er = (torch.randn(n, device="cuda", generator=g) * 15.0 + 400.0).contiguous()
    if n >= 8:
        er[5] = er[3]  # an exact tie: the lower index ranks first
    expo = torch.empty(n, device="cuda").exponential_(1, generator=g)
    kmin = min(8, kmax)
    scores = torch.full((rmax + kmax + 1,), float("nan"), device="cuda")
    lst, stats = h.topk_race_window(er, expo, tau, kmax, kmin, rmax, list_scores=scores)
    torch.cuda.synchronize()
    kk = min(kmax + 1, n)
    # score part: descending, ties to the lower index
    order = sorted(range(n), key=lambda i: (-float(er[i]), i))[:kk]
    assert lst[rmax : rmax + kk].tolist() == order
    # race part: the rmax best by tau * E - log q, best racer at rmax - 1, running backwards
    key = (torch.tensor(tau, dtype=torch.float32, device="cuda") * er) - torch.log(expo)
    korder = sorted(range(n), key=lambda i: (-float(key[i]), i))[:rmax]
    got = lst[:rmax].flip(0).tolist()
    if got != korder:  # (logf on the device and torch.log may differ by an ulp: accept a swap of keys closer than that)
        kd = key.double().cpu()
        for a, b in zip(got, korder):
            assert a == b or abs(float(kd[a] - kd[b])) <= 4e-6 * max(1.0, abs(float(kd[a]))), (a, b)
    assert torch.equal(scores[rmax - rmax : rmax + kk], er[lst[: rmax + kk].long()])
    assert float(stats[2]) == float(er.max())
    # without the window statistics (what the planner asks for): up to 2048 candidates the ranking launch writes the listed scores itself
    scores2 = torch.full((rmax + kmax + 1,), float("nan"), device="cuda")
    lst2, st2 = h.topk_race_window(er, expo, tau, kmax, kmin, rmax, list_scores=scores2, want_stats=False)
    torch.cuda.synchronize()
    assert st2 is None and torch.equal(lst2[: rmax + kk], lst[: rmax + kk]) and torch.equal(scores2[: rmax + kk], scores[: rmax + kk])
    h.close()


@pytest.mark.parametrize("n,tau,r,nn,delta", [(1024, 0.01, 2, 8, 5.0), (1024, 0.01, 0, 8, 5.0), (4096, 1.0, 5, 12, 0.05), (300, 0.01, 32, 128, 50.0)])
def test_rescore_merge_race_against_torch(n, tau, r, nn, delta):
    """c, deviation, merged, need and need_race of m3pc_rescore_merge_race restated with torch on synthetic 'bf16' scores
    b = f + shift + noise."""
    h = _handle()
    g = torch.Generator(device="cuda").manual_seed(7 * n + r)
    scale = 15.0 if tau < 0.1 else 0.5
    f_all = torch.randn(n, device="cuda", generator=g) * scale + 100.0
    b = (f_all + 3.25 + (torch.rand(n, device="cuda", generator=g) - 0.5) * 1.6 * delta).contiguous()
    expo = torch.empty(n, device="cuda").exponential_(1, generator=g)
    kmax, rmax = 128, RACE_MAX
    scores = torch.empty((rmax + kmax + 1,), device="cuda")
    lst, _ = h.topk_race_window(b, expo, tau, kmax, min(nn, kmax), rmax, list_scores=scores)
    o = rmax - r
    sub = lst[o : rmax + nn].contiguous()
    fl = f_all[sub.long()].contiguous()
    merged, stats = h.rescore_merge_race(b, expo, tau, sub, r, nn, scores[o:].contiguous(), fl, delta=delta)
    torch.cuda.synchronize()
    st = stats.cpu().tolist()
    d = b[sub.long()] - fl
    c = float(torch.sort(d).values[(d.numel() - 1) // 2])
    assert st[0] == c
    assert abs(st[1] - float((d - c).abs().max())) <= 1e-6
    ref = b - c
    ref[sub.long()] = fl
    assert torch.equal(merged, ref)
    fbest = float(fl.max())
    assert int(st[2]) == int((b > torch.tensor(fbest + c - delta, dtype=torch.float32)).sum())
    t32 = torch.tensor(tau, dtype=torch.float32, device="cuda")
    kf = (t32 * fl - torch.log(expo[sub.long()])).max()
    thr = kf + t32 * torch.tensor(c - delta, dtype=torch.float32, device="cuda")
    kb = t32 * b - torch.log(expo)
    want = int((kb >= thr).sum())
    near = int(((kb - thr).abs() <= 4e-6 * thr.abs().clamp(min=1.0)).sum())  # (device logf vs torch.log: an ulp at the edge)
    assert abs(int(st[5]) - want) <= near, (st, want, near)
    assert abs(st[6] - float(kf)) <= 4e-6 * max(1.0, abs(float(kf)))
    # the certificate means what it says: when need_race <= r every candidate that can reach K* is among the r listed racers
    if int(st[5]) <= r and near == 0:
        racers = set(sub[:r].tolist())
        assert set(torch.nonzero(kb >= thr).flatten().tolist()) <= racers
    h.close()


def _cfg(T, N, H, tau=0.01, guidance="rtg_guiding"):
    return types.SimpleNamespace(traj_length=T, action_samples=N, horizon=H, discount=0.99, temperature=tau, lmbda=0.6,
                                 plan_guidance=guidance, device="cuda")


def test_c2_bf16_reproduces_the_reference_sample_index():
    """BASELINE config 2, bf16 + certified re-score: the reference's own multinomial draw for the stored seed
    (torch.multinomial(p, 1, generator=manual_seed(77)) -> index 8, tests/golden/make_golden.py) is reproduced: the planner
    draws with the same Exp(1) variates ATen's CPU multinomial uses."""
    g = np.load(os.path.join(GD, "g2_c2.npz"))
    dims = synth.Dims(11, 3, 32)
    want = int(g["sample_idx"].reshape(-1)[0])
    q = torch.empty(1024, dtype=torch.float32).exponential_(1, generator=torch.Generator().manual_seed(77))
    pr = torch.from_numpy(g["p"].reshape(-1))
    assert int(torch.argmax(pr / q)) == want  # (the replay IS ATen's draw)
    for prec in ("bf16", "fp32"):
        p = HipPlanner(_cfg(32, 1024, 16), synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision=prec)
        eps = synth.make_eps(1024, dims, 1).cuda()
        p._eps = lambda shape: eps
        p._draw_expo = lambda: q.cuda()
        hist = synth.make_history(dims, 0)
        hist["path_length"] = 500
        sa = p.action_sample(hist, plan=True, eval=False, rtg=3.0)
        last = p.last
        assert int(last["argmax"].item()) == int(g["argmax"])
        assert int(last["sample_idx"].item()) == want, (prec, int(last["sample_idx"].item()), want)
        assert np.abs(sa.cpu().numpy().reshape(-1) - g["sample_action"].reshape(-1)).max() < 2e-5
        if prec == "bf16":
            assert want in last["race"].tolist() and last["need_race"] <= last["n_race"] <= RACE_MAX and not last["saturated"]
        p.handle.close()


@pytest.mark.parametrize("env,guidance,tau,N,T,H", [("hopper", "rtg_guiding", 0.01, 1024, 32, 16), ("walker2d", "critic_lambda_guiding", 1.0, 512, 32, 16),
                                                    ("hopper", "rtg_guiding", 0.01, 625, 8, 4)])
def test_bf16_sample_index_is_the_fp32_planners(env, guidance, tau, N, T, H):
    """A handful of steps per shape (the long sweeps live in test_certificate_gpu.py): same candidates, same Exp(1) variates,
    bf16 + certified re-score against fp32 end to end -- arg-max, multinomial index and sampled action identical."""
    S, A = synth.ENV_DIMS[env]
    dims = synth.Dims(S, A, T)
    qsd, om, os_ = synth.make_critic(dims, 0)
    mk = lambda prec: HipPlanner(_cfg(T, N, H, tau, guidance), synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), qsd, om,
                                 os_, precision=prec, generator=torch.Generator(device="cuda").manual_seed(5))
    pb, pf = mk("bf16"), mk("fp32")
    mode = capi.MODE_RTG if guidance == "rtg_guiding" else capi.MODE_CRITIC
    for t in range(8):
        hist = synth.make_history(dims, t)
        hist["path_length"] = [500, 37, 321, 998, 640, 77, 250, 123][t]
        eps = synth.make_eps(N, dims, 50 + t).cuda()
        s, a, r, h, rtg = pb.assemble_window(hist, rtg=3.0)
        sab, _ = pb._guide(mode, s, a, r, rtg, h, 0.6, eps=eps)
        saf, _ = pf._guide(mode, s, a, r, rtg, h, 0.6, eps=eps)
        assert torch.equal(pb.last["argmax"], pf.last["argmax"])
        assert torch.equal(pb.last["sample_idx"], pf.last["sample_idx"]), (t, pb.last["sample_idx"], pf.last["sample_idx"], pb.last["need_race"])
        assert torch.equal(sab, saf)
    pb.handle.close()
    pf.handle.close()


def test_noise_adding_lambda_in_bf16_returns_the_fp32_planners_actions():
    """noise_adding_lambda (learner.py:142-208: mean + 0.09 randn candidates, critic scoring, temperature 1) through the certified
    bf16 step: both planners draw the same variates from equally seeded generators; arg-max, multinomial index and both returned
    actions equal the fp32 planner's."""
    dims = synth.Dims(17, 6, 16)
    qsd, om, os_ = synth.make_critic(dims, 0)
    mk = lambda prec: HipPlanner(_cfg(16, 512, 8, 1.0, "noise_adding_lambda"), synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0),
                                 qsd, om, os_, precision=prec, generator=torch.Generator(device="cuda").manual_seed(17))
    pb, pf = mk("bf16"), mk("fp32")
    for t in range(6):
        hist = synth.make_history(dims, t)
        hist["path_length"] = [300, 12, 640, 997, 55, 4][t]
        for ev in (True, False):
            ab = pb.action_sample(hist, plan=True, eval=ev, rtg=3.0)
            af = pf.action_sample(hist, plan=True, eval=ev, rtg=3.0)
            assert torch.equal(pb.last["argmax"], pf.last["argmax"]) and torch.equal(pb.last["sample_idx"], pf.last["sample_idx"]), (t, ev)
            if ev:
                assert float((ab - af).abs().max()) <= 2e-2
            else:
                assert torch.equal(ab, af)
    pb.handle.close()
    pf.handle.close()

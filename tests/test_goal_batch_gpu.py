"""BASELINE config 5 at its size: the zero-shot goal-reaching call (research/zeroshot_omtm/learner.py:151-261 action_piid_sample,
60-149 action_id_sample; masks research/zeroshot_omtm/masks.py:30-91) for thousands of windows per launch through the exactly
pruned many-window path (m3pc_goal_step_batch) -- against the reference goldens (G3), the oracle, and the single-window call."""
import os
import types

import numpy as np
import pytest
import torch

from m3pc_amd import capi, synth
from m3pc_amd.planner import HipPlanner
from oracle import mtm_oracle as O

pytestmark = pytest.mark.gpu
GD = os.path.join(os.path.dirname(__file__), "golden")
T, S, A = 8, 11, 3
BF16_LOC_TOL = 3e-2   # |loc_bf16 - loc_fp32|, loc = O(1): bf16 operands through two chained forwards (measured ~1e-2 at most)
BF16_STD_REL = 8e-2   # std = exp(-5 + 3.5 (tanh(.) + 1)): relative


def _planner(goal_batch, precision="bf16", max_batch=1):
    dims = synth.Dims(S, A, T)
    cfg = types.SimpleNamespace(traj_length=T, action_samples=1, horizon=4, discount=0.99, temperature=1.0, lmbda=0.6,
                                plan_guidance="rtg_guiding", index_jump=4)
    return dims, HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision=precision,
                            goal_batch=goal_batch, max_batch=max_batch)


def _windows(E, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn((E, T, S), generator=g).cuda(), (torch.rand((E, T, A), generator=g) * 2 - 1).cuda()


def _golden_window(p, g, pl):
    dims = synth.Dims(S, A, T)
    hist = synth.make_history(dims, 0)
    hist["observations"] = g[f"obs_pl{pl}"]
    hist["path_length"] = pl
    s, a, r, h, _ = p.assemble_goal_window(hist, rtg=2.5)
    return s.clone(), a.clone(), h


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_golden_windows_inside_a_large_batch(precision):
    """G3's four windows (three different horizons) embedded at scattered rows of batches of 8192 / 300 random windows: the
    rows reproduce the reference's loc / std at T-h, eval action and the observation rows its second forward saw."""
    g = np.load(os.path.join(GD, "g3_zeroshot.npz"))
    _, p = _planner(8192, precision)
    prec = capi.PREC_FP32 if precision == "fp32" else capi.PREC_BF16
    for pl, E, row in ((0, 300, 17), (2, 300, 299), (37, 8192, 4097), (997, 8192, 8191)):
        s, a, h = _golden_window(p, g, pl)
        assert h == int(g[f"action_piid_sample_pl{pl}_horizon"])
        idx = T - h
        st, ac = _windows(E, pl)
        st[row], ac[row] = s, a
        mu, sd, win = p.handle.goal_step_batch(st, ac, idx, capi.GOAL_PIID, prec, want_window=True)
        torch.cuda.synchronize()
        pre = f"action_piid_sample_pl{pl}_"
        loc_g, std_g = g[pre + "loc"][0, idx, 0], g[pre + "std"][0, idx, 0]
        ws_g = g[pre + "win_states_after"][0]
        if precision == "fp32":
            assert np.abs(mu[row].cpu().numpy() - loc_g).max() <= 2e-5
            assert np.abs(sd[row].cpu().numpy() - std_g).max() <= 2e-5 * max(1.0, float(np.abs(std_g).max()))
            assert np.abs(win[row].cpu().numpy() - ws_g).max() <= 1e-4
            assert np.abs(torch.tanh(mu[row]).cpu().numpy() - g[pre + "eval_action"][0]).max() <= 2e-5
        else:
            assert np.abs(mu[row].cpu().numpy() - loc_g).max() <= BF16_LOC_TOL
            assert np.abs(sd[row].cpu().numpy() / std_g - 1).max() <= BF16_STD_REL
            assert np.abs(win[row].cpu().numpy() - ws_g).max() <= 3e-2 * max(1.0, float(np.abs(ws_g).max()))
        # rows the overlay does not touch are the caller's
        keep = [t for t in range(T) if not (t <= idx or (idx + 2 <= t < T - 1))]
        assert torch.equal(win[:, keep], st[:, keep])
        # action_id_sample (one forward under the gid mask) on the same batch
        mu2, sd2 = p.handle.goal_step_batch(st, ac, idx, capi.GOAL_ID, prec)
        pre2 = f"action_id_sample_pl{pl}_"
        tol = 2e-5 if precision == "fp32" else BF16_LOC_TOL
        assert np.abs(mu2[row].cpu().numpy() - g[pre2 + "loc"][0, idx, 0]).max() <= tol
    p.handle.close()


def test_fp32_rows_at_8192_equal_the_single_window_call_and_do_not_depend_on_the_batch():
    """fp32, E = 8192: sampled rows equal the un-pruned single-window call (m3pc_goal_step, the few-row fp32 kernels) to fp32
    rounding, and are BIT-identical to the same windows planned in smaller batches (what environment sharding relies on)."""
    _, p = _planner(8192, "fp32")
    from m3pc_amd.masks import create_fid_mask, create_pi_mask, mask_rows
    for h in (4, 2, 7):
        idx = T - h
        st, ac = _windows(8192, 100 + h)
        mu, sd, win = p.handle.goal_step_batch(st, ac, idx, capi.GOAL_PIID, capi.PREC_FP32, want_window=True)
        rows = [0, 1, 127, 128, 4095, 6000, 8191]
        for r in rows:
            m1, s1, inf1, w1 = p.handle.goal_step(st[r : r + 1], ac[r : r + 1], torch.zeros((1, T, 1), device="cuda"), [2.5],
                                                  mask_rows(create_pi_mask(T, "cpu", idx)), mask_rows(create_fid_mask(T, "cpu", idx)), idx)
            assert float((m1[0, idx] - mu[r]).abs().max()) <= 2e-5
            assert float((s1[0, idx] / sd[r] - 1).abs().max()) <= 1e-4
            assert float((w1[0] - win[r]).abs().max()) <= 1e-4 * max(1.0, float(w1.abs().max()))
        for lo, hi in ((0, 2048), (4096, 4096 + 1000), (8192 - 512, 8192)):
            m2, s2, w2 = p.handle.goal_step_batch(st[lo:hi].contiguous(), ac[lo:hi].contiguous(), idx, capi.GOAL_PIID, capi.PREC_FP32,
                                                  want_window=True)
            assert torch.equal(m2, mu[lo:hi]) and torch.equal(s2, sd[lo:hi]) and torch.equal(w2, win[lo:hi])
    p.handle.close()


def test_bf16_at_8192_against_the_oracle_and_the_fp32_path():
    """bf16, E = 8192: loc within BF16_LOC_TOL of the fp32 path on every row and of the ORACLE on sampled rows; a shard of the
    windows gives the same bits as the whole batch (same kernel regime)."""
    dims, p = _planner(8192, "bf16")
    sd_w = synth.make_state_dict(dims, 0)
    stats = O.make_stats(synth.make_tokenizer_stats(dims, 0))
    ocfg = O.PlanCfg(T, 4, 1, 0.99, 1.0, 0.6)
    h = 4
    idx = T - h
    st, ac = _windows(8192, 7)
    mu_b, sd_b, win_b = p.handle.goal_step_batch(st, ac, idx, capi.GOAL_PIID, capi.PREC_BF16, want_window=True)
    mu_f, sd_f, win_f = p.handle.goal_step_batch(st, ac, idx, capi.GOAL_PIID, capi.PREC_FP32, want_window=True)
    torch.cuda.synchronize()
    assert float((mu_b - mu_f).abs().max()) <= BF16_LOC_TOL
    assert float((sd_b / sd_f - 1).abs().max()) <= BF16_STD_REL
    assert float((win_b - win_f).abs().max()) <= 3e-2 * max(1.0, float(win_f.abs().max()))
    for r in (0, 1234, 8191):
        traj = dict(states=st[r : r + 1].cpu(), actions=ac[r : r + 1].cpu(), rewards=torch.zeros((1, T, 1)),
                    returns=torch.full((1, T, 1), 2.5, dtype=torch.float64))
        loc, std, _ = O.goal_piid(sd_w, stats, ocfg, traj, h)
        assert float((loc[0, idx, 0] - mu_f[r].cpu()).abs().max()) <= 2e-5
        assert float((loc[0, idx, 0] - mu_b[r].cpu()).abs().max()) <= BF16_LOC_TOL
        assert float((std[0, idx, 0] / sd_b[r].cpu() - 1).abs().max()) <= BF16_STD_REL
    for lo, hi in ((0, 4096), (4096, 8192), (1024, 2048 + 1024)):
        m2, s2 = p.handle.goal_step_batch(st[lo:hi].contiguous(), ac[lo:hi].contiguous(), idx, capi.GOAL_PIID, capi.PREC_BF16)
        assert torch.equal(m2, mu_b[lo:hi]) and torch.equal(s2, sd_b[lo:hi])
    # Shards that fall into ANOTHER kernel regime than the whole batch (include/m3pc_hip.h, m3pc_goal_step_batch: below 2048
    # windows the call runs as one part, below 12288 token rows the fused layer tails give way to the split / GEMM forms, below
    # 1024 (window, head) items the direct attention kernel runs) -- a remainder shard of an environment-sharded run: the
    # bf16 bits may differ there; what holds is the fp32-level agreement every bf16 result is held to.  fp32 calls
    # stay bit-identical at any size.
    for lo, hi in ((0, 1500), (5000, 5200), (8000, 8040)):
        m2, s2 = p.handle.goal_step_batch(st[lo:hi].contiguous(), ac[lo:hi].contiguous(), idx, capi.GOAL_PIID, capi.PREC_BF16)
        assert float((m2 - mu_f[lo:hi]).abs().max()) <= BF16_LOC_TOL and float((s2 / sd_f[lo:hi] - 1).abs().max()) <= BF16_STD_REL
        assert float((m2 - mu_b[lo:hi]).abs().max()) <= 2 * BF16_LOC_TOL
        m3, s3 = p.handle.goal_step_batch(st[lo:hi].contiguous(), ac[lo:hi].contiguous(), idx, capi.GOAL_PIID, capi.PREC_FP32)
        assert torch.equal(m3, mu_f[lo:hi]) and torch.equal(s3, sd_f[lo:hi])
    p.handle.close()


def test_planner_batch_call_takes_the_pruned_path_beyond_64_windows():
    """action_piid_sample_batch with E = 200 histories of mixed horizons (pruned path, fp32 planner): every row equals the
    single-window action_piid_sample."""
    dims, p = _planner(256, "fp32")
    rng = np.random.RandomState(0)
    hists = []
    for i in range(200):
        hst = synth.make_history(dims, i % 5)
        hst["path_length"] = int(rng.choice([0, 1, 2, 3, 50, 400, 996, 997, 998]))
        hists.append(hst)
    acts = p.action_piid_sample_batch(hists, percentage=1.0, eval=True, rtg=2.5)
    assert acts.shape == (200, 3)
    for i in (0, 7, 13, 31, 63, 150, 199):
        one = p.action_piid_sample(hists[i], eval=True, rtg=2.5)
        assert float((one - acts[i]).abs().max()) < 2e-5
    sa = p.action_piid_sample_batch(hists, percentage=1.0, eval=False, rtg=2.5)
    assert sa.shape == (200, 3) and float(sa.abs().max()) <= 1.0
    with pytest.raises(ValueError):
        p.action_piid_sample_batch(hists * 2, percentage=1.0, eval=True, rtg=2.5)
    p.handle.close()

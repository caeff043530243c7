"""Pipelined plan steps (HipPlanner.plan_async / PlanTicket, rollout.PipelinedPlanner) against the serial order.

The reference plans one window per call and reads the action back before the next one (replay_buffer.py:204-232;
learner.py:645-741 runs its evaluation episodes one after the other).  For independent windows the planner keeps several plan
steps in flight: candidate passes back to back on the caller's stream, the policy pass of the next step and the fp32
re-score + select of the previous one on a second stream in the library's chain workspace.  What must hold: every step's
scores, arg-max, multinomial index and actions are BIT-IDENTICAL to the serial call (same kernels, same rows, same draws)."""
import types

import numpy as np
import pytest
import torch

from m3pc_amd import capi, synth
from m3pc_amd.planner import HipPlanner

pytestmark = pytest.mark.gpu


def _cfg(T, N, H, temp, guidance):
    return types.SimpleNamespace(traj_length=T, action_samples=N, horizon=H, discount=0.99, temperature=temp, lmbda=0.6,
                                 plan_guidance=guidance, device="cuda")


def _windows(dims, n):
    out = []
    for i in range(n):
        h = synth.make_history(dims, i % 3)
        h["path_length"] = [500, 37, 321, 998, 5, 640, 77, 250][i % 8]
        out.append(h)
    return out


def _planner(dims, N, H, guidance, precision, seed=11, **kw):
    qsd, om, os_ = synth.make_critic(dims, 0) if "critic" in guidance else (None, None, None)
    gen = torch.Generator(device="cuda").manual_seed(seed)
    return HipPlanner(_cfg(dims.traj_length, N, H, 0.01 if guidance == "rtg_guiding" else 1.0, guidance), synth.make_state_dict(dims, 0),
                      synth.make_tokenizer_stats(dims, 0), qsd, om, os_, precision=precision, generator=gen, **kw)


KEYS = ("expect_return", "p", "argmax", "sample_idx", "eval_action", "sample_action", "sample_actions", "loc", "std")


def _snap(info):
    return {k: info[k].clone() for k in KEYS}


@pytest.mark.parametrize("guidance,precision,N,T,H", [
    ("rtg_guiding", "bf16", 1024, 32, 16),          # BASELINE config 2: two candidate halves + chain stream = three streams
    ("rtg_guiding", "fp32", 64, 16, 8),             # BASELINE config 1 shapes, no re-score
    ("critic_lambda_guiding", "bf16", 512, 32, 16),  # critic scoring (config 3 shapes, smaller N)
])
@pytest.mark.parametrize("depth", [1, 2, 3])
def test_pipelined_steps_are_bit_identical_to_the_serial_order(guidance, precision, N, T, H, depth):
    S, A = (11, 3) if guidance == "rtg_guiding" else (17, 6)
    dims = synth.Dims(S, A, T)
    wins = _windows(dims, 7)
    ps = _planner(dims, N, H, guidance, precision)
    serial = []
    for w in wins:
        ps.action_sample(w, plan=True, eval=True, rtg=3.0)
        serial.append(_snap(ps.last))
    n_re_serial = [ps.last.get("n_rescored")]
    ps.handle.close()
    pp = _planner(dims, N, H, guidance, precision, pipeline_depth=depth)
    flight, got = [], []
    for w in wins:
        flight.append(pp.plan_async(w, eval=True, rtg=3.0))
        if len(flight) > depth:
            tk = flight.pop(0)
            ev = tk.result()
            got.append((_snap(tk.info), ev.clone()))
    while flight:
        tk = flight.pop(0)
        ev = tk.result()
        got.append((_snap(tk.info), ev.clone()))
    torch.cuda.synchronize()
    assert len(got) == len(serial)
    for i, ((g, ev), s) in enumerate(zip(got, serial)):
        for k in KEYS:
            assert torch.equal(g[k], s[k]), (i, k)
        assert torch.equal(ev, s["eval_action"]), i
    assert n_re_serial is not None
    pp.handle.close()


def test_pipelined_batch_call_equals_single_window_calls():
    """action_sample_batch (default: pipelined) over 6 windows == 6 action_sample calls, bit for bit, sampled actions too."""
    dims = synth.Dims(11, 3, 32)
    wins = _windows(dims, 6)
    ps = _planner(dims, 512, 16, "rtg_guiding", "bf16", seed=5)
    one = [ps.action_sample(w, plan=True, eval=False, rtg=2.0).clone() for w in wins]
    ps.handle.close()
    pb = _planner(dims, 512, 16, "rtg_guiding", "bf16", seed=5)
    out = pb.action_sample_batch(wins, eval=False, rtg=2.0)
    assert out.shape == (6, 3)
    for i in range(6):
        assert torch.equal(out[i], one[i][0]), i
    assert len(pb.last["windows"]) == 6 and all(w["horizon"] in (16, 27) for w in pb.last["windows"])
    pb.handle.close()


def test_tickets_resolve_out_of_order_and_slots_recycle():
    """More tickets than slots: issuing a step into a slot whose previous owner is unresolved resolves that owner first;
    results stay available on the ticket, in any order."""
    dims = synth.Dims(11, 3, 16)
    wins = _windows(dims, 9)
    ps = _planner(dims, 128, 8, "rtg_guiding", "bf16", seed=3)
    serial = [ps.action_sample(w, plan=True, eval=True, rtg=3.0).clone() for w in wins]
    ps.handle.close()
    pp = _planner(dims, 128, 8, "rtg_guiding", "bf16", seed=3)
    tickets = [pp.plan_async(w, eval=True, rtg=3.0) for w in wins]  # 9 tickets, capi.SLOTS slots
    assert sum(t.out is not None for t in tickets) >= len(wins) - capi.SLOTS
    for i in reversed(range(len(wins))):
        assert torch.equal(tickets[i].result(), serial[i]), i
    pp.handle.close()


def test_weight_update_between_pipelined_steps_is_ordered():
    """load_state_dict resolves the steps in flight before it touches the weights: the tickets issued before it hold the OLD
    weights' results, the step issued after it the NEW weights' -- each compared by value with a serial planner fed the same
    variates (the values, not the shapes)."""
    dims = synth.Dims(11, 3, 16)
    wins = _windows(dims, 3)
    eps = synth.make_eps(128, dims, 1).cuda()
    sd1 = synth.make_state_dict(dims, 1)

    def fixed(p):
        p._eps = lambda shape: eps
        return p

    pp = fixed(_planner(dims, 128, 8, "rtg_guiding", "fp32", seed=3))
    t0 = pp.plan_async(wins[0], eval=True, rtg=3.0)
    t1 = pp.plan_async(wins[1], eval=True, rtg=3.0)
    pp.load_state_dict(sd1)
    assert t0.out is not None and t1.out is not None
    a = pp.action_sample(wins[2], plan=True, eval=True, rtg=3.0).clone()
    old0, old1 = t0.result().clone(), t1.result().clone()
    pp.handle.close()
    po = fixed(_planner(dims, 128, 8, "rtg_guiding", "fp32", seed=3))  # the old weights, serial
    assert torch.equal(po.action_sample(wins[0], plan=True, eval=True, rtg=3.0), old0)
    assert torch.equal(po.action_sample(wins[1], plan=True, eval=True, rtg=3.0), old1)
    stale = po.action_sample(wins[2], plan=True, eval=True, rtg=3.0).clone()  # what stale weights would have planned
    po.handle.close()
    pr = fixed(_planner(dims, 128, 8, "rtg_guiding", "fp32", seed=3))  # the new weights, serial
    pr.load_state_dict(sd1)
    c = pr.action_sample(wins[2], plan=True, eval=True, rtg=3.0)
    assert torch.equal(a, c) and not torch.equal(a, stale)
    pr.handle.close()


@pytest.mark.parametrize("prec,rescore", [("fp32", "bound"), ("bf16", "topk")])
def test_more_windows_than_twice_the_slots_without_host_reads(prec, rescore):
    """fp32 mode and the fixed top-k re-score read nothing back per step, so the host runs far ahead of the device and re-fills a
    slot's pinned window buffer while -- without the slot's copy event -- the H2D copy of the window that used it before may still be
    queued.  2 SLOTS + 3 windows through action_sample_batch against the serial calls."""
    dims = synth.Dims(11, 3, 32)
    n = 2 * capi.SLOTS + 3
    wins = _windows(dims, n)
    kw = dict(rescore=rescore) if prec == "bf16" else {}
    ps = _planner(dims, 1024, 16, "rtg_guiding", prec, seed=5, **kw)
    serial = [ps.action_sample(w, plan=True, eval=True, rtg=3.0).clone() for w in wins]
    ps.handle.close()
    pp = _planner(dims, 1024, 16, "rtg_guiding", prec, seed=5, **kw)
    got = pp.action_sample_batch(wins, eval=True, rtg=3.0)
    for i in range(n):
        assert torch.equal(got[i], serial[i]), i
    pp.handle.close()


def test_deferred_join_orders_other_users_of_the_candidate_workspace():
    """C-ABI contract of M3PC_PLAN_DEFER_JOIN: the second candidate half stays un-joined on the library's stream until
    m3pc_candidate_join -- but every other call into the candidate workspace (here: m3pc_forward right behind the pass, on the same
    stream, nothing joined) is ordered behind it by the library, and the joined scores equal those of an ordinary pass."""
    dims = synth.Dims(11, 3, 32)
    N, H = 1024, 16
    p = _planner(dims, N, H, "rtg_guiding", "bf16", max_batch=4)
    hd = p.handle
    hist = synth.make_history(dims, 0)
    hist["path_length"] = 500
    s, a, r, h, rtg = p.assemble_window(hist, rtg=3.0)
    eps = synth.make_eps(N, dims, 1).cuda().reshape(N, -1, 3)
    from m3pc_amd.masks import create_rcbc_mask, mask_rows
    toks = [hd.tokenize(capi.STATES, torch.stack([s] * 4)), torch.stack([a] * 4), hd.tokenize(capi.REWARDS, torch.stack([r] * 4)),
            hd.tokenize(capi.RETURNS, torch.full((4, 32, 1), 3.0, dtype=torch.float64, device="cuda"))]
    masks = mask_rows(create_rcbc_mask(32, "cpu", 16))
    want_fwd = [t.clone() for t in hd.forward(toks, masks, want=("actions",), precision=capi.PREC_BF16)["actions"]]
    hd.policy_pass(capi.MODE_RTG, s, a, r, h, rtg, slot=1)
    ref = hd.candidate_pass(capi.MODE_RTG, s, a, r, eps, h, 0.6, 0.99, N, precision=capi.PREC_BF16, slot=1)
    want_er = ref["expect_return"].clone()
    torch.cuda.synchronize()
    for _ in range(3):
        res = hd.candidate_pass(capi.MODE_RTG, s, a, r, eps, h, 0.6, 0.99, N, precision=capi.PREC_BF16, slot=1, defer_join=True)
        got_fwd = hd.forward(toks, masks, want=("actions",), precision=capi.PREC_BF16)["actions"]  # (same workspace, no join)
        hd.candidate_join(1)
        torch.cuda.synchronize()
        assert torch.equal(res["expect_return"], want_er)
        assert torch.equal(got_fwd[0], want_fwd[0]) and torch.equal(got_fwd[1], want_fwd[1])
    hd.close()


@pytest.mark.parametrize("guidance,N,T,H", [("rtg_guiding", 1024, 32, 16), ("critic_lambda_guiding", 512, 32, 16), ("rtg_guiding", 625, 8, 4)])
def test_chain_modes_and_options_give_the_same_bits(guidance, N, T, H):
    """Round 5: `chain_mode="alternate"` (a step's policy pass and re-score on the chain stream of its slot's parity, two chain
    workspaces per kind in the library, the tail of step t enqueued behind the policy pass of step t + 2) against "split" (rounds
    3-4: every policy pass on one stream, every re-score on the other) and against the serial order: 14 windows, three in flight --
    every step's scores, certificates' outcomes (arg-max, multinomial index) and actions bit for bit, and the same number of
    candidates re-scored (the adaptive first pass is lagged by the slot count precisely so that it does not depend on the order)."""
    S, A = (11, 3) if guidance == "rtg_guiding" else (17, 6)
    dims = synth.Dims(S, A, T)
    wins = _windows(dims, 14)

    def run(**kw):
        p = _planner(dims, N, H, guidance, "bf16", pipeline_depth=3, **kw)
        flight, out = [], []
        for i, w in enumerate(wins):
            flight.append(p.plan_async(w, eval=bool(i % 2), rtg=3.0))
            if len(flight) > 3:
                tk = flight.pop(0)
                res = tk.result().clone()
                out.append((_snap(tk.info), res, tk.info["n_rescored"], tk.info["n_race"]))
        while flight:
            tk = flight.pop(0)
            res = tk.result().clone()
            out.append((_snap(tk.info), res, tk.info["n_rescored"], tk.info["n_race"]))
        torch.cuda.synchronize()
        p.handle.close()
        return out

    ps = _planner(dims, N, H, guidance, "bf16")
    serial = []
    for i, w in enumerate(wins):
        res = ps.action_sample(w, plan=True, eval=bool(i % 2), rtg=3.0).clone()
        serial.append((_snap(ps.last), res, ps.last["n_rescored"], ps.last["n_race"]))
    ps.handle.close()
    for name, got in (("alternate", run(chain_mode="alternate")), ("split", run(chain_mode="split")),
                      ("alternate, one chain stream", run(chain_mode="alternate", tail_stream=False))):
        for i, ((g, res, nr, nc), (s, res_s, nr_s, nc_s)) in enumerate(zip(got, serial)):
            for k in KEYS:
                assert torch.equal(g[k], s[k]), (name, i, k)
            assert torch.equal(res, res_s) and (nr, nc) == (nr_s, nc_s), (name, i, nr, nr_s, nc, nc_s)

#!/usr/bin/env python3
"""Companion of tools/pairing_probe.py for the OTHER fp32 chain: what would ONE policy pass at batch 2 for two consecutive
independent plan steps buy (m3pc_policy_pass_batch exists; the planner issues one step at a time)?  Timing stand-in: every odd
step skips its policy pass (its slot keeps an older step's policy head: a stale result, timing only), every even step runs its
(single) policy pass -- a lower bound of a batch-2 pass's cost.   python tools/policy_pairing_probe.py [steps]"""
import os
import sys
import time
import types
from collections import deque

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m3pc_amd import capi, synth  # noqa: E402
from m3pc_amd import planner as P  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 60
MODE = {"paired": False, "n": 0}
orig_pp = capi.Handle.policy_pass


def pp(self, mode, states, actions, rewards, horizon, rtg, slot=0, returns=None, loc=None, std=None, pruned=False):
    if not MODE["paired"]:
        return orig_pp(self, mode, states, actions, rewards, horizon, rtg, slot=slot, returns=returns, loc=loc, std=std, pruned=pruned)
    MODE["n"] += 1
    if MODE["n"] & 1:
        return None  # (the partner's batch-2 pass would have left this step's head)
    # (a batch-2 pass leaves its heads in the batched-slot form m3pc_rescore does not read; the single pass is the same ~30-launch
    # chain on 65 instead of 130 rows -- latency-bound launches: a lower bound of the paired pass's cost, so an UPPER bound of the gain)
    return orig_pp(self, mode, states, actions, rewards, horizon, rtg, slot=slot, returns=returns, loc=loc, std=std, pruned=pruned)


capi.Handle.policy_pass = pp
dims = synth.Dims(11, 3, 32)
cfg = types.SimpleNamespace(traj_length=32, action_samples=1024, horizon=16, discount=0.99, temperature=0.01, lmbda=0.6,
                            plan_guidance="rtg_guiding")
hist = synth.make_history(dims, 0)
hist["path_length"] = 500
for rep in range(3):
    for paired in (False, True):
        MODE["paired"] = False
        pl = P.HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision="bf16",
                          generator=torch.Generator(device="cuda").manual_seed(1), pipeline_depth=3, auto_fp32=False, max_batch=2)
        s, a, r, h, rtg = pl.assemble_window(hist, rtg=3.0)

        def run(k):
            flight = deque()
            for _ in range(k):
                flight.append(pl._issue(capi.MODE_RTG, s, a, r, rtg, h, 0.6, pipelined=True, inputs_ready=True))
                if len(flight) > 3:
                    flight.popleft().pair()
            while flight:
                flight.popleft().pair()

        run(24)
        MODE["paired"] = paired
        run(10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(STEPS)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%-9s %.1f plan-steps/s  (%.4f ms/step)" % ("paired:" if paired else "per step:", STEPS / dt, 1e3 * dt / STEPS), flush=True)
        MODE["paired"] = False
        pl.handle.close()

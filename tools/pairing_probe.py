#!/usr/bin/env python3
"""VERDICT r5 item 7 asked for a measurement: ONE fp32 re-score chain for the lists of TWO consecutive plan steps instead of one
chain per step.  Building it needs a two-slot gather of the returns tokens inside m3pc_score_actions and a fixed-row tiling; what
it could buy at most is measured here with a timing stand-in that costs exactly what pairing would: every odd step skips its
re-score chain (select on the bf16 scores: a WRONG result for that step, timing only), every even step re-scores twice its list,
behind the odd partner's candidate pass (--no-dep: without that wait -- the chain's cost alone).
Same pipelined loop as bench.py, depth 3, three interleaved runs of each arrangement.
    python tools/pairing_probe.py [steps]"""
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
DEP = "--no-dep" not in sys.argv  # the paired chain waits for the partner's candidate pass (what real pairing has to do)
orig_enq, orig_fin = P.HipPlanner._enqueue_tail, P.HipPlanner._finish
MODE = {"paired": False}


def enq(self, tk):
    if not MODE["paired"] or self.rescore != "bound":
        return orig_enq(self, tk)
    if tk.index & 1:  # the partner's chain carries this step's list: no chain of its own
        self.rescore = "none"
        try:
            return orig_enq(self, tk)
        finally:
            self.rescore = "bound"
    tk.kfirst_in = 2 * tk.kfirst_in + tk.rfirst_in  # both steps' score entries + the partner's race entries in this chain
    if DEP and tk.tchain is not None:  # the chain needs the partner's bf16 scores: behind the partner's candidate pass
        for sl in self._slots:
            if sl.owner is not None and sl.owner.index == tk.index + 1:
                tk.tchain.wait_event(sl.ev_cand)
    return orig_enq(self, tk)


def fin(self, tk):
    if MODE["paired"] and self.rescore == "bound" and (tk.index & 1) and tk.out is None:
        self.rescore = "none"
        try:
            return orig_fin(self, tk)
        finally:
            self.rescore = "bound"
    return orig_fin(self, tk)


P.HipPlanner._enqueue_tail, P.HipPlanner._finish = enq, fin
dims = synth.Dims(11, 3, 32)
cfg = types.SimpleNamespace(traj_length=32, action_samples=1024, horizon=16, discount=0.99, temperature=0.01, lmbda=0.6,
                            plan_guidance="rtg_guiding")
hist = synth.make_history(dims, 0)
hist["path_length"] = 500
for rep in range(3):
    for paired in (False, True):
        MODE["paired"] = False
        pl = P.HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision="bf16",
                          generator=torch.Generator(device="cuda").manual_seed(1), pipeline_depth=3, auto_fp32=False)
        s, a, r, h, rtg = pl.assemble_window(hist, rtg=3.0)

        def run(k):
            flight = deque()
            for _ in range(k):
                flight.append(pl._issue(capi.MODE_RTG, s, a, r, rtg, h, 0.6, pipelined=True, inputs_ready=True))
                if len(flight) > 3:
                    flight.popleft().pair()
            while flight:
                flight.popleft().pair()

        run(24)          # calibration passes, un-paired
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

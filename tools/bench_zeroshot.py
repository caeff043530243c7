#!/usr/bin/env python3
"""BASELINE config 5 shape (zeroshot_omtm goal reaching, config_hopper: T=8, H=4): E env windows per launch through
the two chained forwards of action_piid_sample (pi mask -> write inferred states -> fid mask).  Prints windows/s.
A side measurement, not the contract bench (bench.py stays on config 2).  Usage: python tools/bench_zeroshot.py [E] [steps]"""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m3pc_amd import synth  # noqa: E402
from m3pc_amd.planner import HipPlanner  # noqa: E402


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    S, A = synth.ENV_DIMS["hopper"]
    dims = synth.Dims(S, A, 8)
    cfg = types.SimpleNamespace(traj_length=8, action_samples=1, horizon=4, discount=0.99, temperature=1.0, lmbda=0.6,
                                plan_guidance="rtg_guiding", index_jump=4)
    p = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, max_batch=E)
    hists = []
    for e in range(E):
        h = synth.make_history(dims, e)
        h["path_length"] = 100 + (7 * e) % 800
        hists.append(h)
    for _ in range(5):
        p.action_piid_sample_batch(hists, eval=True, rtg=2.5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = p.action_piid_sample_batch(hists, eval=True, rtg=2.5)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"E={E} windows per call (host window assembly included): {1e3 * dt:.3f} ms/call, {E / dt:.0f} windows/s; out {tuple(out.shape)}")


if __name__ == "__main__":
    main()

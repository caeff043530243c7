#!/usr/bin/env python3
"""Host-side enqueue time of one plan step (no device sync inside the loop) next to the device time -- shows how far
the CPU runs ahead of the GPU (a profiling aid, not part of the product)."""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m3pc_amd import capi, synth  # noqa: E402
from m3pc_amd.planner import HipPlanner  # noqa: E402


def main():
    S, A = synth.ENV_DIMS["hopper"]
    dims = synth.Dims(S, A, 32)
    cfg = types.SimpleNamespace(traj_length=32, action_samples=1024, horizon=16, discount=0.99, temperature=0.01, lmbda=0.6,
                                plan_guidance="rtg_guiding")
    p = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision="bf16")
    hist = synth.make_history(dims, 0)
    hist["path_length"] = 500
    s, a, r, h, rtg = p.assemble_window(hist, rtg=3.0)
    for _ in range(20):
        p._guide(capi.MODE_RTG, s, a, r, rtg, h, 0.6)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        p._guide(capi.MODE_RTG, s, a, r, rtg, h, 0.6)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, device-complete {1e3 * (t2 - t0) / n:.3f} ms/step")
    # from an empty queue (no back-pressure): host cost of single steps
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p._guide(capi.MODE_RTG, s, a, r, rtg, h, 0.6)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"  single step: host {1e3 * (t1 - t0):.3f} ms, until complete {1e3 * (t2 - t0):.3f} ms")
    # split of the host cost
    import cProfile, pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(20):
        p._guide(capi.MODE_RTG, s, a, r, rtg, h, 0.6)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Does one plan step replay faster as a captured HIP graph?  (measurement aid, not part of the product)
Usage on the GPU box:  python tools/graph_probe.py [steps]"""
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
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    S, A = synth.ENV_DIMS["hopper"]
    T, H, N = 32, 16, 1024
    dims = synth.Dims(S, A, T)
    cfg = types.SimpleNamespace(traj_length=T, action_samples=N, horizon=H, discount=0.99, temperature=0.01, lmbda=0.6,
                                plan_guidance="rtg_guiding")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)
    planner = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, None, None,
                         precision="bf16", rescore_topk=16, device=0, generator=gen)
    hist = synth.make_history(dims, 0)
    hist["path_length"] = 500
    states, actions, rewards, h, rtg = planner.assemble_window(hist, rtg=3.0)

    def step():
        return planner._guide(capi.MODE_RTG, states, actions, rewards, rtg, h, 0.6)

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    for _ in range(30):
        step()
    print("eager  %.4f ms/step" % timed(step, steps), flush=True)
    g = torch.cuda.CUDAGraph()
    g.register_generator_state(gen)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = step()
    g.replay()
    torch.cuda.synchronize()
    print("graph  %.4f ms/step" % timed(g.replay, steps), flush=True)
    print("eager  %.4f ms/step" % timed(step, steps), flush=True)
    print("graph  %.4f ms/step" % timed(g.replay, steps), flush=True)
    print("sample action", [round(float(v), 4) for v in out[0].flatten()[:3]])


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Host-side cost of the library calls of one plan step: wall time of the call itself (it only enqueues), measured with the device
idle-synchronised before each batch of calls so that no queue back-pressure is included (a profiling aid, not part of the product)."""
import os, sys, time, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m3pc_amd import capi, synth  # noqa: E402
from m3pc_amd.planner import HipPlanner  # noqa: E402

dims = synth.Dims(11, 3, 32)
cfg = types.SimpleNamespace(traj_length=32, action_samples=1024, horizon=16, discount=0.99, temperature=0.01, lmbda=0.6, plan_guidance="rtg_guiding")
p = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision="bf16")
hist = synth.make_history(dims, 0); hist["path_length"] = 500
s, a, r, h, rtg = p.assemble_window(hist, rtg=3.0)
for _ in range(10):
    p._guide(capi.MODE_RTG, s, a, r, rtg, h, 0.6)
torch.cuda.synchronize()
hd = p.handle
eps = torch.randn((1024, 32, 3), device="cuda")
top = torch.arange(8, dtype=torch.int32, device="cuda")
out = torch.empty(8, device="cuda")
def timeit(name, fn, n=20):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    ts.sort()
    print(f"{name:28s} host {1e6 * ts[len(ts)//2][0]:7.1f} us   until complete {1e6 * sorted(t[1] for t in ts)[len(ts)//2]:7.1f} us")
timeit("policy_pass", lambda: hd.policy_pass(capi.MODE_RTG, s, a, r, h, rtg, slot=0))
timeit("candidate_pass bf16", lambda: hd.candidate_pass(capi.MODE_RTG, s, a, r, eps, h, 0.6, 0.99, 1024, precision=capi.PREC_BF16, slot=0))
timeit("rescore (8)", lambda: hd.rescore(capi.MODE_RTG, s, a, r, eps, top, h, rtg, 0.6, 0.99, 1024, slot=0, out=out, want_actions=False))
timeit("torch.randn eps", lambda: torch.randn((1024, 1, 32, 1, 3), device="cuda"))
timeit("whole serial step", lambda: p._guide(capi.MODE_RTG, s, a, r, rtg, h, 0.6))

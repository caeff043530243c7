"""Serial vs pipelined plan-step rate on BASELINE config 2 (run on the GPU box): python tools/pipeline_probe.py [steps]"""
import os
import sys
import time
import types
from collections import deque

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3pc_amd import capi, synth  # noqa: E402
from m3pc_amd.planner import HipPlanner  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
DEPTHS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3, 0, 2]
dims = synth.Dims(11, 3, 32)
cfg = types.SimpleNamespace(traj_length=32, action_samples=1024, horizon=16, discount=0.99, temperature=0.01, lmbda=0.6,
                            plan_guidance="rtg_guiding")
KW = eval(sys.argv[3]) if len(sys.argv) > 3 else {}
print("planner kw", KW)
p = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision="bf16",
               generator=torch.Generator(device="cuda").manual_seed(1), **KW)
hist = synth.make_history(dims, 0)
hist["path_length"] = 500
s, a, r, h, rtg = p.assemble_window(hist, rtg=3.0)


def run(depth, k):
    fl = deque()
    for _ in range(k):
        if depth == 0:
            p._guide(capi.MODE_RTG, s, a, r, rtg, h, 0.6)
            continue
        fl.append(p._issue(capi.MODE_RTG, s, a, r, rtg, h, 0.6, pipelined=True, inputs_ready=True))
        if len(fl) > depth:
            fl.popleft().pair()
    while fl:
        fl.popleft().pair()


for depth in DEPTHS:
    run(depth, 20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(depth, K)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"depth {depth}: {1e3 * dt:.4f} ms/step = {1 / dt:.1f} steps/s   n_rescored {p.last.get('n_rescored')} n_first {p.last.get('n_first')} "
          f"delta {(p.last.get('delta') or 0):.3f} grown {p.delta_grown}", flush=True)
# host cost of issuing alone
torch.cuda.synchronize()
t0 = time.perf_counter()
fl = deque()
for _ in range(3):
    fl.append(p._issue(capi.MODE_RTG, s, a, r, rtg, h, 0.6, pipelined=True, inputs_ready=True))
t1 = time.perf_counter()
print(f"host issue time per step (3 in a row): {1e3 * (t1 - t0) / 3:.3f} ms")
while fl:
    fl.popleft().pair()

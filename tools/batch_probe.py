"""action_sample_batch on E different windows: per-window re-score statistics and call time (run on the GPU box).
    python tools/batch_probe.py [E] [reps] ["dict(planner kw)"]"""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3pc_amd import synth  # noqa: E402
from m3pc_amd.planner import HipPlanner  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 8
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 12
KW = eval(sys.argv[3]) if len(sys.argv) > 3 else {}
LOCK = eval(sys.argv[4]) if len(sys.argv) > 4 else False  # False | True | "onepass"
if LOCK:
    KW = dict(max_batch=E, max_windows=E, **KW)
dims = synth.Dims(11, 3, 32)
cfg = types.SimpleNamespace(traj_length=32, action_samples=1024, horizon=16, discount=0.99, temperature=0.01, lmbda=0.6,
                            plan_guidance="rtg_guiding")
p = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision="bf16",
               generator=torch.Generator(device="cuda").manual_seed(1), **KW)
hs = []
for i in range(E):
    hi = synth.make_history(dims, i)
    hi["path_length"] = 500
    hs.append(hi)
for _ in range(3):
    p.action_sample_batch(hs, eval=True, rtg=3.0, lockstep=LOCK)
torch.cuda.synchronize()
for rep in range(REPS):
    t0 = time.perf_counter()
    p.action_sample_batch(hs, eval=True, rtg=3.0, lockstep=LOCK)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    w = p.last["windows"]
    if LOCK:
        print(f"call {rep}: {1e3 * dt:7.3f} ms = {E / dt:6.1f} steps/s  n_rescored {[i['n_rescored'] for i in w]}")
        continue
    print(f"call {rep}: {1e3 * dt:7.3f} ms = {E / dt:6.1f} steps/s  n_rescored {[i['n_rescored'] for i in w]} n_first {[i['n_first'] for i in w]} "
          f"in_window {[i['n_in_window'] for i in w]} delta {[round(i['delta'], 2) for i in w]} sat {[int(i['saturated']) for i in w]}")
print("delta_grown", p.delta_grown)

"""The reference's shipped planning config (finetune_omtm/config.yaml:5,77-79: N=625, H=4, T=8) and the zero-shot B=1 call,
closed loop, for a kernel trace (run on the GPU box):  python tools/small_step.py [bf16|fp32|zeroshot] [calls]"""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3pc_amd import synth  # noqa: E402
from m3pc_amd.planner import HipPlanner  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "bf16"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 30
d8 = synth.Dims(11, 3, 8)
c8 = types.SimpleNamespace(traj_length=8, action_samples=625, horizon=4, discount=0.99, temperature=0.01, lmbda=0.6,
                           plan_guidance="rtg_guiding")
h8 = synth.make_history(d8, 0)
h8["path_length"] = 500
p8 = HipPlanner(c8, synth.make_state_dict(d8, 0), synth.make_tokenizer_stats(d8, 0), None, precision="fp32" if what != "bf16" else "bf16",
                generator=torch.Generator(device="cuda").manual_seed(1))
f = (lambda: p8.action_piid_sample(h8, eval=True, rtg=2.5).cpu()) if what == "zeroshot" else \
    (lambda: p8.action_sample(h8, plan=True, eval=True, rtg=3.0).cpu())
for _ in range(5):
    f()
ts = []
for _ in range(calls):
    t0 = time.perf_counter()
    f()
    ts.append(1e3 * (time.perf_counter() - t0))
ts.sort()
print(f"{what}: p50 {ts[len(ts) // 2]:.4f} ms  min {ts[0]:.4f}  max {ts[-1]:.4f}")

"""Same-box price of certificate settings on the reference's shipped config (N=625, H=4, T=8): closed loop per call and 8
environments pipelined.  python tools/shipped_ab.py '{"calibration_factor": 1.6}' '{"calibration_factor": 2.1}' ..."""
import json
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3pc_amd import synth  # noqa: E402
from m3pc_amd.planner import HipPlanner  # noqa: E402

d8 = synth.Dims(11, 3, 8)
c8 = types.SimpleNamespace(traj_length=8, action_samples=625, horizon=4, discount=0.99, temperature=0.01, lmbda=0.6,
                           plan_guidance="rtg_guiding")
hists = []
for i in range(8):
    h = synth.make_history(d8, i)
    h["path_length"] = 500
    hists.append(h)
for rep in range(3):
    for arg in sys.argv[1:]:
        kw = json.loads(arg)
        p = HipPlanner(c8, synth.make_state_dict(d8, 0), synth.make_tokenizer_stats(d8, 0), None, precision="bf16",
                       generator=torch.Generator(device="cuda").manual_seed(1), **kw)
        for _ in range(24):
            p.action_sample(hists[0], plan=True, eval=True, rtg=3.0)
        ts, nr = [], []
        for _ in range(40):
            t0 = time.perf_counter()
            p.action_sample(hists[0], plan=True, eval=True, rtg=3.0).cpu()
            ts.append(1e3 * (time.perf_counter() - t0))
            nr.append(p.last["n_rescored"] + p.last["n_race"])
        ts.sort()
        for _ in range(3):
            p.action_sample_batch(hists, eval=True, rtg=3.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            p.action_sample_batch(hists, eval=True, rtg=3.0)
        torch.cuda.synchronize()
        e8 = 1e3 * (time.perf_counter() - t0) / 80
        print(f"{arg}: closed loop p50 {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f}  re-scored mean {sum(nr) / len(nr):.1f} max {max(nr)}  delta {p.last['delta']:.2f}"
              f"  |  8 environments pipelined {e8:.3f} ms per step", flush=True)
        p.handle.close()

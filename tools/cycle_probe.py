"""Which objects of a pipelined plan step only the cyclic collector frees (reference cycles): python tools/cycle_probe.py"""
import gc
import os
import sys
import types
from collections import Counter, deque

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from m3pc_amd import capi, synth  # noqa: E402
from m3pc_amd.planner import HipPlanner, PlanTicket  # noqa: E402

dims = synth.Dims(11, 3, 32)
cfg = types.SimpleNamespace(traj_length=32, action_samples=1024, horizon=16, discount=0.99, temperature=0.01, lmbda=0.6, plan_guidance="rtg_guiding")
p = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision="bf16", device=0,
               generator=torch.Generator(device="cuda").manual_seed(1), pipeline_depth=3)
hist = synth.make_history(dims, 0)
hist["path_length"] = 500
s_, a_, r_, h, rtg = p.assemble_window(hist, rtg=3.0)


def run(k):
    flight = deque()
    for _ in range(k):
        flight.append(p._issue(capi.MODE_RTG, s_, a_, r_, rtg, h, 0.6, pipelined=True, inputs_ready=True))
        if len(flight) > 3:
            flight.popleft().pair()
    while flight:
        flight.popleft().pair()


run(30)
gc.collect()
gc.set_debug(gc.DEBUG_SAVEALL)
n0 = len(gc.get_objects())
run(20)
n1 = len(gc.get_objects())
found = gc.collect()
print("tracked objects before / after 20 steps:", n0, n1, " unreachable found by the collector:", found)
print(Counter(type(o).__name__ for o in gc.garbage).most_common(12))
tks = [o for o in gc.garbage if isinstance(o, PlanTicket)]
print("tickets in cycles:", len(tks))
if tks:
    tk = tks[0]
    for r in gc.get_referrers(tk):
        if r is not gc.garbage and not isinstance(r, types.FrameType):
            print("  referrer:", type(r).__name__, (list(r.keys())[:8] if isinstance(r, dict) else ""))

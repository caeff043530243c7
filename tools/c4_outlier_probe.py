"""Why bench.py's c4_shard leg is bimodal (4.1-4.5 ms on most runs, 5.5-5.7 on some): the leg in bench's order (behind c5 and c3) and
alone, with what the certificates asked for in the timed steps.   python tools/c4_outlier_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from m3pc_amd import planner as _pl  # noqa: E402

seen = []
_orig_close = None


def leg():
    # plan_leg with the planner's history kept: wrap HipPlanner.__init__ to remember the instance
    made = []
    orig = _pl.HipPlanner.__init__

    def init(self, *a, **k):
        orig(self, *a, **k)
        made.append(self)

    _pl.HipPlanner.__init__ = init
    try:
        t0 = time.perf_counter()
        r = bench.plan_leg(0, "halfcheetah", "rtg_guiding", 2048, 32, 64, steps=20, settle=12)
        wall = time.perf_counter() - t0
    finally:
        _pl.HipPlanner.__init__ = orig
    p = made[-1]
    h = sorted(p._hist.items())
    tail = [(i, v[3]) for i, v in h[-24:]]
    return r["ms_per_step"], max(v for _, v in tail), sum(v for _, v in tail) / len(tail), round(float(p._delta), 3), round(wall, 2), p.fp32_fallback


for rep in range(3):
    print("alone      ", leg(), flush=True)
for rep in range(2):
    bench.goal_leg(0, 8192, steps=20, fp32_steps=4)
    bench.plan_leg(0, "walker2d", "critic_lambda_guiding", 4096, 16, 32, steps=20, settle=12)
    print("behind c5+c3", leg(), flush=True)
    print("again       ", leg(), flush=True)

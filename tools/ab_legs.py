"""Same-box timing of bench.py's secondary legs (config 3, a config-4 shard, config 5) for library A/Bs:
M3PC_LIB=... python tools/ab_legs.py [c3] [c4_shard] [c4_full] [c5] [c2_t16]   -> ms per step / call, three repetitions each"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import bench  # noqa: E402

which = sys.argv[1:] or ["c3", "c4_shard"]
for rep in range(3):
    out = []
    if "c3" in which:
        out.append("c3 %.4f" % bench.plan_leg(0, "walker2d", "critic_lambda_guiding", 4096, 16, 32, steps=20, settle=12)["ms_per_step"])
    if "c4_shard" in which:
        out.append("c4_shard %.4f" % bench.plan_leg(0, "halfcheetah", "rtg_guiding", 2048, 32, 64, steps=20, settle=12)["ms_per_step"])
    if "c4_full" in which:
        out.append("c4_full %.4f" % bench.c4_full_leg(0)["ms_per_step"])
    if "c5" in which:
        out.append("c5 %.4f" % bench.goal_leg(0, 8192, precisions=("bf16",))["bf16"]["ms_per_call"])
    if "c2_t16" in which:
        out.append("c2_t16 %.4f" % bench.plan_leg(0, "hopper", "rtg_guiding", 1024, 16, 16, steps=20, settle=12)["ms_per_step"])
    print(os.environ.get("M3PC_LIB", "product"), " ".join(out), flush=True)

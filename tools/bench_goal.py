"""BASELINE config 5 leg alone: windows/s of the pruned zero-shot path at E windows per call (bench.py goal_leg)."""
import json
import sys

sys.path.insert(0, ".")
import bench  # noqa: E402

if __name__ == "__main__":
    for E in [int(a) for a in sys.argv[1:]] or [8192]:
        print(json.dumps(bench.goal_leg(0, E)))

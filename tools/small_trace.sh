#!/bin/bash
# kernel trace of the small-N closed-loop calls (run on the GPU box): tools/small_trace.sh <bf16|fp32|zeroshot>
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box}"
W=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/small_step.py $W 30
rocprofv3 --kernel-trace -d gpurun_out/small_${W} -o kt --output-format csv -- python3 tools/small_step.py $W 10 > gpurun_out/small_${W}.log 2>&1
python3 - <<P
import csv, glob
f = glob.glob("gpurun_out/small_${W}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last call: from the last-but-one D2H-preceding kernel... take the last 1/15 of the rows by the final select/actor kernel
marks = [i for i, r in enumerate(rows) if ("select_kernel" in r["Kernel_Name"] and "topk" not in r["Kernel_Name"]) or ("$W" == "zeroshot" and "actor_head" in r["Kernel_Name"])]
step = ${W@Q} == "zeroshot" and 2 or 1
a, b = marks[-1 - step] + 1, marks[-1] + 1
t0 = int(rows[a]["Start_Timestamp"]); prev = None; out = []
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append("%8.1f %6.1f %6.1f q%s %s grid=%s" % ((s - t0) / 1e3, (e - s) / 1e3, ((s - prev) / 1e3 if prev else 0), r.get("Queue_Id"), r["Kernel_Name"].replace("m3pc::", "").replace("void ", "")[:64], r["Grid_Size_X"]))
    prev = e
out.append("# span %.1f us, %d kernels, busy %.1f us" % ((int(rows[b - 1]["End_Timestamp"]) - t0) / 1e3, b - a, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a:b]) / 1e3))
open("gpurun_out/small_${W}_timeline.txt", "w").write("\n".join(out) + "\n")
P
rm -rf gpurun_out/small_${W}

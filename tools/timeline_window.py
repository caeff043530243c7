#!/usr/bin/env python3
"""Kernel timeline of a time window of a rocprofv3 --kernel-trace CSV, all queues interleaved (profiling aid):
    python3 tools/timeline_window.py <trace dir> <out.txt> [n_steps_to_show=2]
The window starts at the select_kernel in the middle of the trace and covers the next n steps."""
import csv
import glob
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    f = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "::select_kernel" in r["Kernel_Name"] or "merge_select_kernel" in r["Kernel_Name"]]
    k = len(ends) // 2
    a, b = ends[k] + 1, ends[k + n] + 1
    step = rows[a:b]
    t0 = int(step[0]["Start_Timestamp"])
    lines = []
    qs = sorted({r.get("Queue_Id", "?") for r in step})
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("m3pc::", "").replace("void ", "").replace("(anonymous namespace)::", "")[:60]
        col = qs.index(r.get("Queue_Id", "?"))
        lines.append("%8.1f %7.1f  %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, " " * (col * 28), name))
    span = (int(step[-1]["End_Timestamp"]) - t0) / 1e3
    lines.append("# window %.1f us over %d select kernels, queues %s" % (span, n, qs))
    open(out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()

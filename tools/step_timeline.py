#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace CSV of `bench.py` into the kernel timeline of ONE plan step
(start offset, duration, gap to the previous kernel, name, grid) -- a profiling aid, not part of the product.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o kt --output-format csv -- python3 bench.py ...
    python3 tools/step_timeline.py gpurun_out/prof_kt [out.txt]
"""
import csv
import glob
import sys


def main():
    src = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    f = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a step ends with its select: a launch of its own, or (round 5) the second half of merge_select_kernel (not topk_select_kernel)
    ends = [i for i, r in enumerate(rows) if "::select_kernel" in r["Kernel_Name"] or "merge_select_kernel" in r["Kernel_Name"]]
    k = len(ends) // 5  # bench.py runs warmup, timed, latency and instrumented passes: this one is inside the timed region
    a, b = ends[k] + 1, ends[k + 1] + 1
    step = rows[a:b]
    t0 = int(step[0]["Start_Timestamp"])
    prev = None
    lines = []
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev) if prev else 0
        prev = e
        name = r["Kernel_Name"].replace("m3pc::", "").replace("void ", "")[:70]
        lines.append("%8.1f %7.1f %6.1f  q%s %s grid=%s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, r.get("Queue_Id", "?"), name,
                                                         r.get("Grid_Size_X", "")))
    span = (int(step[-1]["End_Timestamp"]) - t0) / 1e3
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step) / 1e3
    lines.append("# step span %.1f us, %d kernels, sum of durations %.1f us" % (span, len(step), busy))
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Turn the rocprofv3 outputs that a gpurun call merged into gpurun_out/ into the committed summaries under
profiles/:  <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats),  <tag>_pmc_traffic.json (HBM bytes per
launch per kernel from the FETCH_SIZE and WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes:
counters are KiB, FETCH_SIZE is doubled on gfx950) and pmc_traffic.json (what bench.py reads for `traffic`).

    python tools/summarize_profiles.py r01 gpurun_out/r01_trace gpurun_out/r01_fetch gpurun_out/r01_write
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(pattern):
    hits = glob.glob(pattern, recursive=True)
    if not hits:
        raise SystemExit(f"nothing matches {pattern}")
    return hits[0]


def agg(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]][0] += 1
        d[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return d


def main():
    tag, trace, fetch, write = sys.argv[1:5]
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    shutil.copy(one(os.path.join(trace, "**", "*kernel_stats.csv")), os.path.join(prof, f"{tag}_kernel_stats.csv"))
    F, W = agg(one(os.path.join(fetch, "**", "*counter_collection.csv"))), agg(one(os.path.join(write, "**", "*counter_collection.csv")))
    out = {}
    for k, (n, f) in F.items():
        if not k.startswith(("void m3pc", "m3pc::", "_ZN4m3pc")):
            continue
        wn, w = W.get(k, [0, 0.0])
        f_kib, w_kib = f / n, (w / wn if wn else 0.0)
        out[k] = {"launches_profiled": n, "FETCH_SIZE_KiB_per_launch": round(f_kib, 1), "WRITE_SIZE_KiB_per_launch": round(w_kib, 1),
                  "hbm_bytes_per_launch": int((2.0 * f_kib + w_kib) * 1024)}
    doc = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 10 --warmup 3`",
           "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads)",
           "kernels": out}
    for name in (f"{tag}_pmc_traffic.json", "pmc_traffic.json"):
        json.dump(doc, open(os.path.join(prof, name), "w"), indent=1, sort_keys=True)
    print("wrote", len(out), "kernels")
    if len(sys.argv) > 5:
        sq_summary(tag, sys.argv[5], prof)


def sq_summary(tag, sqdir, prof, clock_ghz=1.75, simds=1024):
    """SQ pass (own run): per kernel the wave-cycle split and the matrix-pipe utilisation.
    SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; utilisation = busy / (SIMDs * duration * clock), with the
    1.75 GHz the GEMM holds under load (in-kernel s_memtime / s_memrealtime probe, tools/gemm_bench.py 26)."""
    rows = list(csv.DictReader(open(one(os.path.join(sqdir, "**", "*counter_collection.csv")))))
    cnt = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    dur = collections.defaultdict(float)
    seen = set()
    for r in rows:
        k = r["Kernel_Name"]
        if not k.startswith(("void m3pc", "m3pc::", "_ZN4m3pc")):
            continue
        cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            n[k] += 1
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out = {}
    for k, c in cnt.items():
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        d_ns = dur[k] / n[k]
        e = {"launches_profiled": n[k], "avg_duration_us_under_pmc": round(d_ns / 1e3, 1)}
        for name in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            e[name + "_frac_of_wave_cycles"] = round(c.get(name, 0.0) / wc, 3) if wc else None
        e["SQ_LDS_BANK_CONFLICT_per_launch"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / n[k]
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n[k]
        e["mfma_busy_frac"] = round(busy / (simds * d_ns * clock_ghz), 3) if d_ns else None
        out[k] = e
    doc = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                     "SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 on `python3 bench.py --steps 10 --warmup 3`",
           "notes": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * duration * 1.75 GHz); durations are those of the PMC "
                    "run (counter collection serialises and slows kernels; use <tag>_kernel_stats.csv for timings)",
           "kernels": out}
    json.dump(doc, open(os.path.join(prof, f"{tag}_sq_counters.json"), "w"), indent=1, sort_keys=True)
    print("wrote SQ summary for", len(out), "kernels")


if __name__ == "__main__":
    main()

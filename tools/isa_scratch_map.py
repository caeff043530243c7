#!/usr/bin/env python3
"""Where a kernel's scratch (spill) instructions sit relative to its MFMAs: hipcc -save-temps .s -> per kernel the number of
scratch ops between consecutive MFMAs, and the loop labels.   python tools/isa_scratch_map.py file.s [name filter]"""
import bisect
import re
import sys

L = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else "block_fused_kernel"
starts = [i for i, l in enumerate(L) if re.match(r"^_Z\w+:", l) and flt in l]
for i in starts:
    end = next(j for j in range(i, len(L)) if "s_endpgm" in L[j])
    body = L[i:end]
    mf = [k for k, x in enumerate(body) if "v_mfma" in x]
    sc = [k for k, x in enumerate(body) if "scratch_" in x]
    print(L[i].split(":")[0][-40:], "lines", len(body), "mfma", len(mf), "scratch ops", len(sc))
    b = {}
    for k in sc:
        b[bisect.bisect(mf, k)] = b.get(bisect.bisect(mf, k), 0) + 1
    print("  scratch ops after MFMA #:", sorted(b.items()))
    lab = [(k, x) for k, x in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", x)]
    br = [(k, x.strip()) for k, x in enumerate(body) if "s_cbranch" in x or "s_branch" in x]
    print("  labels at MFMA #:", [(x[:-1], bisect.bisect(mf, k)) for k, x in lab])
    print("  branches at MFMA #:", [(bisect.bisect(mf, k), x) for k, x in br])

#!/usr/bin/env python3
"""kv_fused_kernel alone: time per launch and the in-kernel phase stamps of one workgroup (a profiling aid)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from m3pc_amd import capi  # noqa: E402
import test_block_fused_gpu as T  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    lib = capi.load_library()
    dev = torch.device("cuda")
    D = 512
    Le, kept, off = 49, (17, 32), (0, 17)
    rn = lambda *s: torch.randn(*s, device=dev)
    Z = rn(n * Le, D).to(torch.bfloat16)
    We = [(rn(D, D) / D ** 0.5).to(torch.bfloat16) for _ in range(2)]
    Wkv = (rn(2 * D, D) / D ** 0.5).to(torch.bfloat16)
    rowtab = [0.5 * rn(kept[k], D) for k in range(2)]
    g, b, bkv = 1 + 0.1 * rn(D), 0.1 * rn(D), 0.1 * rn(2 * D)
    stamps = torch.zeros(64, dtype=torch.int64, device=dev)
    for _ in range(3):
        T._kv_call(lib, Z, n, Le, kept, off, We, Wkv, rowtab, g, b, bkv, stamps)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(10):
        e0.record()
        T._kv_call(lib, Z, n, Le, kept, off, We, Wkv, rowtab, g, b, bkv, stamps)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("n=%d: launch (incl. the two pack launches) %.1f us min" % (n, 1e3 * min(ts)))
    s = stamps.cpu().view(4, 16)
    names = ["prologue", "embed", "LN", "K", "store K + V + store V"]
    for w in range(4):
        d = [int(s[w, k + 1] - s[w, k]) for k in range(5)]
        print("wave %d: " % w + ", ".join("%s %d" % (nm, v) for nm, v in zip(names, d)) + "  total %d (x10 ns)" % sum(d))


if __name__ == "__main__":
    main()

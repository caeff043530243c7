"""(Timing variants other than the product kernels live in the lab build: `python -m m3pc_amd.build --lab`, then
run with M3PC_LIB=m3pc_amd/libm3pc_hip_lab.so.)
Time the fused layer tail (block_fused.hip) alone on the plan step's row counts.
usage: python tools/block_bench.py [rows ...]     variants: 0 product, 1 no DMA pieces, 2 no gelu"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from m3pc_amd import capi  # noqa: E402
from tests.test_block_fused_gpu import D, FF, _call, make_params  # noqa: E402


def main():
    lib = capi.load_library()
    dev = torch.device("cuda")
    rows = [int(a) for a in sys.argv[1:]] or [50176, 32768, 25088]
    W, p, lnB, g = make_params(0)
    nbytes = lib.m3pc_debug_block_stream_bytes
    nbytes.restype = C.c_longlong
    sb = torch.empty(int(nbytes()), dtype=torch.uint8, device=dev)
    for M in rows:
        O = torch.randn(M, D, device=dev, generator=g).to(torch.bfloat16)
        R = torch.randn(M, D, device=dev, generator=g)
        X = torch.empty_like(R)
        H = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
        _call(lib, O, R, None, 1, W, sb, 1, p, [None] * 4, 0, 0, X, H)
        flops = 2.0 * M * (D * D + 2 * D * FF)
        stamps = torch.zeros(4, 16, dtype=torch.int64, device=dev)
        for rep in range(3):
            _call(lib, O, R, None, 1, W, sb, 0, p, [None] * 4, 0, 0, X, H, 0, stamps=stamps)
        s = stamps.cpu()
        names = ["prologue", "out-proj", "LN2", "FFN", "X store", "LN + H store"]
        for w in range(4):
            dl = [int(s[w, k + 1] - s[w, k]) for k in range(6)]
            print(f"rows {M} wave {w} clocks: " + "  ".join(f"{n} {v}" for n, v in zip(names, dl)) + f"  total {int(s[w, 6] - s[w, 0])}")
        for v in (3, 4, 0):
            stamps.zero_()
            _call(lib, O, R, None, 1, W, sb, 0, p, [None] * 4, 0, 0, X, H, v, stamps=stamps)
            s = stamps.cpu()
            print(f"rows {M} variant {v}: FFN {int(s[0, 4] - s[0, 3])} out-proj {int(s[0, 2] - s[0, 1])} total {int(s[0, 6] - s[0, 0])}"
                  + (f"  per phase kind A0 {int(s[0, 8]) // 32} A1 {int(s[0, 9]) // 32} B1 {int(s[0, 10]) // 32} B2 {int(s[0, 11]) // 32}" if v == 3 else ""))
        for variant in ():
            for xo in (True, False):
                ts = []
                for _ in range(12):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    _call(lib, O, R, None, 1, W, sb, 0, p, [None] * 4, 0, 0, X if xo else None, H, variant, sync=False)
                    b.record()
                    torch.cuda.synchronize()
                    ts.append(a.elapsed_time(b) * 1e3)
                ts.sort()
                print(f"rows {M:6d} variant {variant} xout {int(xo)}: min {ts[0]:7.1f} us  med {ts[len(ts)//2]:7.1f} us  "
                      f"{flops / ts[0] / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()

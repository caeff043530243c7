#!/usr/bin/env python3
"""(Timing variants other than the product kernels live in the lab build: `python -m m3pc_amd.build --lab`, then
run with M3PC_LIB=m3pc_amd/libm3pc_hip_lab.so.)
Time ONE bf16 GEMM shape on chosen kernel variants (kernel iteration tool, not part of the product).
Usage on the GPU box:  python tools/gemm_one.py M N K gelu res f32out variant [variant...]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m3pc_amd import capi  # noqa: E402


def main():
    M, N, K, gelu, res, f32out = (int(v) for v in sys.argv[1:7])
    variants = [int(v) for v in sys.argv[7:]] or [0]
    lib = capi.load_library()
    fn = lib.m3pc_debug_gemm
    fn.restype = C.c_int
    vp, i = C.c_void_p, C.c_int
    fn.argtypes = [i, vp, vp, vp, vp, vp, i, i, i, i, i, i, vp]
    dev = torch.device("cuda")
    torch.manual_seed(0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev) if res else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32out else torch.bfloat16)
    line = f"M={M} N={N} K={K} gelu={gelu} res={res} f32out={f32out}"
    for v in variants:
        def run():
            rc = fn(1, A.data_ptr(), W.data_ptr(), bias.data_ptr(), R.data_ptr() if res else None, out.data_ptr(), M, N, K, gelu, f32out, v, st)
            assert rc == 0, lib.m3pc_last_error()
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50.0
        line += f" | v{v}: {us:7.1f}us {2.0 * M * N * K / us / 1e6:7.1f} TF/s"
    print(line, flush=True)


if __name__ == "__main__":
    main()

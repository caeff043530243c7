#!/usr/bin/env python3
"""(Timing variants other than the product kernels live in the lab build: `python -m m3pc_amd.build --lab`, then
run with M3PC_LIB=m3pc_amd/libm3pc_hip_lab.so.)
Time the library's MFMA GEMM on the shapes of one C2 plan step (N=1024, H=16, T=32) -- a kernel
iteration tool, not part of the product.  Usage on the GPU box:  python tools/gemm_bench.py [variants...]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m3pc_amd import capi  # noqa: E402

SHAPES = [  # name, M, N, K, gelu, res, f32out
    ("enc.qkv", 50176, 1536, 512, 0, 0, 0),
    ("enc.out_proj", 50176, 512, 512, 0, 1, 1),
    ("enc.ffn1", 50176, 2048, 512, 1, 0, 0),
    ("enc.ffn2", 50176, 512, 2048, 0, 1, 1),
    ("dec.kv", 50176, 1024, 512, 0, 0, 0),
    ("dec.ffn1", 32768, 2048, 512, 1, 0, 0),
    ("dec.ffn2", 32768, 512, 2048, 0, 1, 1),
    ("dec.head1", 16384, 512, 512, 1, 0, 1),
]


def main():
    variants = [int(v) for v in sys.argv[1:]] or [0]
    lib = capi.load_library()
    fn = lib.m3pc_debug_gemm
    fn.restype = C.c_int
    vp, i = C.c_void_p, C.c_int
    fn.argtypes = [i, vp, vp, vp, vp, vp, i, i, i, i, i, i, vp]
    dev = torch.device("cuda")
    torch.manual_seed(0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    tot = {v: 0.0 for v in variants}
    for name, M, N, K, gelu, res, f32out in SHAPES:
        pad = int(os.environ.get("M3PC_DEBUG_LDPAD", "0"))  # operand row stride = K + pad elements (the library reads the same env)
        Ap = torch.zeros(M, K + pad, device=dev, dtype=torch.bfloat16)
        Wp = torch.zeros(N, K + pad, device=dev, dtype=torch.bfloat16)
        Ap[:, :K] = torch.randn(M, K, device=dev).to(torch.bfloat16)
        Wp[:, :K] = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        A, W = Ap[:, :K], Wp[:, :K]  # strided views: data_ptr() is the padded buffer's base
        bias = torch.randn(N, device=dev)
        R = torch.randn(M, N, device=dev) if res else None
        Cout = torch.empty(M, N, device=dev, dtype=torch.float32 if f32out else torch.bfloat16)
        ref = None
        line = f"{name:14s} M={M:6d} N={N:5d} K={K:5d}"
        for v in variants:
            def run():
                rc = fn(1, A.data_ptr(), W.data_ptr(), bias.data_ptr(), R.data_ptr() if res else None, Cout.data_ptr(),
                        M, N, K, gelu, f32out, v, st)
                assert rc == 0, lib.m3pc_last_error()
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100.0
            tot[v] += us
            out = Cout.float()
            if ref is None:
                sel = torch.cat([torch.arange(256), torch.arange(M - 1280, M)]).to(dev)  # head + peeled tail rows
                x = A[sel].float() @ W.float().T + bias
                if gelu:
                    x = torch.nn.functional.gelu(x)
                if res:
                    x = x + R[sel]
                ref = x
            err = float((out[sel] - ref).abs().max())
            line += f" | v{v}: {us:7.1f}us {2.0 * M * N * K / us / 1e6:7.1f} TF/s err {err:.2e}"
            if 26 <= v <= 30:  # ring debug variants carry a clock probe: shader clocks / 100-MHz ticks of one workgroup
                buf = (C.c_longlong * 2)()
                lib.m3pc_debug_clock(buf)
                if buf[1] > 0:
                    line += f" clk {100.0 * buf[0] / buf[1]:.0f}MHz wg {buf[1] / 100.0:.1f}us"
            if 37 <= v <= 42:  # gemm_big.hip phase timers of one workgroup (100-MHz ticks)
                buf = (C.c_longlong * 4)()
                lib.m3pc_debug_clock_big(buf)
                line += f" [pro {buf[0] / 100.0:.1f} loop {buf[1] / 100.0:.1f} epi {buf[2] / 100.0:.1f}us clk {100.0 * buf[3] / max(buf[1], 1):.0f}MHz]"
        print(line, flush=True)
    print("total us per variant:", {v: round(t, 1) for v, t in tot.items()})


if __name__ == "__main__":
    main()

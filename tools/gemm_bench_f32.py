#!/usr/bin/env python3
"""Time the fp32 MFMA GEMM on the few-row shapes of the policy pass (batch 1) and of the top-16 re-score -- a
kernel iteration tool, not part of the product.  Usage on the GPU box:  python tools/gemm_bench_f32.py [variants...]
variant 0 = default (split-K where the launcher picks it), 1 = no split-K workspace."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m3pc_amd import capi  # noqa: E402

SHAPES = [  # name, M, N, K, gelu, res
    ("rs.enc.qkv", 784, 1536, 512, 0, 0),
    ("rs.enc.out", 784, 512, 512, 0, 1),
    ("rs.enc.ffn1", 784, 2048, 512, 1, 0),
    ("rs.enc.ffn2", 784, 512, 2048, 0, 1),
    ("rs.dec.kv", 784, 1024, 512, 0, 0),
    ("rs.dec.ffn1", 512, 2048, 512, 1, 0),
    ("rs.dec.ffn2", 512, 512, 2048, 0, 1),
    ("pp.enc.qkv", 65, 1536, 512, 0, 0),
    ("pp.enc.out", 65, 512, 512, 0, 1),
    ("pp.enc.ffn1", 65, 2048, 512, 1, 0),
    ("pp.enc.ffn2", 65, 512, 2048, 0, 1),
    ("pp.dec.ffn1", 16, 2048, 512, 1, 0),
    ("pp.dec.ffn2", 16, 512, 2048, 0, 1),
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
    for name, M, N, K, gelu, res in SHAPES:
        A = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) / K ** 0.5
        bias = torch.randn(N, device=dev)
        R = torch.randn(M, N, device=dev) if res else None
        Cout = torch.empty(M, N, device=dev)
        x = A.double() @ W.double().T + bias.double()
        if gelu:
            x = torch.nn.functional.gelu(x)
        if res:
            x = x + R.double()
        line = f"{name:12s} M={M:4d} N={N:5d} K={K:5d}"
        for v in variants:
            def run():
                rc = fn(0, A.data_ptr(), W.data_ptr(), bias.data_ptr(), R.data_ptr() if res else None, Cout.data_ptr(),
                        M, N, K, gelu, 1, v, st)
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
            tot[v] += us
            err = float((Cout.double() - x).abs().max())
            line += f" | v{v}: {us:6.1f}us {2.0 * M * N * K / us / 1e6:6.1f} TF/s err {err:.1e}"
        print(line, flush=True)
    print("total us per variant:", {v: round(t, 1) for v, t in tot.items()})


if __name__ == "__main__":
    main()

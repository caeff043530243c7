#!/usr/bin/env python3
"""How fast does the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) run the plan step's shapes?  A yardstick for
tools/gemm_bench.py, not used by the product (plain GEMM: no bias / GELU / residual epilogue)."""
import torch

SHAPES = [("enc.qkv", 50176, 1536, 512), ("enc.out_proj", 50176, 512, 512), ("enc.ffn1", 50176, 2048, 512),
          ("enc.ffn2", 50176, 512, 2048), ("dec.kv", 50176, 1024, 512), ("dec.ffn1", 32768, 2048, 512),
          ("dec.ffn2", 32768, 512, 2048), ("big", 32768, 4096, 4096)]
dev = torch.device("cuda")
for name, M, N, K in SHAPES:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = torch.randn(N, K, device=dev).to(torch.bfloat16)
    for _ in range(3):
        C = A @ W.T
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        C = A @ W.T
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    print(f"{name:14s} M={M:6d} N={N:5d} K={K:5d}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF/s (bf16 in, bf16 out, no epilogue)")

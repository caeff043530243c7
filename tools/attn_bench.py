#!/usr/bin/env python3
"""The bf16 attention of an encoder layer alone (lab build): the pipelined kernel against the direct one, cache-cold (buffer sets
rotated so that nothing is served from the 256-MiB MALL).   M3PC_LIB=m3pc_amd/libm3pc_hip_lab.so python tools/attn_bench.py [batch]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from m3pc_amd import capi  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 512
    lib = capi.load_library()
    fn = lib.m3pc_debug_attention_bf16
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p] * 2
    nset = 10
    shapes = ((97, 0), (33, 64)) if "T64" in sys.argv else ((49, 0), (17, 32))  # (T64: the 97-row shapes of BASELINE config 4)
    for n_own, n_sh in shapes:
        L = n_own + n_sh
        sets = [(torch.randn(batch, n_own, 1536, device="cuda").to(torch.bfloat16), torch.randn(max(n_sh, 1), 1536, device="cuda").to(torch.bfloat16),
                 torch.empty(batch, L, 512, device="cuda", dtype=torch.bfloat16)) for _ in range(nset)]
        mb = batch * (n_own * 3072 + L * 1024) / 1e6
        for kernel, name in ((0, "pipelined"), (1, "direct"), (2, "pipe:loads"), (3, "pipe:math")):
            s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            ts = []
            for rep in range(4):
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(nset + 1)]
                ev[0].record()
                for i, (q, qs, o) in enumerate(sets):
                    assert fn(q.data_ptr(), qs.data_ptr() if n_sh else None, o.data_ptr(), batch, n_own, n_sh, kernel, s, None) == 0
                    ev[i + 1].record()
                torch.cuda.synchronize()
                if rep:
                    ts += [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(nset)]
            ts.sort()
            if kernel == 0 and n_own + n_sh <= 52:
                stamps = torch.zeros(16, dtype=torch.int64, device="cuda")
                q, qs, o = sets[0]
                for _ in range(2):
                    fn(q.data_ptr(), qs.data_ptr() if n_sh else None, o.data_ptr(), batch, n_own, n_sh, 0, s, stamps.data_ptr())
                torch.cuda.synchronize()
                st = stamps.cpu().tolist()
                names = ["scores", "softmax", "P V", "O -> LDS", "read back + barrier", "stores"]
                for w in (0, 1):
                    print(f"   wave {w} item 3 clocks: " + "  ".join(f"{n} {st[8 * w + i + 1] - st[8 * w + i]}" for i, n in enumerate(names))
                          + f"  total {st[8 * w + 6] - st[8 * w]}")
            print(f"batch {batch} own {n_own} shared {n_sh} {name:9s}: min {ts[0]:6.1f} us  med {ts[len(ts) // 2]:6.1f} us  "
                  f"{mb / ts[len(ts) // 2]:.2f} TB/s algorithmic ({mb:.0f} MB)", flush=True)


def dec(batch):
    lib = capi.load_library()
    fn0 = lib.m3pc_debug_attention_dec_le_bf16
    fn0.restype = C.c_int
    fn0.argtypes = [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_void_p]
    nset, nq, Lm, Le = (10, 64, 95, 97) if "T64" in sys.argv else (10, 32, 47, 49)
    fn = lambda a, b, c, d, e, n, q_, lm, kernel, s: fn0(a, b, c, d, e, n, q_, lm, Le, kernel, s)
    qtab = torch.randn(nq, 1536, device="cuda").to(torch.bfloat16)
    qkvm = torch.randn(Lm, 1536, device="cuda").to(torch.bfloat16)
    pre = torch.zeros(4 * nq * 130, device="cuda")
    sets = [(torch.randn(batch, Le, 1024, device="cuda").to(torch.bfloat16), torch.empty(batch, nq, 512, device="cuda", dtype=torch.bfloat16))
            for _ in range(nset)]
    mb = batch * (Le * 2048 + nq * 1024) / 1e6
    for kernel, name in ((0, "pipelined"), (1, "direct"), (2, "pipe:loads"), (3, "pipe:math")):
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        ts = []
        for rep in range(4):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(nset + 1)]
            ev[0].record()
            for i, (kv, o) in enumerate(sets):
                assert fn(qtab.data_ptr(), qkvm.data_ptr(), kv.data_ptr(), o.data_ptr(), pre.data_ptr(), batch, nq, Lm, kernel, s) == 0
                ev[i + 1].record()
            torch.cuda.synchronize()
            if rep:
                ts += [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(nset)]
        ts.sort()
        print(f"decoder batch {batch} {name:9s}: min {ts[0]:6.1f} us  med {ts[len(ts) // 2]:6.1f} us  {mb / ts[len(ts) // 2]:.2f} TB/s algorithmic "
              f"({mb:.0f} MB; incl. the prestats launch)", flush=True)


if __name__ == "__main__":
    main()
    dec(int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 512)

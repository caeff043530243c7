#!/usr/bin/env python3
"""would the batch-1 fp32 policy pass gain from living on ONE XCD (coherent L2 => cheap stage barriers
for a persistent per-layer kernel)?  The experiment here needs no new kernel: a stream created with
hipExtStreamCreateWithCUMask confines the EXISTING policy-pass launches to the CUs of one XCD; if those launches -- whose work
would be the stages of the persistent kernel -- already take longer there than the kill criterion allows (120 us for the pass),
no barrier saving can rescue it.  Step 1 maps CU-mask bits to XCDs with a probe kernel (HW_REG_XCC_ID), step 2 times
`mtm_sampling` (= the policy pass + sampling) on the whole chip and on one XCD, step 3 the same for single GEMM launches.
A profiling aid, not part of the product."""
import ctypes as C
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m3pc_amd import capi, synth  # noqa: E402
from m3pc_amd.planner import HipPlanner  # noqa: E402


def masked_stream(hip, words):
    st = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), len(words), arr)
    assert rc == 0, rc
    return st


def probe(lib, stream_ptr, n=512):
    out = torch.full((2 * n,), -1, dtype=torch.int32, device="cuda")
    lib.m3pc_debug_xcc_probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    assert lib.m3pc_debug_xcc_probe(out.data_ptr(), n, stream_ptr) == 0
    torch.cuda.synchronize()
    o = out.cpu().view(n, 2)
    return sorted(set(int(x) for x in o[:, 0])), len(set((int(a), int(b)) for a, b in o))


def main():
    torch.zeros(1, device="cuda")
    hip = C.CDLL("libamdhip64.so")
    lib = capi.load_library()
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    print("CUs:", n_cu)
    # step 1: which XCDs does a mask of 32 consecutive bits / of every 8th bit reach?
    cands = {}
    for k in range(8):
        w = [0] * 8
        w[k] = 0xFFFFFFFF
        cands["bits %d..%d" % (32 * k, 32 * k + 31)] = w
    for k in range(8):
        bits = [i for i in range(256) if i % 8 == k]
        w = [0] * 8
        for b in bits:
            w[b // 32] |= 1 << (b % 32)
        cands["bits = %d mod 8" % k] = w
    one_xcd = None
    for name, w in cands.items():
        st = masked_stream(hip, w)
        xcds, nslots = probe(lib, st)
        print("mask %-16s -> XCDs %s (%d distinct (xcc, hw_id) values)" % (name, xcds, nslots))
        if len(xcds) == 1 and one_xcd is None:
            one_xcd = (name, w)
        hip.hipStreamDestroy(st)
    if one_xcd is None:
        print("no probed mask stays inside one XCD")
        return
    print("one-XCD mask:", one_xcd[0])
    # step 2: the policy pass on the whole chip vs on that XCD
    S, A = synth.ENV_DIMS["hopper"]
    dims = synth.Dims(S, A, 32)
    cfg = types.SimpleNamespace(traj_length=32, action_samples=1024, horizon=16, discount=0.99, temperature=0.01, lmbda=0.6,
                                plan_guidance="rtg_guiding")
    p = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision="bf16")
    hist = synth.make_history(dims, 0)
    hist["path_length"] = 500
    s, a, r, h, rtg = p.assemble_window(hist, rtg=3.0)
    traj = {"states": s[None], "actions": a[None], "rewards": r[None], "_rtg": rtg}

    def time_pass(stream, n=50):
        with torch.cuda.stream(stream):
            for _ in range(5):
                p.mtm_sampling(traj, h)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                p.mtm_sampling(traj, h)
            e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / n

    full = time_pass(torch.cuda.current_stream())
    st = masked_stream(hip, one_xcd[1])
    ext = torch.cuda.ExternalStream(st.value)
    one = time_pass(ext)
    print("policy pass (mtm_sampling, %d launches): whole chip %.1f us, one XCD %.1f us" % (30, full, one))
    # step 3: single GEMM launches of the pass (M = 65 rows), back to back
    fn = lib.m3pc_debug_gemm
    fn.restype = C.c_int
    vp, i = C.c_void_p, C.c_int
    fn.argtypes = [i, vp, vp, vp, vp, vp, i, i, i, i, i, i, vp]
    for (N, K) in ((1536, 512), (512, 512), (2048, 512), (512, 2048)):
        M = 65
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda")
        bias = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda")
        res = {}
        for nm, strm, sp in (("chip", torch.cuda.current_stream(), C.c_void_p(torch.cuda.current_stream().cuda_stream)), ("xcd", ext, st)):
            with torch.cuda.stream(strm):
                for _ in range(5):
                    assert fn(0, x.data_ptr(), w.data_ptr(), bias.data_ptr(), None, out.data_ptr(), M, N, K, 0, 1, 0, sp) == 0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    fn(0, x.data_ptr(), w.data_ptr(), bias.data_ptr(), None, out.data_ptr(), M, N, K, 0, 1, 0, sp)
                e1.record()
            torch.cuda.synchronize()
            res[nm] = 1e3 * e0.elapsed_time(e1) / 50
        print("  GEMM %d x %d x %d fp32: whole chip %.1f us, one XCD %.1f us per launch" % (M, N, K, res["chip"], res["xcd"]))

if __name__ == "__main__":
    main()

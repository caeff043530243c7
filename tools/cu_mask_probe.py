#!/usr/bin/env python3
"""Round 6: do the fp32 chains of a pipelined plan step (policy pass, re-score + select: ~15 % of the CU time for < 1 % of the
FLOPs, priced by CU ACQUISITION -- their launches only get CUs at the fused tiles' boundaries, DESIGN section 4 "Round 5") run
better on CUs of their own?  hipExtStreamCreateWithCUMask streams: the two chain streams confined to R CUs per XCD, the
candidate passes' two streams (the caller's and the library's) to the rest.
Step 1: is a mask honoured at all, and how do its bits map to (XCD, CU)?  (probe kernel: HW_REG_XCC_ID / HW_ID per workgroup.)
Step 2: the bench's pipelined loop, default streams against masked ones, interleaved, three runs each.
Lab build: M3PC_LIB=m3pc_amd/libm3pc_hip_lab.so python tools/cu_mask_probe.py [reserved CUs per XCD ...]"""
import ctypes as C
import os
import subprocess
import sys
import time
import types
from collections import deque

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def words_of(bits):
    w = [0] * 8
    for b in bits:
        w[b // 32] |= 1 << (b % 32)
    return w


def masked_stream(hip, words):
    st = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), len(words), arr)
    assert rc == 0, rc
    return st


def probe(torch, lib, stream_ptr, n=4096):
    out = torch.full((2 * n,), -1, dtype=torch.int32, device="cuda")
    lib.m3pc_debug_xcc_probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    assert lib.m3pc_debug_xcc_probe(out.data_ptr(), n, stream_ptr) == 0
    torch.cuda.synchronize()
    o = out.cpu().view(n, 2)
    cus = {}
    for x, hw in o.tolist():
        cu = (x, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)  # (xcc, se, sh, cu)
        cus.setdefault(x, set()).add(cu)
    return {x: len(v) for x, v in sorted(cus.items())}


def step1():
    import torch
    from m3pc_amd import capi
    torch.zeros(1, device="cuda")
    hip = C.CDLL("libamdhip64.so")
    lib = capi.load_library()
    print("default stream:", probe(torch, lib, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    for name, bits in (("bits 0..15", range(16)), ("bits 0..31", range(32)), ("bits 32..255", range(32, 256)),
                       ("bits 0..7", range(8)), ("bits 8..255", range(8, 256)), ("bits 0,8,16,24", (0, 8, 16, 24))):
        st = masked_stream(hip, words_of(bits))
        print("mask %-14s -> CUs per XCD %s" % (name, probe(torch, lib, st)))
        hip.hipStreamDestroy(st)


def run_bench(reserved, steps=60, warmup=10):
    """One process per arrangement (the library's aux stream is created once per process)."""
    import torch
    from m3pc_amd import capi, synth
    from m3pc_amd import planner as P
    torch.zeros(1, device="cuda")
    hip = C.CDLL("libamdhip64.so")
    main_stream = torch.cuda.current_stream()
    # reserved > 0: the chains on `reserved` CUs per XCD of their own, the candidate passes on the rest;
    # reserved < 0: the candidate passes kept OFF |reserved| CUs per XCD, the chains free to run anywhere (a fast lane: a chain
    # launch always finds those CUs free of fused tiles)
    if reserved != 0:
        r = abs(reserved)
        chain_bits = list(range(8 * r))          # bit i -> XCD i % 8 (step 1): r CUs on every XCD
        cand_bits = list(range(8 * r, 256))
        if reserved > 0:
            P._CHAIN_STREAMS[(str(torch.device("cuda", 0)), -1)] = tuple(
                torch.cuda.ExternalStream(masked_stream(hip, words_of(chain_bits)).value) for _ in range(2))
        main_stream = torch.cuda.ExternalStream(masked_stream(hip, words_of(cand_bits)).value)
    dims = synth.Dims(11, 3, 32)
    cfg = types.SimpleNamespace(traj_length=32, action_samples=1024, horizon=16, discount=0.99, temperature=0.01, lmbda=0.6,
                                plan_guidance="rtg_guiding")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)
    with torch.cuda.stream(main_stream):
        pl = P.HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision="bf16",
                          device=0, generator=gen, pipeline_depth=3)
        hist = synth.make_history(dims, 0)
        hist["path_length"] = 500
        s, a, r, h, rtg = pl.assemble_window(hist, rtg=3.0)

        def run(k):
            flight = deque()
            for _ in range(k):
                flight.append(pl._issue(capi.MODE_RTG, s, a, r, rtg, h, 0.6, pipelined=True, inputs_ready=True))
                if len(flight) > 3:
                    flight.popleft().pair()
            while flight:
                flight.popleft().pair()

        run(12)
        run(warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # one step alone
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for e0, e1 in ev:
            e0.record()
            pl._guide(capi.MODE_RTG, s, a, r, rtg, h, 0.6)
            e1.record()
        torch.cuda.synchronize()
        lat = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    print("reserved %d CUs/XCD for the chains: %.1f plan-steps/s  (%.4f ms/step), one step alone p50 %.3f ms" %
          (reserved, steps / dt, 1e3 * dt / steps, lat[len(lat) // 2]), flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--run":
        return run_bench(int(sys.argv[2]))
    if "--no-map" not in sys.argv:
        step1()
    sys.argv = [a for a in sys.argv if a != "--no-map"]
    arrangements = [int(a) for a in sys.argv[1:]] or [0, 2, 4]
    for rep in range(3):
        for r in arrangements:
            env = dict(os.environ)
            if r != 0:
                env["M3PC_AUX_CU_MASK"] = ",".join("%x" % w for w in words_of(range(8 * abs(r), 256)))
            subprocess.run([sys.executable, os.path.abspath(__file__), "--run", str(r)], env=env, timeout=300)


if __name__ == "__main__":
    main()

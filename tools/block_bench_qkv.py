"""Phase clocks and launch time of the fused layer tail WITH the next layer's Q|K|V projection (lab build):
    M3PC_LIB=m3pc_amd/libm3pc_hip_lab.so python tools/block_bench_qkv.py [rows ...]"""
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
    rows = [int(a) for a in sys.argv[1:]] or [25088, 50176]
    W, p, lnB, g = make_params(0)
    Wqkv = (torch.randn(3 * D, D, device=dev, generator=g) / D ** 0.5).to(torch.bfloat16)
    bqkv = 0.1 * torch.randn(3 * D, device=dev, generator=g)
    nbytes = lib.m3pc_debug_block_stream_bytes
    nbytes.restype = C.c_longlong
    sb = torch.empty(int(nbytes()), dtype=torch.uint8, device=dev)
    fn = lib.m3pc_debug_block_fused_qkv
    fn.restype = C.c_int
    vp = C.c_void_p
    fn.argtypes = [vp, C.c_int] + [vp] * 18 + [C.c_int]
    for M in rows:
        O = torch.randn(M, D, device=dev, generator=g).to(torch.bfloat16)
        R = torch.randn(M, D, device=dev, generator=g)
        X = torch.empty_like(R)
        H = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
        Q = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
        stamps = torch.zeros(4, 16, dtype=torch.int64, device=dev)

        Rb = R.to(torch.bfloat16)   # (round 6: the same launches on a bf16 residual stream, in place as the candidate pass runs them)
        Xb = Rb.clone()

        def qkv(st=None, xb=0):
            s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            rc = fn(O.data_ptr(), M, (Xb if xb else R).data_ptr(), W["o"].data_ptr(), W["1"].data_ptr(), W["2"].data_ptr(), Wqkv.data_ptr(),
                    sb.data_ptr(), p["bo"].data_ptr(), p["b1"].data_ptr(), p["b2"].data_ptr(), p["g2"].data_ptr(),
                    p["be2"].data_ptr(), p["gA"].data_ptr(), p["bA"].data_ptr(), bqkv.data_ptr(), (Xb if xb else X).data_ptr(), Q.data_ptr(), s,
                    st.data_ptr() if st is not None else None, xb)
            assert rc == 0, lib.m3pc_last_error()

        def plain(st=None):
            _call(lib, O, R, None, 1, W, sb, 0, p, [None] * 4, 0, 0, X, H, 0, sync=False, stamps=st)

        def plain_nox(st=None):   # (the last encoder layer's tail: X'' is dead)
            _call(lib, O, R, None, 1, W, sb, 0, p, [None] * 4, 0, 0, None, H, 0, sync=False, stamps=st)

        def plain_xb(st=None):
            _call(lib, O, Rb, None, 1, W, sb, 0, p, [None] * 4, 0, 0, None, H, 16, sync=False, stamps=st)

        fl_q, fl_p = 2.0 * M * (D * D + 2 * D * FF + 3 * D * D), 2.0 * M * (D * D + 2 * D * FF)
        for name, f, fl in (("tail+qkv", qkv, fl_q), ("tail+qkv bf16-res", lambda st=None: qkv(st, 1), fl_q), ("tail", plain, fl_p),
                            ("tail no-X", plain_nox, fl_p), ("tail no-X bf16-res", plain_xb, fl_p)):
            for _ in range(3):
                f(stamps)
            torch.cuda.synchronize()
            s = stamps.cpu()
            names = ["prologue", "out-proj", "LN2", "FFN", "X store", "LN + H store | qkv"]
            for w in (0, 3):
                dl = [int(s[w, k + 1] - s[w, k]) for k in range(6)]
                extra = f"  [fragments {int(s[w, 8] - s[w, 5])}  48 phases {int(s[w, 9] - s[w, 8])}  last pair {int(s[w, 6] - s[w, 9])}]" if name.startswith("tail+qkv") else ""
                print(f"{name} rows {M} wave {w} clocks: " + "  ".join(f"{n} {v}" for n, v in zip(names, dl)) + f"  total {int(s[w, 6] - s[w, 0])}" + extra)
            ts = []
            for _ in range(12):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                f(None)
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            ts.sort()
            print(f"{name} rows {M:6d}: min {ts[0]:7.1f} us  med {ts[len(ts)//2]:7.1f} us  {fl / ts[0] / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()

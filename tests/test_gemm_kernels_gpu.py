"""GPU checks of the bf16 GEMM kernels against each other and against torch fp32 (needs an MI355X).

The library picks a GEMM kernel by shape (128x128 three-slot ring by default, 256x256 tiles at one wave per SIMD for
long-K many-row problems).  Sharded and single-GPU runs must agree bit for bit (tests/test_hip_parity.py::
test_sharding_is_exact), and a shard sees a different row count, hence possibly a different kernel: every
kernel on the default dispatch must therefore produce IDENTICAL bits -- same MFMA instruction, same k order, same
epilogue arithmetic.  This test holds them to that through the library's debug entry (m3pc_debug_gemm; lab build: include/m3pc_hip_debug.h).
"""
import ctypes as C

import pytest
import torch

from m3pc_amd import capi  # noqa: F401
from hip_util import lab_library

pytestmark = pytest.mark.gpu


def _gemm(lib, A, W, bias, R, out, gelu, variant):
    fn = lib.m3pc_debug_gemm
    fn.restype = C.c_int
    vp, i = C.c_void_p, C.c_int
    fn.argtypes = [i, vp, vp, vp, vp, vp, i, i, i, i, i, i, vp]
    M, K = A.shape
    N = W.shape[0]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = fn(1, A.data_ptr(), W.data_ptr(), bias.data_ptr(), R.data_ptr() if R is not None else None, out.data_ptr(),
            M, N, K, gelu, int(out.dtype == torch.float32), variant, st)
    assert rc == 0, lib.m3pc_last_error()
    torch.cuda.synchronize()


# (M, N, K, residual): full 256-row tiles, a ragged last tile, the long-K residual GEMM the 256x256 kernel is used for
@pytest.mark.parametrize("M,N,K,res", [(57344, 512, 2048, True), (57344 + 77, 512, 1024, True), (61440, 256, 2048, False)])
def test_big_tile_kernel_is_bit_identical_to_the_ring(M, N, K, res):
    lib = lab_library()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    R = torch.randn(M, N, device=dev, generator=g) if res else None
    outs = {}
    for v in (0, 26, 37, 2, 43, 44):  # default dispatch, ring (no peeling), 256x256 tiles, plain double buffer, line 128/256
        out = torch.full((M, N), float("nan"), device=dev, dtype=torch.float32)
        _gemm(lib, A, W, bias, R, out, 0, v)
        outs[v] = out
    assert torch.equal(outs[37], outs[26]), "256x256 kernel and 128x128 ring differ"
    assert torch.equal(outs[0], outs[26]), "default dispatch differs from the ring"
    assert torch.equal(outs[2], outs[26]), "double-buffer kernel differs from the ring"
    assert torch.equal(outs[43], outs[26]), "whole-line 128x128 kernel differs from the ring"
    assert torch.equal(outs[44], outs[26]), "whole-line 256x256 kernel differs from the ring"
    # and all of them are the right product: fp32 reference on a row sample (bf16 operands are exact in fp32)
    sel = torch.cat([torch.arange(300), torch.arange(M - 300, M)]).to(dev)
    ref = A[sel].float() @ W.float().T + bias
    if res:
        ref = ref + R[sel]
    err = float((outs[37][sel] - ref).abs().max())
    assert err <= 2e-5 * float(ref.abs().max()), err  # fp32 accumulation, summation order only


# the K = 512 class (bf16 out, optional GELU): default dispatch = whole-line kernel, persistent workgroups; ragged and
# row-mapped shapes take its one-workgroup-per-tile form
@pytest.mark.parametrize("M,N,K,gelu", [(50176, 1536, 512, 0), (50176 + 40, 1024, 512, 1), (16384, 512, 192, 0), (70000, 2048, 512, 1)])
def test_line_kernel_is_bit_identical_to_the_ring_bf16_out(M, N, K, gelu):
    lib = lab_library()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(M + N + K + gelu)
    A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    outs = {}
    for v in (0, 26, 43):
        out = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        _gemm(lib, A, W, bias, None, out, gelu, v)
        outs[v] = out
    assert torch.equal(outs[43], outs[26]), "whole-line kernel and ring differ"
    assert torch.equal(outs[0], outs[26]), "default dispatch differs from the ring"
    sel = torch.cat([torch.arange(300), torch.arange(M - 300, M)]).to(dev)
    ref = A[sel].float() @ W.float().T + bias
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    err = float((outs[43][sel].float() - ref).abs().max())
    assert err <= 2.0 ** -8 * max(float(ref.abs().max()), 1.0), err  # one bf16 rounding of the result


# top-k kernels (rank-by-counting for n <= 2048, k-round selection up to 16384, bitonic beyond / for k > 64): descending
# value, ties to the lower index -- torch's stable sort of the negated scores
@pytest.mark.parametrize("n", [5, 64, 1000, 1024, 1025, 2048, 2049, 8192, 16384])
@pytest.mark.parametrize("k", [1, 16, 64])
def test_topk_order_matches_stable_sort(n, k):
    if k > n:
        pytest.skip("k <= n by contract")
    lib = lab_library()
    fn = lib.m3pc_debug_topk
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(n * 131 + k)
    for ties in (False, True):
        v = torch.randn(n, device=dev, generator=g)
        if ties:
            v = torch.round(v * 2.0) / 2.0  # a handful of distinct values: the order is decided by the index
        out = torch.full((k,), -1, device=dev, dtype=torch.int32)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert fn(v.data_ptr(), n, k, out.data_ptr(), st) == 0, lib.m3pc_last_error()
        torch.cuda.synchronize()
        ref = torch.sort(-v, stable=True).indices[:k].to(torch.int32)
        assert torch.equal(out, ref), (n, k, ties)

"""GPU check of the fused layer tail (m3pc_amd/csrc/block_fused.hip) against plain torch fp32 with the same bf16
rounding points: out-proj + residual -> LayerNorm2 -> Linear/GELU/Linear + residual -> LayerNorm(s)
(mtm_model.py:379-409 halves of nn.TransformerEncoderLayer, norm_first, exact-erf GELU).  Through the library's debug
entry m3pc_debug_block_fused (lab build: include/m3pc_hip_debug.h)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from m3pc_amd import capi  # noqa: F401
from hip_util import lab_library

pytestmark = pytest.mark.gpu

D, FF = 512, 2048


def _call(lib, O, res, rowtab, rt_mod, W, stream_buf, pack, p, lnB, out_mod, out_grp, Xout, Hout, variant=0, sync=True,
          stamps=None):
    fn = lib.m3pc_debug_block_fused
    fn.restype = C.c_int
    vp, i = C.c_void_p, C.c_int
    fn.argtypes = [vp, i, vp, vp, i, vp, vp, vp, vp, i] + [vp] * 11 + [i, i, vp, vp, i, vp, vp]
    ptr = lambda t: t.data_ptr() if t is not None else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = fn(O.data_ptr(), O.shape[0], ptr(res), ptr(rowtab), rt_mod, W["o"].data_ptr(), W["1"].data_ptr(), W["2"].data_ptr(),
            stream_buf.data_ptr(), pack, p["bo"].data_ptr(), p["b1"].data_ptr(), p["b2"].data_ptr(), p["g2"].data_ptr(),
            p["be2"].data_ptr(), p["gA"].data_ptr(), p["bA"].data_ptr(), ptr(lnB[0]), ptr(lnB[1]), ptr(lnB[2]), ptr(lnB[3]),
            out_mod, out_grp, ptr(Xout), ptr(Hout), variant, st, ptr(stamps))
    assert rc == 0, lib.m3pc_last_error()
    if sync:
        torch.cuda.synchronize()


def make_params(seed):
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(seed)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    W = {"o": (rn(D, D) / D ** 0.5).to(torch.bfloat16), "1": (rn(FF, D) / D ** 0.5).to(torch.bfloat16),
         "2": (rn(D, FF) / FF ** 0.5).to(torch.bfloat16)}
    p = {"bo": 0.1 * rn(D), "b1": 0.1 * rn(FF), "b2": 0.1 * rn(D), "g2": 1 + 0.1 * rn(D), "be2": 0.1 * rn(D),
         "gA": 1 + 0.1 * rn(D), "bA": 0.1 * rn(D)}
    lnB = [1 + 0.1 * rn(D), 0.1 * rn(D), 1 + 0.1 * rn(D), 0.1 * rn(D)]
    return W, p, lnB, g


def reference(O, R, W, p, lnB=None, sel=None):
    """fp32 torch with the kernel's bf16 rounding points (operands bf16, accumulation fp32)."""
    x1 = R + p["bo"] + O.float() @ W["o"].float().T
    a = F.layer_norm(x1, (D,), p["g2"], p["be2"], 1e-5).to(torch.bfloat16).float()
    hid = F.gelu(a @ W["1"].float().T + p["b1"]).to(torch.bfloat16).float()
    x2 = x1 + p["b2"] + hid @ W["2"].float().T
    y = F.layer_norm(x2, (D,), p["gA"], p["bA"], 1e-5)
    if lnB is not None:
        y0 = F.layer_norm(y, (D,), lnB[0], lnB[1], 1e-5)
        y1 = F.layer_norm(y, (D,), lnB[2], lnB[3], 1e-5)
        y = torch.where(sel[:, None] == 0, y0, y1)
    return x2, y


@pytest.mark.parametrize("M", [128, 4096 + 37, 50176])
def test_block_fused_matches_torch(M):
    lib = lab_library()
    dev = torch.device("cuda")
    W, p, _, g = make_params(M)
    O = torch.randn(M, D, device=dev, generator=g).to(torch.bfloat16)
    R = torch.randn(M, D, device=dev, generator=g)
    nbytes = lib.m3pc_debug_block_stream_bytes
    nbytes.restype = C.c_longlong
    sb = torch.empty(int(nbytes()), dtype=torch.uint8, device=dev)
    Xout = torch.full((M, D), float("nan"), device=dev)
    Hout = torch.full((M, D), float("nan"), device=dev, dtype=torch.bfloat16)
    _call(lib, O, R, None, 1, W, sb, 1, p, [None] * 4, 0, 0, Xout, Hout)
    rows = torch.arange(M, device=dev) if M <= 8192 else torch.cat([torch.arange(2048), torch.arange(M - 2048, M),
                                                                     torch.randint(0, M, (4096,))]).to(dev)
    x2, y = reference(O[rows], R[rows], W, p)
    assert torch.isfinite(Xout).all() and torch.isfinite(Hout.float()).all()
    ex = float((Xout[rows] - x2).abs().max()) / float(x2.abs().max())
    ey = float((Hout[rows].float() - y).abs().max())
    # X'': fp32 accumulation of bf16 products; a hidden value on a bf16 rounding boundary may round the other way
    assert ex <= 2e-3, ex
    assert ey <= 4e-2, ey  # bf16 output (|y| <~ 4: half an ulp is 1.6e-2) on top of ex
    # in place (Xout aliases the residual rows), and bit-identical to the first launch; every row depends on itself only
    R2 = R.clone()
    H2 = torch.empty_like(Hout)
    _call(lib, O, R2, None, 1, W, sb, 0, p, [None] * 4, 0, 0, R2, H2)
    assert torch.equal(R2, Xout) and torch.equal(H2, Hout)
    if M >= 4096:  # a shard of the rows gives the same bits (sharding exactness)
        lo, n = 1280, 1000
        X3 = torch.empty(n, D, device=dev)
        H3 = torch.empty(n, D, device=dev, dtype=torch.bfloat16)
        _call(lib, O[lo:lo + n].contiguous(), R[lo:lo + n].contiguous(), None, 1, W, sb, 0, p, [None] * 4, 0, 0, X3, H3)
        assert torch.equal(X3, Xout[lo:lo + n]) and torch.equal(H3, Hout[lo:lo + n])


def test_block_fused_rowtab_and_head_norms():
    """decoder form: residual from a shared row table, no fp32 output, decoder.norm then one of two head LayerNorms by
    row group, rows regrouped per head"""
    lib = lab_library()
    dev = torch.device("cuda")
    n, nq, hh = 512, 32, 16
    M = n * nq
    W, p, lnB, g = make_params(7)
    O = torch.randn(M, D, device=dev, generator=g).to(torch.bfloat16)
    tab = torch.randn(nq, D, device=dev, generator=g)
    nbytes = lib.m3pc_debug_block_stream_bytes
    nbytes.restype = C.c_longlong
    sb = torch.empty(int(nbytes()), dtype=torch.uint8, device=dev)
    Hout = torch.full((M, D), float("nan"), device=dev, dtype=torch.bfloat16)
    _call(lib, O, None, tab, nq, W, sb, 1, p, lnB, nq, hh, None, Hout)
    r = torch.arange(M, device=dev)
    sel = (r % nq) // hh
    _, y = reference(O, tab[r % nq], W, p, lnB, sel)
    orow = sel * (M // nq) * hh + (r // nq) * hh + r % hh
    got = Hout[orow].float()
    assert torch.isfinite(Hout.float()).all()
    assert float((got - y).abs().max()) <= 4e-2


@pytest.mark.parametrize("M", [128, 3 * 1024 + 77, 25088])
def test_block_fused_with_next_qkv(M):
    """encoder form with the next layer's Q|K|V projection behind the tail: QKV = bf16(LN_A(X'')) Wqkv^T + bqkv; X'' as without"""
    lib = lab_library()
    dev = torch.device("cuda")
    W, p, _, g = make_params(1000 + M)
    Wqkv = (torch.randn(3 * D, D, device=dev, generator=g) / D ** 0.5).to(torch.bfloat16)
    bqkv = 0.1 * torch.randn(3 * D, device=dev, generator=g)
    O = torch.randn(M, D, device=dev, generator=g).to(torch.bfloat16)
    R = torch.randn(M, D, device=dev, generator=g)
    nbytes = lib.m3pc_debug_block_stream_bytes
    nbytes.restype = C.c_longlong
    sb = torch.empty(int(nbytes()), dtype=torch.uint8, device=dev)
    fn = lib.m3pc_debug_block_fused_qkv
    fn.restype = C.c_int
    vp = C.c_void_p
    fn.argtypes = [vp, C.c_int] + [vp] * 18 + [C.c_int]

    def call(O_, R_, Xout, QKV, xb=0):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = fn(O_.data_ptr(), O_.shape[0], R_.data_ptr(), W["o"].data_ptr(), W["1"].data_ptr(), W["2"].data_ptr(), Wqkv.data_ptr(),
                sb.data_ptr(), p["bo"].data_ptr(), p["b1"].data_ptr(), p["b2"].data_ptr(), p["g2"].data_ptr(), p["be2"].data_ptr(),
                p["gA"].data_ptr(), p["bA"].data_ptr(), bqkv.data_ptr(), Xout.data_ptr(), QKV.data_ptr(), st, None, xb)
        assert rc == 0, lib.m3pc_last_error()
        torch.cuda.synchronize()

    Xout = torch.full((M, D), float("nan"), device=dev)
    QKV = torch.full((M + 1, 3 * D), float("nan"), device=dev, dtype=torch.bfloat16)  # (one guard row behind the end)
    call(O, R, Xout, QKV)
    assert torch.isnan(QKV[M].float()).all(), "rows past M written"
    QKV = QKV[:M]
    rows = torch.arange(M, device=dev) if M <= 8192 else torch.cat([torch.arange(2048), torch.arange(M - 2048, M),
                                                                     torch.randint(0, M, (4096,))]).to(dev)
    x2, y = reference(O[rows], R[rows], W, p)
    assert torch.isfinite(Xout).all() and torch.isfinite(QKV.float()).all()
    assert float((Xout[rows] - x2).abs().max()) / float(x2.abs().max()) <= 2e-3
    # from the kernel's own X'' (so that only the projection is compared): LN_A -> bf16 -> Wqkv
    yk = F.layer_norm(Xout[rows], (D,), p["gA"], p["bA"], 1e-5).to(torch.bfloat16).float()
    want = yk @ Wqkv.float().T + bqkv
    err = float((QKV[rows].float() - want).abs().max())
    assert err <= 6e-2, err  # bf16 output (|q| <~ 6) + LN values on a bf16 rounding boundary
    # the tail without the projection gives the same X'' bits, and a shard of the rows the same Q|K|V bits
    X1 = torch.empty_like(Xout)
    H1 = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    _call(lib, O, R, None, 1, W, sb, 0, p, [None] * 4, 0, 0, X1, H1)
    assert torch.equal(X1, Xout)
    if M >= 3000:
        lo, n = 640, 1000
        X3 = torch.empty(n, D, device=dev)
        Q3 = torch.empty(n, 3 * D, device=dev, dtype=torch.bfloat16)
        call(O[lo:lo + n].contiguous(), R[lo:lo + n].contiguous(), X3, Q3)
        assert torch.equal(X3, Xout[lo:lo + n]) and torch.equal(Q3, QKV[lo:lo + n])
    # round 6: the same launch on a bf16 residual stream (BlockP::x_bf16: residual rows in and X'' rows out are bf16).  With the
    # residual the fp32 launch saw rounded to bf16 beforehand, X' -- hence every Q|K|V bit -- is the fp32-row launch's, and X''
    # is that launch's X'' rounded to bf16; in place (Xout = res) as the candidate pass runs it
    Rb = R.to(torch.bfloat16)
    Xf = torch.empty_like(Xout)
    Qf = torch.full((M + 1, 3 * D), float("nan"), device=dev, dtype=torch.bfloat16)
    call(O, Rb.float(), Xf, Qf)
    Xb = torch.cat([Rb.clone(), torch.full((1, D), float("nan"), device=dev, dtype=torch.bfloat16)])  # (guard row)
    Qb = torch.full((M + 1, 3 * D), float("nan"), device=dev, dtype=torch.bfloat16)
    call(O, Xb, Xb, Qb, xb=1)
    assert torch.isnan(Xb[M].float()).all() and torch.isnan(Qb[M].float()).all(), "rows past M written"
    assert torch.equal(Qb[:M], Qf[:M])
    assert torch.equal(Xb[:M], Xf.to(torch.bfloat16))


@pytest.mark.parametrize("M", [128, 4096 + 37, 25088])
def test_block_fused_on_a_bf16_residual_stream(M):
    """BlockP::x_bf16 on the plain form (the last encoder layer's tail: X'' is not stored, encoder.norm rows are; and with X''
    stored): bit-identical to the fp32-row launch fed the bf16-rounded residual."""
    lib = lab_library()
    dev = torch.device("cuda")
    W, p, _, g = make_params(77 + M)
    O = torch.randn(M, D, device=dev, generator=g).to(torch.bfloat16)
    Rb = torch.randn(M, D, device=dev, generator=g).to(torch.bfloat16)
    nbytes = lib.m3pc_debug_block_stream_bytes
    nbytes.restype = C.c_longlong
    sb = torch.empty(int(nbytes()), dtype=torch.uint8, device=dev)
    Xf = torch.empty(M, D, device=dev)
    Hf = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    _call(lib, O, Rb.float(), None, 1, W, sb, 1, p, [None] * 4, 0, 0, Xf, Hf)
    x2, y = reference(O, Rb.float(), W, p)
    assert float((Xf - x2).abs().max()) / float(x2.abs().max()) <= 2e-3
    Hb = torch.full((M + 1, D), float("nan"), device=dev, dtype=torch.bfloat16)
    _call(lib, O, Rb, None, 1, W, sb, 0, p, [None] * 4, 0, 0, None, Hb, variant=16)       # X'' not stored
    assert torch.isnan(Hb[M].float()).all() and torch.equal(Hb[:M], Hf)
    Xb = torch.cat([Rb.clone(), torch.full((1, D), float("nan"), device=dev, dtype=torch.bfloat16)])
    Hb2 = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    _call(lib, O, Xb, None, 1, W, sb, 0, p, [None] * 4, 0, 0, Xb, Hb2, variant=16)        # in place
    assert torch.isnan(Xb[M].float()).all() and torch.equal(Hb2, Hf) and torch.equal(Xb[:M], Xf.to(torch.bfloat16))


@pytest.mark.parametrize("n,hh,detok", [(512, 16, True), (333, 5, False), (64, 32, True)])
def test_block_fused_with_scalar_output_heads(n, hh, detok):
    """decoder form with the two scalar output heads inside the tail (rtg_guiding's rewards / returns heads, mtm_model.py:428-433):
    head_s(row) = w2_s . gelu(W1_s LN_s(decoder.norm(X'')) + b1_s) + b2_s, de-tokenised; rows of group s = (r % 2h) // h"""
    lib = lab_library()
    dev = torch.device("cuda")
    nq = 2 * hh
    M = n * nq
    W, p, lnB, g = make_params(4000 + n)
    rn = lambda *sh: torch.randn(*sh, device=dev, generator=g)
    Wh = (rn(2, D, D) / D ** 0.5).to(torch.bfloat16)
    hb1, hw2, hb2 = 0.1 * rn(2, D), rn(2, D) / D ** 0.5, 0.1 * rn(2)
    hmean, hstd = rn(2), rn(2).abs() + 0.5
    O = rn(M, D).to(torch.bfloat16)
    tab = rn(nq, D)
    nbytes = lib.m3pc_debug_block_stream_bytes
    nbytes.restype = C.c_longlong
    sb = torch.empty(int(nbytes()), dtype=torch.uint8, device=dev)
    out = [torch.full((M // 2 + 1,), float("nan"), device=dev) for _ in range(2)]  # (one guard element behind the end)
    fn = lib.m3pc_debug_block_fused_heads
    fn.restype = C.c_int
    vp, i = C.c_void_p, C.c_int
    fn.argtypes = [vp, i, vp, i] + [vp] * 16 + [i, i] + [vp] * 9
    ptr = lambda t: t.data_ptr() if t is not None else None
    rc = fn(O.data_ptr(), M, tab.data_ptr(), nq, W["o"].data_ptr(), W["1"].data_ptr(), W["2"].data_ptr(), Wh.data_ptr(), sb.data_ptr(),
            p["bo"].data_ptr(), p["b1"].data_ptr(), p["b2"].data_ptr(), p["g2"].data_ptr(), p["be2"].data_ptr(), p["gA"].data_ptr(),
            p["bA"].data_ptr(), lnB[0].data_ptr(), lnB[1].data_ptr(), lnB[2].data_ptr(), lnB[3].data_ptr(), nq, hh, hb1.data_ptr(),
            hw2.data_ptr(), hb2.data_ptr(), ptr(hmean if detok else None), ptr(hstd if detok else None), out[0].data_ptr(),
            out[1].data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream), None)
    assert rc == 0, lib.m3pc_last_error()
    torch.cuda.synchronize()
    r = torch.arange(M, device=dev)
    sel = (r % nq) // hh
    _, y = reference(O, tab[r % nq], W, p, lnB, sel)  # LN_head(decoder.norm(X'')) per row, fp32
    yb = y.to(torch.bfloat16).float()
    for s_ in range(2):
        rows = r[sel == s_]  # in order: the i-th row of group s
        hid = F.gelu(yb[rows] @ Wh[s_].float().T + hb1[s_])
        want = hid @ hw2[s_] + hb2[s_]
        if detok:
            want = want * hstd[s_] + hmean[s_]
        got = out[s_][: M // 2]
        assert torch.isnan(out[s_][M // 2]), "element past the end written"
        assert torch.isfinite(got).all()
        err = float((got - want).abs().max())
        assert err <= 3e-2 * max(1.0, float(want.abs().max())), (s_, err)


# ------------------------------------------------------------------------------------------------ decoder input (kv_fused_kernel)
def _kv_call(lib, Z, n, Le, kept, off, We, Wkv, rowtab, g, b, bkv, stamps=None):
    fn = lib.m3pc_debug_kv_fused
    fn.restype = C.c_int
    vp, i = C.c_void_p, C.c_int
    fn.argtypes = [vp, i, i, i, i, i, i] + [vp] * 12
    lib.m3pc_debug_kv_stream_bytes.restype = C.c_longlong
    sb = torch.empty(2 * lib.m3pc_debug_kv_stream_bytes(), dtype=torch.uint8, device="cuda")
    KV = torch.zeros(n * Le, 2 * D, dtype=torch.bfloat16, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = fn(Z.data_ptr(), n, Le, kept[0], off[0], kept[1], off[1], We[0].data_ptr(), We[1].data_ptr(), Wkv.data_ptr(), sb.data_ptr(),
            rowtab[0].data_ptr(), rowtab[1].data_ptr(), g.data_ptr(), b.data_ptr(), bkv.data_ptr(), KV.data_ptr(), st,
            stamps.data_ptr() if stamps is not None else None)
    assert rc == 0, lib.m3pc_last_error()
    torch.cuda.synchronize()
    return KV


def _kv_reference(Z, n, Le, kept, off, We, Wkv, rowtab, g, b, bkv):
    out = torch.zeros(n * Le, 2 * D, dtype=torch.float32, device="cuda")
    Zr = Z.float().view(n, Le, D)
    o = out.view(n, Le, 2 * D)
    for k in range(2):
        if kept[k] == 0:
            continue
        y = Zr[:, off[k]:off[k] + kept[k]] @ We[k].float().T + rowtab[k][None]
        a = F.layer_norm(y, (D,), g, b, 1e-5).to(torch.bfloat16).float()
        o[:, off[k]:off[k] + kept[k]] = a @ Wkv.float().T + bkv
    return out


@pytest.mark.parametrize("n,Le,kept,off", [(64, 49, (17, 32), (0, 17)), (100, 49, (17, 32), (0, 17)), (37, 40, (9, 31), (0, 9)),
                                           (16, 32, (32, 0), (0, 0)), (1024, 49, (17, 32), (0, 17))])
def test_kv_fused_matches_reference(n, Le, kept, off):
    lib = lab_library()
    dev = torch.device("cuda")
    g_ = torch.Generator(device=dev).manual_seed(7 + n)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g_)
    Z = rn(n * Le, D).to(torch.bfloat16)
    We = [(rn(D, D) / D ** 0.5).to(torch.bfloat16) for _ in range(2)]
    Wkv = (rn(2 * D, D) / D ** 0.5).to(torch.bfloat16)
    rowtab = [0.5 * rn(max(kept[k], 1), D) for k in range(2)]
    g, b, bkv = 1 + 0.1 * rn(D), 0.1 * rn(D), 0.1 * rn(2 * D)
    KV = _kv_call(lib, Z, n, Le, kept, off, We, Wkv, rowtab, g, b, bkv).float()
    ref = _kv_reference(Z, n, Le, kept, off, We, Wkv, rowtab, g, b, bkv)
    used = torch.zeros(Le, dtype=torch.bool, device=dev)
    for k in range(2):
        used[off[k]:off[k] + kept[k]] = True
    m = used.repeat(n)
    assert (KV[~m] == 0).all()                       # rows of no group are not touched
    err = (KV[m] - ref[m]).abs()
    # one bf16 ulp of the output plus the flips of the LayerNorm rows' roundings
    assert float(err.max()) <= 0.06 and float(err.mean()) <= 4e-3, (float(err.max()), float(err.mean()))


def test_kv_fused_rows_independent_of_batch():
    """A candidate's K|V rows do not depend on how many candidates the launch holds (candidate sharding stays bit-exact)."""
    lib = lab_library()
    dev = torch.device("cuda")
    g_ = torch.Generator(device=dev).manual_seed(3)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g_)
    n, Le, kept, off = 300, 49, (17, 32), (0, 17)
    Z = rn(n * Le, D).to(torch.bfloat16)
    We = [(rn(D, D) / D ** 0.5).to(torch.bfloat16) for _ in range(2)]
    Wkv = (rn(2 * D, D) / D ** 0.5).to(torch.bfloat16)
    rowtab = [0.5 * rn(kept[k], D) for k in range(2)]
    g, b, bkv = 1 + 0.1 * rn(D), 0.1 * rn(D), 0.1 * rn(2 * D)
    full = _kv_call(lib, Z, n, Le, kept, off, We, Wkv, rowtab, g, b, bkv)
    part = _kv_call(lib, Z[37 * Le:(37 + 101) * Le].contiguous(), 101, Le, kept, off, We, Wkv, rowtab, g, b, bkv)
    assert torch.equal(full[37 * Le:(37 + 101) * Le], part)


# ------------------------------------------------------------------------------------------------ bf16 attention (attn_bf16.hip)
@pytest.mark.parametrize("batch,n_own,n_sh", [(512, 49, 0), (512, 17, 32), (300, 49, 0), (257, 17, 32), (256, 20, 29),
                                              (512, 97, 0), (300, 97, 0), (512, 33, 64), (257, 33, 64), (256, 40, 57)])
def test_pipelined_attention_is_the_direct_kernel_bit_for_bit(batch, n_own, n_sh):
    """attn_bf16_pipe_kernel (persistent workgroups, rows by LDS-DMA, two items in flight) against attn_bf16_direct_kernel on the two
    encoder-layer shapes of the candidate pass (second layer: 49 own rows; first: 17 own + 32 history rows shared by the batch): same
    products, same order -- equal bits -- and both against a float64 softmax.  (20 + 29: a split only the direct kernel takes.)
    Item counts that are / are not multiples of the grid, so that workgroups end on different iterations.  97 rows (33 own + 64
    shared in the first layer): the T = 64 pass of BASELINE config 4, attn_bf16_pipe_wide_kernel against attn_bf16_direct_kernel<4, 2, 4>."""
    lib = lab_library()
    fn = lib.m3pc_debug_attention_bf16
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p] * 2
    g = torch.Generator(device="cuda").manual_seed(batch + n_sh)
    qkv = torch.randn(batch, n_own, 1536, device="cuda", generator=g).to(torch.bfloat16)
    qkvs = torch.randn(max(n_sh, 1), 1536, device="cuda", generator=g).to(torch.bfloat16)
    L = n_own + n_sh
    outs = []
    for kernel in (0, 1):
        O = torch.full((batch, L, 512), float("nan"), device="cuda", dtype=torch.bfloat16)
        rc = fn(qkv.data_ptr(), qkvs.data_ptr() if n_sh else None, O.data_ptr(), batch, n_own, n_sh, kernel,
                C.c_void_p(torch.cuda.current_stream().cuda_stream), None)
        assert rc == 0, lib.m3pc_last_error()
        torch.cuda.synchronize()
        outs.append(O)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    # float64 reference on a few batch elements
    for b in (0, batch // 2, batch - 1):
        rows = torch.cat([qkvs[:n_sh], qkv[b]], 0).double() if n_sh else qkv[b].double()  # output order: shared rows first
        q, k, v = rows[:, :512], rows[:, 512:1024], rows[:, 1024:]
        ref = torch.cat([torch.softmax(q[:, 128 * h:128 * h + 128] @ k[:, 128 * h:128 * h + 128].T / 128 ** 0.5, -1)
                         @ v[:, 128 * h:128 * h + 128] for h in range(4)], 1)
        assert float((outs[0][b].double() - ref).abs().max()) < 3e-2


@pytest.mark.parametrize("n,nq,Lm", [(512, 32, 47), (300, 32, 47), (257, 20, 30)])
def test_pipelined_decoder_attention_is_the_direct_kernel_bit_for_bit(n, nq, Lm):
    """attn_bf16_pipe_dec_kernel (two compute waves on two different items, K|V rows by LDS-DMA, the pre-reduced block of the masked tokens'
    keys merged in registers) against attn_bf16_direct_kernel<4, 1, 2>: equal bits, and both against a float64 softmax over all keys."""
    lib = lab_library()
    fn = lib.m3pc_debug_attention_dec_bf16
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p]
    g = torch.Generator(device="cuda").manual_seed(n + Lm)
    qtab = torch.randn(nq, 1536, device="cuda", generator=g).to(torch.bfloat16)
    qkvm = torch.randn(Lm, 1536, device="cuda", generator=g).to(torch.bfloat16)
    kv = torch.randn(n, 49, 1024, device="cuda", generator=g).to(torch.bfloat16)
    pre = torch.zeros(4 * nq * 130, device="cuda")
    outs = []
    for kernel in (0, 1):
        O = torch.full((n, nq, 512), float("nan"), device="cuda", dtype=torch.bfloat16)
        rc = fn(qtab.data_ptr(), qkvm.data_ptr(), kv.data_ptr(), O.data_ptr(), pre.data_ptr(), n, nq, Lm, kernel,
                C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, lib.m3pc_last_error()
        torch.cuda.synchronize()
        outs.append(O)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    for b in (0, n // 2, n - 1):
        q = qtab[:, :512].double()
        k = torch.cat([kv[b, :, :512], qkvm[:, 512:1024]], 0).double()
        v = torch.cat([kv[b, :, 512:], qkvm[:, 1024:]], 0).double()
        ref = torch.cat([torch.softmax(q[:, 128 * h:128 * h + 128] @ k[:, 128 * h:128 * h + 128].T / 128 ** 0.5, -1)
                         @ v[:, 128 * h:128 * h + 128] for h in range(4)], 1)
        assert float((outs[0][b].double() - ref).abs().max()) < 3e-2


@pytest.mark.parametrize("n,nq,Lm", [(512, 64, 95), (300, 64, 95), (257, 50, 70)])
def test_pipelined_wide_decoder_attention_is_the_direct_kernel_bit_for_bit(n, nq, Lm):
    """The T = 64 decoder (BASELINE config 4: 64 batch-shared queries, the candidate's own 97 K|V rows, the pre-reduced block of the masked
    tokens' keys): attn_bf16_pipe_wide_kernel<0, 64, 97, 0, true> against attn_bf16_direct_kernel<4, 2, 4>: equal bits, and both
    against a float64 softmax over all keys."""
    lib = lab_library()
    fn = lib.m3pc_debug_attention_dec_le_bf16
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_void_p]
    Le = 97
    g = torch.Generator(device="cuda").manual_seed(n + Lm)
    qtab = torch.randn(nq, 1536, device="cuda", generator=g).to(torch.bfloat16)
    qkvm = torch.randn(Lm, 1536, device="cuda", generator=g).to(torch.bfloat16)
    kv = torch.randn(n, Le, 1024, device="cuda", generator=g).to(torch.bfloat16)
    pre = torch.zeros(4 * nq * 130, device="cuda")
    outs = []
    for kernel in (0, 1):
        O = torch.full((n, nq, 512), float("nan"), device="cuda", dtype=torch.bfloat16)
        rc = fn(qtab.data_ptr(), qkvm.data_ptr(), kv.data_ptr(), O.data_ptr(), pre.data_ptr(), n, nq, Lm, Le, kernel,
                C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, lib.m3pc_last_error()
        torch.cuda.synchronize()
        outs.append(O)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    for b in (0, n // 2, n - 1):
        q = qtab[:, :512].double()
        k = torch.cat([kv[b, :, :512], qkvm[:, 512:1024]], 0).double()
        v = torch.cat([kv[b, :, 512:], qkvm[:, 1024:]], 0).double()
        ref = torch.cat([torch.softmax(q[:, 128 * h:128 * h + 128] @ k[:, 128 * h:128 * h + 128].T / 128 ** 0.5, -1)
                         @ v[:, 128 * h:128 * h + 128] for h in range(4)], 1)
        assert float((outs[0][b].double() - ref).abs().max()) < 3e-2


@pytest.mark.parametrize("batch,L", [(4096, 12), (4097, 10), (64, 16), (65, 1), (333, 7)])
def test_packed_attention_of_short_windows_is_the_direct_kernel_bit_for_bit(batch, L):
    """attn_bf16_pack2_kernel: two windows of <= 16 tokens per 32 x 32 tile behind a block-diagonal mask (the zero-shot passes of config 5)
    against one window per tile: equal bits (16-slot alignment keeps every sum's order), odd batch sizes leave half a tile empty."""
    lib = lab_library()
    fn = lib.m3pc_debug_attention_bf16
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p] * 2
    g = torch.Generator(device="cuda").manual_seed(batch + L)
    qkv = torch.randn(batch, L, 1536, device="cuda", generator=g).to(torch.bfloat16)
    outs = []
    for kernel in (0, 1):
        O = torch.full((batch, L, 512), float("nan"), device="cuda", dtype=torch.bfloat16)
        rc = fn(qkv.data_ptr(), None, O.data_ptr(), batch, L, 0, kernel, C.c_void_p(torch.cuda.current_stream().cuda_stream), None)
        assert rc == 0, lib.m3pc_last_error()
        torch.cuda.synchronize()
        outs.append(O)
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    for b in (0, batch // 2, batch - 1):
        rows = qkv[b].double()
        q, k, v = rows[:, :512], rows[:, 512:1024], rows[:, 1024:]
        ref = torch.cat([torch.softmax(q[:, 128 * h:128 * h + 128] @ k[:, 128 * h:128 * h + 128].T / 128 ** 0.5, -1)
                         @ v[:, 128 * h:128 * h + 128] for h in range(4)], 1)
        assert float((outs[0][b].double() - ref).abs().max()) < 3e-2


@pytest.mark.parametrize("n,Lq,Lq2", [(2048, 1, 31), (300, 1, 31), (257, 2, 30), (512, 1, 20)])
def test_pipelined_mixed_query_attention_is_the_direct_kernel_bit_for_bit(n, Lq, Lq2):
    """attn_bf16_pipe_mix_kernel (critic mode's decoder: own + shared queries in one tile, the 79 batch-shared K|V rows resident in LDS,
    own rows by LDS-DMA) against attn_bf16_direct_kernel<4, 2, 4>: equal bits, and both against a float64 softmax."""
    lib = lab_library()
    fn = lib.m3pc_debug_attention_mix_bf16
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p]
    g = torch.Generator(device="cuda").manual_seed(n + Lq2)
    qown = torch.randn(n, Lq, 512, device="cuda", generator=g).to(torch.bfloat16)
    qsh = torch.randn(Lq2, 1536, device="cuda", generator=g).to(torch.bfloat16)
    kv = torch.randn(n, 49, 1024, device="cuda", generator=g).to(torch.bfloat16)
    qkvm = torch.randn(79, 1536, device="cuda", generator=g).to(torch.bfloat16)
    outs = []
    for kernel in (0, 1):
        O = torch.full((n, Lq + Lq2, 512), float("nan"), device="cuda", dtype=torch.bfloat16)
        rc = fn(qown.data_ptr(), qsh.data_ptr(), kv.data_ptr(), qkvm.data_ptr(), O.data_ptr(), n, Lq, Lq2, kernel,
                C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, lib.m3pc_last_error()
        torch.cuda.synchronize()
        outs.append(O)
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    for b in (0, n // 2, n - 1):
        q = torch.cat([qown[b], qsh[:, :512]], 0).double()
        k = torch.cat([kv[b, :, :512], qkvm[:, 512:1024]], 0).double()
        v = torch.cat([kv[b, :, 512:], qkvm[:, 1024:]], 0).double()
        ref = torch.cat([torch.softmax(q[:, 128 * h:128 * h + 128] @ k[:, 128 * h:128 * h + 128].T / 128 ** 0.5, -1)
                         @ v[:, 128 * h:128 * h + 128] for h in range(4)], 1)
        assert float((outs[0][b].double() - ref).abs().max()) < 3e-2

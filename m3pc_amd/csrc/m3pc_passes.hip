// The forward passes of libm3pc_hip.so: one transformer block (run_block: GEMM chains or the fused layer tail), the encoder, the
// full decoder + heads (forward_impl: omtm.forward, mtm_model.py:593-716), the exactly pruned decoder and the candidate pass
// (learner.py:288-316).  Host side; see m3pc_internal.h.
#include "m3pc_internal.h"

namespace m3pc {


// qkv_done: the previous layer's fused tail already wrote this layer's Q|K|V rows; next_qkv: prefix of the layer whose Q|K|V
// projection this layer's fused tail may compute (-> *next_qkv_done)
int run_block(m3pc_handle* h, const std::string& pfx, float* X, int batch, int L, int dt, hipStream_t st, bool ln1_done,
              int n_sh, const LnP* next_ln, bool* next_ln_done, bool x_dead,
              float* Xnext, int res_nshared, bool qkv_done, const std::string* next_qkv,
              bool* next_qkv_done, bool x_bf16) {
    // x_bf16 (run_encoder decides): X holds bf16 rows -- the residual stream of a bf16 pass all of whose layers take the fused tail
    const int d = h->d, ff = h->ff;
    const int rows = batch * L;
    const int es = (int)dtype_size(dt);
    (void)es;
    LnP ln;
    memset(&ln, 0, sizeof(ln));
    ln.X = X;
    ln.ldx = d;
    ln.rows = rows;
    ln.d = d;
    ln.g1 = W(h, pfx + ".norm1.weight").f;
    ln.b1 = W(h, pfx + ".norm1.bias").f;
    if (dt == DT_BF16)
        ln.Yb = (bf16_t*)h->Hn;
    else
        ln.Yf = (float*)h->Hn;
    // norm1 in front of the Q|K|V projection: folded into that GEMM's operand load in the few-row fp32 passes
    GemmP pqkv = gemm_basic(h->Hn, d, Wop(h, pfx + ".self_attn.in_proj_weight", dt), d, rows, 3 * d, d,
                            W(h, pfx + ".self_attn.in_proj_bias").f);
    gemm_out(pqkv, dt, h->QKV, 3 * d);
    if (!ln1_done && n_sh == 0) {
        GemmP t = pqkv;
        t.A = X;
        t.a_ln_g = ln.g1;
        t.a_ln_b = ln.b1;
        if (can_fold_ln(h, t, dt)) {
            pqkv = t;
            ln1_done = true;
        }
    }
    if (!ln1_done && !qkv_done) launch_layernorm(ln, st);  // the embedding kernel already wrote norm1(X) of the first layer
    if (n_sh > 0) {
        // First layer of a candidate pass: the first n_sh tokens are the same for every candidate (history), so
        // their norm1 rows and Q|K|V projections exist once (n_sh rows behind the compact per-candidate rows in
        // Hn / QKV, written by the embedding kernel) and only the L - n_sh candidate-specific rows go through the
        // big GEMM.  Attention still produces all L output rows per candidate: queries and keys are read from the
        // two segments (own rows first, then the shared ones; softmax is order-independent up to rounding).
        const int n_own = L - n_sh;
        const size_t es2 = dtype_size(dt);
        char* hn_sh = (char*)h->Hn + (size_t)batch * n_own * d * es2;
        char* qkv_sh = (char*)h->QKV + (size_t)batch * n_own * 3 * d * es2;
        {
            GemmP p = gemm_basic(h->Hn, d, Wop(h, pfx + ".self_attn.in_proj_weight", dt), d, batch * n_own, 3 * d, d,
                                 W(h, pfx + ".self_attn.in_proj_bias").f);
            gemm_out(p, dt, h->QKV, 3 * d);
            gemm(h, p, dt, st);
        }
        {
            GemmP p = gemm_basic(hn_sh, d, Wop(h, pfx + ".self_attn.in_proj_weight", dt), d, n_sh, 3 * d, d,
                                 W(h, pfx + ".self_attn.in_proj_bias").f);
            gemm_out(p, dt, qkv_sh, 3 * d);
            gemm(h, p, dt, st);
        }
        AttnP a;
        memset(&a, 0, sizeof(a));
        const char* q = (const char*)h->QKV;
        a.Q = q;
        a.q_bstride = (long long)n_own * 3 * d;
        a.ldq = 3 * d;
        a.Lq = n_own;
        a.orow1 = n_sh;
        a.Q2 = qkv_sh;
        a.ldq2 = 3 * d;
        a.Lq2 = n_sh;
        a.orow2 = 0;
        a.K1 = q + (size_t)d * es2;
        a.V1 = q + (size_t)2 * d * es2;
        a.kv1_bstride = (long long)n_own * 3 * d;
        a.ldkv1 = 3 * d;
        a.L1 = n_own;
        a.K2 = qkv_sh + (size_t)d * es2;
        a.V2 = qkv_sh + (size_t)2 * d * es2;
        a.ldkv2 = 3 * d;
        a.L2 = n_sh;
        a.O = h->O;
        a.o_bstride = (long long)L * d;
        a.ldo = d;
        a.batch = batch;
        a.n_head = h->nh;
        a.hd = h->hd;
        a.scale = 1.0f / sqrtf((float)h->hd);
        launch_attention(a, dt, st);
    } else {
    if (!qkv_done) gemm(h, pqkv, dt, st);
    {
        AttnP a;
        memset(&a, 0, sizeof(a));
        const char* q = (const char*)h->QKV;
        a.Q = q;
        a.q_bstride = (long long)L * 3 * d;
        a.ldq = 3 * d;
        a.K1 = q + (size_t)d * dtype_size(dt);
        a.V1 = q + (size_t)2 * d * dtype_size(dt);
        a.kv1_bstride = (long long)L * 3 * d;
        a.ldkv1 = 3 * d;
        a.L1 = L;
        a.O = h->O;
        a.o_bstride = (long long)L * d;
        a.ldo = d;
        a.batch = batch;
        a.n_head = h->nh;
        a.hd = h->hd;
        a.Lq = L;
        a.scale = 1.0f / sqrtf((float)h->hd);
        launch_attention(a, dt, st);
    }
    }
    // many-row bf16 passes: everything after the attention is one launch (block_fused.hip)
    static const bool no_fused = M3PC_ENV("M3PC_NO_BLOCK_FUSED") != nullptr;  // A/B switch
    static const bool no_split = M3PC_ENV("M3PC_NO_BLOCK_SPLIT") != nullptr;  // A/B switch
    const double step_rows = (double)rows * h->pass_scale;
    if (x_bf16 && (ln1_done == false && !qkv_done)) return fail(M3PC_EINVAL, "%s: norm1 of a bf16 residual stream has no kernel", pfx.c_str());
    if (dt == DT_BF16 && !no_fused && !no_split && step_rows < (double)FUSED_MIN_ROWS && step_rows >= (double)SPLIT_MIN_ROWS && !x_bf16 &&
        h->wstream.count(pfx) && !Xnext && !res_nshared && (long long)rows * block_split_n() * d <= h->R * 4LL * d) {
        // few tiles: four workgroups per tile + the reduce (which also applies the LayerNorm that consumes the block output)
        BlockP b;
        memset(&b, 0, sizeof(b));
        b.O = (const bf16_t*)h->O;
        b.ldo = d;
        b.M = rows;
        b.res = X;
        b.ldr = d;
        b.wstream = h->wstream[pfx];
        b.bo = W(h, pfx + ".self_attn.out_proj.bias").f;
        b.b1 = W(h, pfx + ".linear1.bias").f;
        b.b2 = W(h, pfx + ".linear2.bias").f;
        b.ln2_g = W(h, pfx + ".norm2.weight").f;
        b.ln2_b = W(h, pfx + ".norm2.bias").f;
        b.split = 1;
        b.Xout = (float*)h->F;
        b.ldx = d;
        const bool fuse_ln = next_ln && next_ln->Yb && !next_ln->Yf && !next_ln->g2 && next_ln->X == X && next_ln->xmap.rpg == 0 &&
                             next_ln->rows == rows;
        bool ok;
        {
            GemmTimer t(h, st, 2.0 * rows * ((double)d * d + 2.0 * d * ff), dt, 1);  // (algorithmic: the repeated out-proj is not counted)
            ok = launch_block_fused(b, st);
        }
        if (ok) {
            SplitReduceP r;
            memset(&r, 0, sizeof(r));
            r.slabs = (const float*)h->F;
            r.M = rows;
            r.ldx = d;
            if (!(fuse_ln && x_dead)) r.Xout = X;
            if (fuse_ln) {
                r.lnA_g = next_ln->g1;
                r.lnA_b = next_ln->b1;
                r.Hout = next_ln->Yb;
                r.ldh = d;
            }
            launch_block_split_reduce(r, st);
            if (next_ln_done) *next_ln_done = fuse_ln;
            if (next_qkv_done) *next_qkv_done = false;
            return check_launch(pfx.c_str());
        }
    }
    if (dt == DT_BF16 && !no_fused && step_rows >= (double)FUSED_MIN_ROWS && h->wstream.count(pfx)) {
        BlockP b;
        memset(&b, 0, sizeof(b));
        b.O = (const bf16_t*)h->O;
        b.ldo = d;
        b.M = rows;
        b.res = X;
        b.ldr = d;
        b.wstream = h->wstream[pfx];
        b.bo = W(h, pfx + ".self_attn.out_proj.bias").f;
        b.b1 = W(h, pfx + ".linear1.bias").f;
        b.b2 = W(h, pfx + ".linear2.bias").f;
        b.ln2_g = W(h, pfx + ".norm2.weight").f;
        b.ln2_b = W(h, pfx + ".norm2.bias").f;
        const bool fuse_ln = next_ln && next_ln->Yb && !next_ln->Yf && !next_ln->g2 && next_ln->X == (Xnext ? Xnext : X) &&
                             next_ln->xmap.rpg == 0 && next_ln->rows == rows;
        if (res_nshared > 0) {
            b.res_L = L;
            b.res_nshared = res_nshared;
        }
        b.x_bf16 = x_bf16 ? 1 : 0;
        static const bool no_qkv_fused = M3PC_ENV("M3PC_NO_QKV_FUSED") != nullptr;  // A/B switch
        const bool fuse_qkv = fuse_ln && next_qkv && !no_qkv_fused && next_ln->Yb == (bf16_t*)h->Hn &&
                              (size_t)rows * 3 * d * 2 < 0x7fffffffull;
        if (fuse_qkv) {  // norm1 of the next layer never leaves the kernel: its Q|K|V rows do
            b.lnA_g = next_ln->g1;
            b.lnA_b = next_ln->b1;
            b.QKVout = (bf16_t*)h->QKV;
            b.ldq = 3 * d;
            b.qkv_bytes = (unsigned)((size_t)rows * 3 * d * 2);
            b.bqkv = W(h, *next_qkv + ".self_attn.in_proj_bias").f;
        } else if (fuse_ln) {
            b.lnA_g = next_ln->g1;
            b.lnA_b = next_ln->b1;
            b.Hout = next_ln->Yb;
            b.ldh = d;
        }
        if (next_qkv_done) *next_qkv_done = fuse_qkv;
        if (!(fuse_ln && x_dead)) {
            b.Xout = Xnext ? Xnext : X;
            b.ldx = d;
        }
        if (h->stamp_log) {
            b.stamps = h->stamp_log + 64 * (h->stamp_i++ % h->stamp_cap);
            b.stamp_block = 37;
        }
        GemmTimer t(h, st, 2.0 * rows * ((double)d * d + 2.0 * d * ff + (fuse_qkv ? 3.0 * d * d : 0.0)), dt, 1);
        if (x_bf16 && !(fuse_ln && (fuse_qkv || x_dead)))
            return fail(M3PC_EINVAL, "%s: a bf16 residual stream needs the consumer of the block output inside the tail", pfx.c_str());
        if (launch_block_fused(b, st)) {
            if (next_ln_done) *next_ln_done = fuse_ln;
            return check_launch(pfx.c_str());
        }
        if (next_qkv_done) *next_qkv_done = false;
    }
    if (Xnext || res_nshared || x_bf16) return fail(M3PC_EINVAL, "%s: the fused layer tail did not take a pass set up for it", pfx.c_str());
    {
        GemmP p = gemm_basic(h->O, d, Wop(h, pfx + ".self_attn.out_proj.weight", dt), d, rows, d, d,
                             W(h, pfx + ".self_attn.out_proj.bias").f);
        p.res = X;
        p.ldr = d;
        gemm_out(p, DT_F32, X, d);
        ln.g1 = W(h, pfx + ".norm2.weight").f;
        ln.b1 = W(h, pfx + ".norm2.bias").f;
        // norm2: folded into linear1's operand load (few-row fp32), else on the split-K reduce of this GEMM, else a launch
        GemmP p1 = gemm_basic(h->Hn, d, Wop(h, pfx + ".linear1.weight", dt), d, rows, ff, d, W(h, pfx + ".linear1.bias").f);
        p1.gelu = 1;
        gemm_out(p1, dt, h->F, ff);
        GemmP t = p1;
        t.A = X;
        t.a_ln_g = ln.g1;
        t.a_ln_b = ln.b1;
        if (can_fold_ln(h, t, dt)) {
            gemm(h, p, dt, st);
            gemm(h, t, dt, st);
        } else {
            if (dt == DT_F32) {
                p.ln_g = ln.g1;
                p.ln_b = ln.b1;
                p.ln_out = ln.Yf;
            }
            if (!gemm(h, p, dt, st)) launch_layernorm(ln, st);
            gemm(h, p1, dt, st);
        }
    }
    {
        GemmP p = gemm_basic(h->F, ff, Wop(h, pfx + ".linear2.weight", dt), ff, rows, d, ff, W(h, pfx + ".linear2.bias").f);
        p.res = X;
        p.ldr = d;
        gemm_out(p, DT_F32, X, d);
        if (dt == DT_F32 && next_ln && next_ln->Yf && !next_ln->Yb && !next_ln->g2 && next_ln->X == X && next_ln->xmap.rpg == 0 &&
            next_ln->rows == rows) {
            p.ln_g = next_ln->g1;
            p.ln_b = next_ln->b1;
            p.ln_out = next_ln->Yf;
        }
        const int done = gemm(h, p, dt, st);
        if (next_ln_done) *next_ln_done = done != 0;
    }
    return check_launch(pfx.c_str());
}


int run_encoder(m3pc_handle* h, Plan* pl, const TokIn& in, int batch, int dt, hipStream_t st, bool bf16_out_only,
                int n_indep, int layer_from, int layer_to, PieceState* ln_state) {
    // first-layer pruning (run_block): whole 32-query tiles of shared tokens, bf16 candidate passes only
    int n_sh = 0;
    static const bool no_prune1 = M3PC_ENV("M3PC_NO_PRUNE1") != nullptr;  // A/B switch
    if (dt == DT_BF16 && batch >= 64 && h->d % 256 == 0 && h->d <= 1024 && n_indep >= 32 && !no_prune1)
        n_sh = (n_indep / 32) * 32;
    EmbedP e;
    memset(&e, 0, sizeof(e));
    e.n_indep = n_indep;
    e.n_sh = n_sh;
    e.Hb_sh = n_sh ? (bf16_t*)((char*)h->Hn + (size_t)batch * (pl->Le - n_sh) * h->d * 2) : nullptr;
    for (int k = 0; k < 4; ++k) {
        e.tok[k] = in.ptr[k];
        e.bstride[k] = in.bstride[k];
        e.wstride[k] = in.wstride[k];
        e.normalize[k] = in.normalize[k];
        e.mean[k] = h->tok_mean[k];
        e.stdv[k] = h->tok_std[k];
        e.WT[k] = h->WT[k];
        e.E[k] = h->Eenc[k];
        e.feat[k] = h->feat[k];
    }
    // history rows of the residual stream stored once (sequence 0) when the first layer's tail is the fused kernel: it reads
    // them there and writes the layer output to Y, which carries the stream through the remaining layers
    // (M3PC_SHARED_RES=1; measured on C2: embedding 33 -> 15 us, but every tile of the fused kernel then reads the same 66 KiB
    // and the step is 0.4 % SLOWER -- off)
    static const bool shared_res_on = M3PC_ENV("M3PC_SHARED_RES") != nullptr && M3PC_ENV("M3PC_NO_BLOCK_FUSED") == nullptr;
    const bool shared_res = n_sh > 0 && shared_res_on && (long long)batch * pl->Le >= 512 && h->wstream.count("encoder.layers.0") &&
                            bf16_out_only;
    // Round 6: the residual stream in bf16 between the layers (X: embedding -> layer tails; half of a tile's residual bytes in and
    // of its X'' bytes out; oracle/lowprec_study.py "bf16_res": the bf16 deviation delta x 0.95-1.25) -- when every layer of this
    // pass takes the full-tile fused tail with its consumer inside (next Q|K|V, or encoder.norm with X'' dead), so that nothing
    // but the tails ever reads X.  Decided from the pass's shape alone: the pieces of a pass enqueued layer by layer agree.
    static const bool no_xb16 = M3PC_ENV("M3PC_NO_BF16_RESIDUAL") != nullptr;  // A/B switch
    static const bool no_fused_env = M3PC_ENV("M3PC_NO_BLOCK_FUSED") != nullptr || M3PC_ENV("M3PC_NO_QKV_FUSED") != nullptr;
    bool xb16 = dt == DT_BF16 && bf16_out_only && !shared_res && !no_xb16 && !no_fused_env && block_fused_supported(h->d, h->ff) &&
                (double)batch * pl->Le * h->pass_scale >= (double)FUSED_MIN_ROWS && (size_t)batch * pl->Le * 3 * h->d * 2 < 0x7fffffffull &&
                (unsigned long long)batch * pl->Le * h->d * 2 < 0x80000000ull;
    for (int i = 0; i < h->dm.n_enc_layer && xb16; ++i) xb16 = h->wstream.count("encoder.layers." + std::to_string(i)) != 0;
    e.x_first_only = shared_res ? 1 : 0;
    if (xb16) e.Xb = (bf16_t*)h->X;
    e.widx = in.widx;
    e.tokmap = pl->d_tokmap;
    e.batch = batch;
    e.L = pl->Le;
    e.d = h->d;
    e.T = h->T;
    e.X = h->X;
    e.ln_g = W(h, "encoder.layers.0.norm1.weight").f;  // first layer's norm1 fused into the embedding
    e.ln_b = W(h, "encoder.layers.0.norm1.bias").f;
    if (dt == DT_BF16)
        e.Hb = (bf16_t*)h->Hn;
    else
        e.Hf = (float*)h->Hn;
    if (layer_from <= 0) launch_embed(e, st);
    LnP ln;
    memset(&ln, 0, sizeof(ln));
    ln.X = h->X;
    ln.ldx = h->d;
    ln.rows = batch * pl->Le;
    ln.d = h->d;
    ln.g1 = W(h, "encoder.norm.weight").f;
    ln.b1 = W(h, "encoder.norm.bias").f;
    if (bf16_out_only)
        ln.Yb = (bf16_t*)h->Z;  // the candidate pass consumes the encoder output only as a bf16 GEMM operand
    else
        ln.Yf = h->EncOut;
    bool ln_done = layer_from <= 0 || !ln_state ? true : ln_state->ln;  // norm1 of layer 0 comes from the embedding kernel
    bool qkv_done = layer_from <= 0 || !ln_state ? false : ln_state->qkv;
    const int nl = h->dm.n_enc_layer;
    float* Xs = shared_res && layer_from > 0 ? h->Y : h->X;  // where the residual stream lives
    for (int i = layer_from > 0 ? layer_from : 0; i < nl && i < layer_to; ++i) {
        float* Xn = shared_res && i == 0 ? h->Y : nullptr;
        LnP nxt = ln;  // what follows layer i on X: norm1 of layer i+1 (-> Hn) or encoder.norm (-> EncOut / Z)
        nxt.X = ln.X = Xn ? Xn : Xs;
        if (i + 1 < nl) {
            nxt.g1 = W(h, "encoder.layers." + std::to_string(i + 1) + ".norm1.weight").f;
            nxt.b1 = W(h, "encoder.layers." + std::to_string(i + 1) + ".norm1.bias").f;
            nxt.Yb = dt == DT_BF16 ? (bf16_t*)h->Hn : nullptr;
            nxt.Yf = dt == DT_F32 ? (float*)h->Hn : nullptr;
        }
        const bool l1 = ln_done;
        ln_done = false;
        const std::string nq = "encoder.layers." + std::to_string(i + 1);
        const bool q1 = qkv_done;
        qkv_done = false;
        CHK(run_block(h, "encoder.layers." + std::to_string(i), Xs, batch, pl->Le, dt, st, l1, i == 0 ? n_sh : 0, &nxt, &ln_done,
                      i + 1 == nl && bf16_out_only, Xn, Xn ? n_indep : 0, q1, i + 1 < nl ? &nq : nullptr, &qkv_done, xb16));
        if (Xn) Xs = Xn;
    }
    if (ln_state) {
        ln_state->ln = ln_done;
        ln_state->qkv = qkv_done;
    }
    if (layer_to < nl) return check_launch("encoder");
    ln.X = Xs;
    if (!ln_done) launch_layernorm(ln, st);
    return check_launch("encoder");
}

// decoder-embed of rows of one key: Y[cmap rows] = Z[amap rows] W_dec_k^T + E_dec_k[r % mod]
void dec_embed(m3pc_handle* h, int k, const void* Zop, RowMap amap, float* Yout, RowMap cmap, int M, int mod, int dt,
               hipStream_t st, const float* table) {
    const int d = h->d;
    const std::string kn = KEYN[k];
    GemmP p = gemm_basic(Zop, d, Wop(h, "decoder_embed_dict." + kn + ".weight", dt), d, M, d, d, nullptr);
    p.amap = amap;
    p.cmap = cmap;
    p.rowtab = table ? table : h->Edec[k];
    p.rt_mod = mod;
    p.rt_ld = d;
    gemm_out(p, DT_F32, Yout, d);
    gemm(h, p, dt, st);
}

// Full (un-pruned) decoder on `batch` sequences: Z (4T rows each, operand dtype) -> Y (fp32) after all layers.
int run_decoder_full(m3pc_handle* h, const void* Zop, int batch, int dt, hipStream_t st) {
    const int T = h->T, d = h->d;
    bool grouped = false;
    static const bool no_group = M3PC_ENV("M3PC_NO_GEMM_GROUP") != nullptr || M3PC_ENV("M3PC_NO_F32_DIRECT") != nullptr ||
                                 M3PC_ENV("M3PC_GEMM_VARIANT") != nullptr;  // A/B switches
    if (dt == DT_F32 && !no_group && h->allow_splitk) {  // few-row fp32 pass: the four modality GEMMs as one launch
        GemmP ps[4];
        for (int k = 0; k < 4; ++k) {
            RowMap m{T, 4 * T, k * T};
            ps[k] = gemm_basic(Zop, d, Wop(h, std::string("decoder_embed_dict.") + KEYN[k] + ".weight", dt), d, batch * T, d, d, nullptr);
            ps[k].amap = m;
            ps[k].cmap = m;
            ps[k].rowtab = h->Edec[k];
            ps[k].rt_mod = T;
            ps[k].rt_ld = d;
            gemm_out(ps[k], DT_F32, h->Y, d);
        }
        GemmTimer t(h, st, 4 * 2.0 * batch * T * (double)d * d, dt);
        grouped = launch_gemm_f32_direct_group(ps, 4, st);
    }
    for (int k = 0; k < 4 && !grouped; ++k) {
        RowMap m{T, 4 * T, k * T};
        dec_embed(h, k, Zop, m, h->Y, m, batch * T, T, dt, st);
    }
    for (int i = 0; i < h->dm.n_dec_layer; ++i) CHK(run_block(h, "decoder.layers." + std::to_string(i), h->Y, batch, 4 * T, dt, st));
    return check_launch("decoder");
}

int run_head_tail(m3pc_handle* h, int k, const void* ln_rows, int rows, float* out, int ldy, bool detok, int dt, hipStream_t st);

// Output head of key k (not actions) on `rows` logical rows of Ysrc selected by xmap:
// decoder.norm -> head LN -> Linear+GELU -> Linear(D_k) [-> de-tokenize]
int run_head(m3pc_handle* h, int k, const float* Ysrc, RowMap xmap, int rows, float* out, int ldy, bool detok, int dt,
             hipStream_t st) {
    const int d = h->d;
    const std::string kn = KEYN[k];
    LnP ln;
    memset(&ln, 0, sizeof(ln));
    ln.X = Ysrc;
    ln.ldx = d;
    ln.xmap = xmap;
    ln.rows = rows;
    ln.d = d;
    ln.g1 = W(h, "decoder.norm.weight").f;
    ln.b1 = W(h, "decoder.norm.bias").f;
    ln.g2 = W(h, "output_head_dict." + kn + ".0.weight").f;
    ln.b2 = W(h, "output_head_dict." + kn + ".0.bias").f;
    if (dt == DT_BF16)
        ln.Yb = (bf16_t*)h->Hn;
    else
        ln.Yf = (float*)h->Hn;
    launch_layernorm(ln, st);
    return run_head_tail(h, k, h->Hn, rows, out, ldy, detok, dt, st);
}

// ... from the head's LayerNorm output (rows, d) in the operand dtype on: Linear+GELU -> Linear(D_k) [-> de-tokenize]
int run_head_tail(m3pc_handle* h, int k, const void* ln_rows, int rows, float* out, int ldy, bool detok, int dt, hipStream_t st) {
    const int d = h->d;
    const std::string kn = KEYN[k];
    GemmP p = gemm_basic(ln_rows, d, Wop(h, "output_head_dict." + kn + ".1.weight", dt), d, rows, d, d,
                         W(h, "output_head_dict." + kn + ".1.bias").f);
    p.gelu = 1;
    // many-row bf16 passes: the gelu'd hidden rows cross HBM in bf16 and the last Linear runs on the matrix cores (round 6)
    static const bool no_mfma_head = M3PC_ENV("M3PC_NO_HEAD_OUT_MFMA") != nullptr;  // A/B switch
    // (chosen by the size of the WHOLE step, never by a shard's rows: sharded scores stay bit-identical to the unsharded ones)
    const long long step_rows = (long long)((double)rows * h->pass_scale + 0.5);
    const bool hb = dt == DT_BF16 && !no_mfma_head && head_out_mfma_covers((int)(step_rows < 0x7fffffff ? step_rows : 0x7fffffff), d, h->feat[k]);
    gemm_out(p, hb ? DT_BF16 : DT_F32, h->G, d);
    gemm(h, p, dt, st);
    HeadOutP ho;
    memset(&ho, 0, sizeof(ho));
    if (hb) ho.Xb = (const bf16_t*)h->G;
    else ho.X = h->G;
    ho.ldx = d;
    ho.rows = rows;
    ho.d = d;
    ho.D = h->feat[k];
    ho.W = W(h, "output_head_dict." + kn + ".3.weight").f;
    ho.b = W(h, "output_head_dict." + kn + ".3.bias").f;
    if (detok && h->tok_norm[k]) {
        ho.mean = h->tok_mean[k];
        ho.stdv = h->tok_std[k];
    }
    ho.Y = out;
    ho.ldy = ldy;
    launch_head_out(ho, st);
    return check_launch("head");
}

// Generic forward on `batch` sequences; outputs raw head values (no de-tokenization)
int forward_impl(m3pc_handle* h, Plan* pl, const TokIn& in, int batch, float* out_states, float* out_rewards,
                 float* out_returns, float* out_mu, float* out_std, int dt, hipStream_t st) {
    const int T = h->T, d = h->d;
    if ((long long)batch * 4 * T > h->R) return fail(M3PC_ENOMEM, "batch %d exceeds workspace (max_batch=%d)", batch, h->dm.max_batch);
    CHK(run_encoder(h, pl, in, batch, dt, st));
    GatherP g;
    memset(&g, 0, sizeof(g));
    g.Xe = h->EncOut;
    g.xe_bstride = (long long)pl->Le * d;
    g.table = h->mask_tokens;
    g.rowsrc = pl->d_dec_rowsrc;
    g.rows_per_batch = 4 * T;
    g.batch = batch;
    g.d = d;
    if (dt == DT_BF16)
        g.outb = (bf16_t*)h->Z;
    else
        g.out = (float*)h->Z;
    launch_gather_rows(g, st);
    CHK(run_decoder_full(h, h->Z, batch, dt, st));
    float* outs[4] = {out_states, nullptr, out_rewards, out_returns};
    for (int k = 0; k < 4; ++k) {
        if (k == M3PC_ACTIONS || !outs[k]) continue;
        RowMap m{T, 4 * T, k * T};
        CHK(run_head(h, k, h->Y, m, batch * T, outs[k], h->feat[k], false, dt, st));
    }
    if (out_mu && out_std) {
        // decoder.norm of the action rows, then the actor head: one launch when the head kernel can normalise its rows itself
        const bool fuse = actor_head_fuses_ln(d, h->A);
        if (!fuse) {
            LnP ln;
            memset(&ln, 0, sizeof(ln));
            ln.X = h->Y;
            ln.ldx = d;
            ln.xmap = RowMap{T, 4 * T, M3PC_ACTIONS * T};
            ln.rows = batch * T;
            ln.d = d;
            ln.g1 = W(h, "decoder.norm.weight").f;
            ln.b1 = W(h, "decoder.norm.bias").f;
            ln.Yf = h->G;
            launch_layernorm(ln, st);
        }
        ActorP a;
        memset(&a, 0, sizeof(a));
        a.X = fuse ? h->Y : h->G;
        a.ldx = d;
        if (fuse) {
            a.xmap = RowMap{T, 4 * T, M3PC_ACTIONS * T};
            a.ln_g = W(h, "decoder.norm.weight").f;
            a.ln_b = W(h, "decoder.norm.bias").f;
        }
        a.rows = batch * T;
        a.d = d;
        a.A = h->A;
        a.Wmu = W(h, "output_head_dict.actions.mu.weight").f;
        a.bmu = W(h, "output_head_dict.actions.mu.bias").f;
        a.Wls = W(h, "output_head_dict.actions.log_std.weight").f;
        a.bls = W(h, "output_head_dict.actions.log_std.bias").f;
        a.mu = out_mu;
        a.sd = out_std;
        launch_actor_head(a, st);
    }
    return check_launch("forward");
}


// The exactly pruned decoder (mtm_model.py:663-716 restricted to what the caller reads) behind an encoder pass over `n`
// sequences of plan `pl` (encoder output in Z [bf16] / EncOut [fp32]): decoder inputs and K|V of the un-masked tokens,
// the queries of set `q` (shared table rows when every query token is masked, per-sequence rows else), attention over
// own + masked keys, out-proj / FFN on the n * nq query rows, then
//   TAIL_HEADS: decoder.norm + the output head of each group's key -> h->pred[s] (n * grp, D_k) de-tokenised
//   TAIL_X:     the fp32 block output rows (n * nq, d) -> *xrows (the caller applies decoder.norm / the action head)
int pruned_decoder(m3pc_handle* h, Plan* pl, Plan::Query& q, SharedTables& tb, int n, int dt, hipStream_t st, int tail,
                   float** xrows) {
    const int d = h->d, hh = q.grp, Le = pl->Le, nq = q.nq;
    const size_t es = dtype_size(dt);
    CHK(ensure_edec(h, pl, st));
    float* Y1 = h->EncOut;  // (n*nq, d) decoder residual of the query tokens (EncOut is dead: Z/Y hold its uses)
    if (xrows) *xrows = Y1;
    // decoder inputs of the un-masked tokens
    const void* enc_op = dt == DT_BF16 ? h->Z : (const void*)h->EncOut;
    const std::string pfx = "decoder.layers.0";
    LnP ln;
    memset(&ln, 0, sizeof(ln));
    ln.X = h->Y;
    ln.ldx = d;
    ln.rows = n * Le;
    ln.d = d;
    ln.g1 = W(h, pfx + ".norm1.weight").f;
    ln.b1 = W(h, pfx + ".norm1.bias").f;
    if (dt == DT_BF16)
        ln.Yb = (bf16_t*)h->Hn;
    else
        ln.Yf = (float*)h->Hn;
    bool kv_done = false;
    static const bool no_kv_fused = M3PC_ENV("M3PC_NO_KV_FUSED") != nullptr || M3PC_ENV("M3PC_NO_BLOCK_FUSED") != nullptr;  // A/B switch
    static const bool no_fused_tail = M3PC_ENV("M3PC_NO_BLOCK_FUSED") != nullptr;  // A/B switch
    static const bool no_mix_prefix = M3PC_ENV("M3PC_NO_MIX_PREFIX") != nullptr;   // A/B switch
    const bool kv_fusable = dt == DT_BF16 && !no_kv_fused && (double)n * Le * h->pass_scale >= 512.0 && h->kvstream[0] &&
                            (pl->kept[0] || pl->kept[1]) && !pl->kept[2] && !pl->kept[3];
    // Some query tokens un-masked, as a prefix (Query::nu): the fused decoder input still serves K|V, the nu per-sequence
    // query rows get their decoder inputs / Q projection from few-row GEMMs of their own, and the fused tail takes their
    // residual rows from behind the shared table (many-row bf16 passes only: the choice goes by the size of the whole step)
    const bool mixp = kv_fusable && !q.all_masked && q.nu > 0 && !no_mix_prefix && !no_fused_tail && h->wstream.count(pfx) &&
                      (double)n * nq * h->pass_scale >= (double)FUSED_MIN_ROWS && (long long)nq + (long long)n * q.nu <= h->R &&
                      // (the fused tail's own limits for this layout, block_fused_accepts(): 32-bit buffer offsets of the residual
                      // table [shared rows | n nu per-sequence rows] and of the rows it writes -- decided HERE, before the pass is laid
                      // out for that kernel, so that a pass it would refuse takes the generic per-sequence path instead of failing)
                      (unsigned long long)(nq + (unsigned long long)n * q.nu) * d * 4 < 0xfffffff0ull &&
                      (unsigned long long)n * nq * d * 4 < 0x80000000ull;
    if (kv_fusable && (q.all_masked || mixp)) {
        // embedding, norm1 and the K|V projection in one launch (kv_fused_kernel): the fp32 rows Y are consumed by nothing
        // else when every scored token is masked
        KvFusedP kp;
        memset(&kp, 0, sizeof(kp));
        kp.Z = (const bf16_t*)h->Z;
        kp.ldz = d;
        int g = 0;
        for (int k = 0; k < 2; ++k) {
            if (!pl->kept[k]) continue;
            kp.M[g] = n * pl->kept[k];
            kp.map[g] = RowMap{pl->kept[k], Le, pl->enc_off[k]};
            kp.rowtab[g] = pl->edec_kept[k];
            kp.rt_mod[g] = pl->kept[k];
            kp.wstream[g] = h->kvstream[k];
            ++g;
        }
        kp.ln_g = ln.g1;
        kp.ln_b = ln.b1;
        kp.bkv = W(h, pfx + ".self_attn.in_proj_bias").f + d;
        kp.KV = (bf16_t*)h->QKV;
        kp.ldkv = 2 * d;
        kp.kv_bytes = (unsigned)((size_t)n * Le * 2 * d * 2);
        GemmTimer t(h, st, 2.0 * n * Le * (3.0 * d * d), dt, 2);
        kv_done = launch_kv_fused(kp, st);
    }
    if (!kv_done) {
    bool grouped = false;
    {   // few-row fp32 pass (re-score, pruned policy pass): the decoder-embedding GEMMs of the kept keys as ONE launch
        static const bool no_group = M3PC_ENV("M3PC_NO_GEMM_GROUP") != nullptr || M3PC_ENV("M3PC_NO_F32_DIRECT") != nullptr ||
                                     M3PC_ENV("M3PC_GEMM_VARIANT") != nullptr;  // A/B switches
        int nk = 0;
        GemmP ps[4];
        for (int k = 0; k < 4; ++k) {
            if (!pl->kept[k]) continue;
            RowMap mm{pl->kept[k], Le, pl->enc_off[k]};
            GemmP& g = ps[nk++];
            g = gemm_basic(enc_op, d, Wop(h, std::string("decoder_embed_dict.") + KEYN[k] + ".weight", dt), d, n * pl->kept[k], d, d, nullptr);
            g.amap = mm;
            g.cmap = mm;
            g.rowtab = pl->edec_kept[k] ? pl->edec_kept[k] : h->Edec[k];
            g.rt_mod = pl->kept[k];
            g.rt_ld = d;
            gemm_out(g, DT_F32, h->Y, d);
        }
        if (dt == DT_F32 && !no_group && h->allow_splitk && nk >= 2) {
            GemmTimer t(h, st, 2.0 * n * Le * (double)d * d, dt);
            grouped = launch_gemm_f32_direct_group(ps, nk, st);
        }
    }
    for (int k = 0; k < 4 && !grouped; ++k) {
        if (!pl->kept[k]) continue;
        RowMap mm{pl->kept[k], Le, pl->enc_off[k]};
        dec_embed(h, k, enc_op, mm, h->Y, mm, n * pl->kept[k], pl->kept[k], dt, st, pl->edec_kept[k]);
    }
    {  // K|V of the un-masked tokens: in_proj rows [d, 3d); norm1 rides on the operand load in the few-row fp32 pass
        const char* wkv = (const char*)Wop(h, pfx + ".self_attn.in_proj_weight", dt) + (size_t)d * d * es;
        GemmP p = gemm_basic(h->Hn, d, wkv, d, n * Le, 2 * d, d, W(h, pfx + ".self_attn.in_proj_bias").f + d);
        gemm_out(p, dt, h->QKV, 2 * d);
        GemmP t = p;
        t.A = h->Y;
        t.a_ln_g = ln.g1;
        t.a_ln_b = ln.b1;
        if (can_fold_ln(h, t, dt)) {
            p = t;
        } else {
            launch_layernorm(ln, st);
        }
        gemm(h, p, dt, st);
    }
    }
    // queries
    const void* Qp;
    long long q_bstride;
    int ldq;
    float* Yq_rows = nullptr;  // per-candidate residual rows (n*nq, d) when some scored token is un-masked
    char* kvu = (char*)h->QKV;
    char* qbuf = kvu + (size_t)n * Le * 2 * d * es;  // behind K|V in the same buffer
    float* Rcomb = nullptr;  // mixp: [shared residual table (nq rows)] [per-sequence residual rows of the nu un-masked queries (n nu)]
    if (mixp) {
        const int nu = q.nu, kq = q.nu_key;
        Rcomb = h->X;  // (the encoder residual stream is dead by now)
        float* Yu = Rcomb + (size_t)nq * d;
        HIPCHK(hipMemcpyAsync(Rcomb, tb.Yq, (size_t)nq * d * sizeof(float), hipMemcpyDeviceToDevice, st));
        // decoder inputs of the nu query tokens of every sequence: Z rows nu_enc0 .. of the sequence, the key's embedding
        RowMap am{nu, Le, q.nu_enc0};
        dec_embed(h, kq, enc_op, am, Yu, rowmap_identity(), n * nu, nu, dt, st, pl->edec_kept[kq] + (size_t)q.nu_kept0 * d);
        ln.X = Yu;
        ln.rows = n * nu;
        launch_layernorm(ln, st);
        GemmP p = gemm_basic(h->Hn, d, Wop(h, pfx + ".self_attn.in_proj_weight", dt), d, n * nu, d, d,
                             W(h, pfx + ".self_attn.in_proj_bias").f);
        gemm_out(p, dt, qbuf, d);
        gemm(h, p, dt, st);
        Qp = qbuf;
        q_bstride = (long long)nu * d;
        ldq = d;
    } else if (q.all_masked) {
        Qp = tb.QKVq;
        q_bstride = 0;
        ldq = 3 * d;
    } else {
        Yq_rows = h->X;  // encoder residual stream is dead by now
        GatherP g;
        memset(&g, 0, sizeof(g));
        g.Xe = h->Y;
        g.xe_bstride = (long long)Le * d;
        g.table = tb.Yall;
        g.rowsrc = q.d_q_rowsrc_mix;
        g.rows_per_batch = nq;
        g.batch = n;
        g.d = d;
        g.out = Yq_rows;
        launch_gather_rows(g, st);
        ln.X = Yq_rows;
        ln.rows = n * nq;
        launch_layernorm(ln, st);
        GemmP p = gemm_basic(h->Hn, d, Wop(h, pfx + ".self_attn.in_proj_weight", dt), d, n * nq, d, d,
                             W(h, pfx + ".self_attn.in_proj_bias").f);
        gemm_out(p, dt, qbuf, d);
        gemm(h, p, dt, st);
        Qp = qbuf;
        q_bstride = (long long)nq * d;
        ldq = d;
    }
    {
        AttnP at;
        memset(&at, 0, sizeof(at));
        at.Q = Qp;
        at.q_bstride = q_bstride;
        at.ldq = ldq;
        at.K1 = kvu;
        at.V1 = kvu + (size_t)d * es;
        at.kv1_bstride = (long long)Le * 2 * d;
        at.ldkv1 = 2 * d;
        at.L1 = Le;
        at.K2 = (const char*)tb.QKVm + (size_t)d * es;
        at.V2 = (const char*)tb.QKVm + (size_t)2 * d * es;
        at.ldkv2 = 3 * d;
        at.L2 = pl->Lm;
        at.O = h->O;
        at.o_bstride = (long long)nq * d;
        at.ldo = d;
        at.batch = n;
        at.n_head = h->nh;
        at.hd = h->hd;
        at.Lq = nq;
        at.scale = 1.0f / sqrtf((float)h->hd);
        if (mixp) {  // queries [0, nu) per sequence, the others from the shared table behind them
            at.Lq = q.nu;
            at.orow1 = 0;
            if (nq > q.nu) {
                at.Q2 = (const char*)tb.QKVq + (size_t)q.nu * 3 * d * es;
                at.ldq2 = 3 * d;
                at.Lq2 = nq - q.nu;
                at.orow2 = q.nu;
            }
        }
        if (dt == DT_BF16 && q.all_masked && tb.pre_m) {
            // the masked tokens' keys meet the same (shared) queries for every candidate: that block of the softmax
            // was reduced when the tables were built, only the candidate's own Le keys are visited here
            at.K2 = at.V2 = nullptr;
            at.L2 = 0;
            at.pre_m = tb.pre_m;
            at.pre_l = tb.pre_l;
            at.pre_O = tb.pre_O;
        }
        launch_attention(at, dt, st);
    }
    static const bool no_fused = M3PC_ENV("M3PC_NO_BLOCK_FUSED") != nullptr;  // A/B switch
    bool tail_done = false;
    static const bool no_split = M3PC_ENV("M3PC_NO_BLOCK_SPLIT") != nullptr;  // A/B switch
    const double step_rows = (double)n * nq * h->pass_scale;
    const bool tail_split = !no_split && step_rows < (double)FUSED_MIN_ROWS && step_rows >= (double)SPLIT_MIN_ROWS;  // (run_block)
    if (dt == DT_BF16 && !no_fused && (step_rows >= (double)FUSED_MIN_ROWS || tail_split) && h->wstream.count(pfx)) {
        // out-proj, norm2, FFN, decoder.norm and the two heads' LayerNorms in one launch (block_fused.hip): the rows of
        // head s land in the s-th block of n*h rows of Hn
        BlockP b;
        memset(&b, 0, sizeof(b));
        b.O = (const bf16_t*)h->O;
        b.ldo = d;
        b.M = n * nq;
        if (mixp) {
            b.rowtab = Rcomb;
            b.rt_mod = nq;
            b.res_nu = q.nu;
        } else if (q.all_masked) {
            b.rowtab = tb.Yq;
            b.rt_mod = nq;
        } else {
            b.res = Yq_rows;
            b.ldr = d;
        }
        b.wstream = h->wstream[pfx];
        b.bo = W(h, pfx + ".self_attn.out_proj.bias").f;
        b.b1 = W(h, pfx + ".linear1.bias").f;
        b.b2 = W(h, pfx + ".linear2.bias").f;
        b.ln2_g = W(h, pfx + ".norm2.weight").f;
        b.ln2_b = W(h, pfx + ".norm2.bias").f;
        if (tail == TAIL_HEADS) {
            b.lnA_g = W(h, "decoder.norm.weight").f;
            b.lnA_b = W(h, "decoder.norm.bias").f;
            for (int s = 0; s < 2; ++s) {  // (one group: LN_B[0] for every row; the kernel's tables still hold two)
                const int ks = q.qkeys[s < q.n_groups ? s : 0];
                b.lnB_g[s] = W(h, std::string("output_head_dict.") + KEYN[ks] + ".0.weight").f;
                b.lnB_b[s] = W(h, std::string("output_head_dict.") + KEYN[ks] + ".0.bias").f;
            }
            if (q.n_groups == 2) {
                b.out_mod = nq;
                b.out_grp = hh;
            }
        }
        // both scored keys have scalar heads (rtg_guiding: rewards, returns): the heads run inside the tail, on workgroups
        // that each own rows of one key; else the heads' LayerNorm rows go to Hn and the heads are launches of their own
        static const bool no_head_fused = M3PC_ENV("M3PC_NO_HEAD_FUSED") != nullptr;  // A/B switch
        const bool fuse_heads = tail == TAIL_HEADS && q.n_groups == 2 && !tail_split && !no_head_fused && q.qkeys[0] == M3PC_REWARDS && q.qkeys[1] == M3PC_RETURNS &&
                                h->feat[M3PC_REWARDS] == 1 && h->feat[M3PC_RETURNS] == 1;
        if (tail_split) {  // few tiles: four workgroups per tile, the LayerNorms on the reduce of their partials
            b.split = 1;
            b.Xout = (float*)h->F;
            b.ldx = d;
        } else if (fuse_heads) {
            for (int s = 0; s < 2; ++s) {
                const std::string hp = std::string("output_head_dict.") + KEYN[q.qkeys[s]];
                b.head_out[s] = h->pred[s];
                b.hb1[s] = W(h, hp + ".1.bias").f;
                b.hw2[s] = W(h, hp + ".3.weight").f;
                b.hb2[s] = W(h, hp + ".3.bias").f;
                if (h->tok_norm[q.qkeys[s]]) {
                    b.hmean[s] = h->tok_mean[q.qkeys[s]];
                    b.hstd[s] = h->tok_std[q.qkeys[s]];
                }
            }
        } else if (tail == TAIL_HEADS) {
            b.Hout = (bf16_t*)h->Hn;
            b.ldh = d;
        } else {
            b.Xout = Y1;
            b.ldx = d;
        }
        if (h->stamp_log) {
            b.stamps = h->stamp_log + 64 * (h->stamp_i++ % h->stamp_cap);
            b.stamp_block = 37;
        }
        bool ok;
        {
            GemmTimer t(h, st, 2.0 * n * nq * ((double)d * d + 2.0 * d * h->ff + (fuse_heads ? (double)d * d : 0.0)), dt, 1);
            ok = launch_block_fused(b, st);
        }
        if (ok && tail_split) {
            SplitReduceP r;
            memset(&r, 0, sizeof(r));
            r.slabs = (const float*)h->F;
            r.M = n * nq;
            if (tail == TAIL_HEADS) {
                r.lnA_g = b.lnA_g;
                r.lnA_b = b.lnA_b;
                for (int s = 0; s < 2; ++s) {
                    r.lnB_g[s] = b.lnB_g[s];
                    r.lnB_b[s] = b.lnB_b[s];
                }
                r.out_mod = b.out_mod;
                r.out_grp = b.out_grp;
                r.Hout = (bf16_t*)h->Hn;
                r.ldh = d;
            } else {
                r.Xout = Y1;
                r.ldx = d;
            }
            launch_block_split_reduce(r, st);
        }
        if (ok && !fuse_heads && tail == TAIL_HEADS) {
            for (int s = 0; s < q.n_groups; ++s)
                CHK(run_head_tail(h, q.qkeys[s], (const char*)h->Hn + (size_t)s * n * hh * d * es, n * hh, h->pred[s],
                                  h->feat[q.qkeys[s]], true, dt, st));
        }
        tail_done = ok;
    }
    if (!tail_done && mixp) return fail(M3PC_EINVAL, "pruned_decoder: the fused layer tail did not take a pass set up for it");
    if (!tail_done) {
    {
        GemmP p = gemm_basic(h->O, d, Wop(h, pfx + ".self_attn.out_proj.weight", dt), d, n * nq, d, d,
                             W(h, pfx + ".self_attn.out_proj.bias").f);
        if (q.all_masked) {
            p.rowtab = tb.Yq;
            p.rt_mod = nq;
            p.rt_ld = d;
        } else {
            p.res = Yq_rows;
            p.ldr = d;
        }
        gemm_out(p, DT_F32, Y1, d);
        ln.X = Y1;
        ln.rows = n * nq;
        ln.g1 = W(h, pfx + ".norm2.weight").f;
        ln.b1 = W(h, pfx + ".norm2.bias").f;
        GemmP p1 = gemm_basic(h->Hn, d, Wop(h, pfx + ".linear1.weight", dt), d, n * nq, h->ff, d, W(h, pfx + ".linear1.bias").f);
        p1.gelu = 1;
        gemm_out(p1, dt, h->F, h->ff);
        GemmP t = p1;
        t.A = Y1;
        t.a_ln_g = ln.g1;
        t.a_ln_b = ln.b1;
        if (can_fold_ln(h, t, dt)) {  // re-score: norm2 rides on linear1's operand load ...
            gemm(h, p, dt, st);
            gemm(h, t, dt, st);
        } else {
            if (dt == DT_F32) {       // ... or on the split-K reduce when there is one
                p.ln_g = ln.g1;
                p.ln_b = ln.b1;
                p.ln_out = ln.Yf;
            }
            if (!gemm(h, p, dt, st)) launch_layernorm(ln, st);
            gemm(h, p1, dt, st);
        }
    }
    {
        GemmP p = gemm_basic(h->F, h->ff, Wop(h, pfx + ".linear2.weight", dt), h->ff, n * nq, d, h->ff, W(h, pfx + ".linear2.bias").f);
        p.res = Y1;
        p.ldr = d;
        gemm_out(p, DT_F32, Y1, d);
        gemm(h, p, dt, st);
    }
    // heads of the scored keys -> pred[s] (n*grp, D_k), de-tokenized
    bool heads_done = false;
    {   // few-row fp32 pass with scalar heads only (the re-score of an rtg_guiding step): both heads in ONE launch
        static const bool no_head_f32 = M3PC_ENV("M3PC_NO_HEAD_F32_FUSED") != nullptr || M3PC_ENV("M3PC_NO_F32_DIRECT") != nullptr;  // A/B switch
        bool scalar = tail == TAIL_HEADS && dt == DT_F32 && h->allow_splitk && !no_head_f32 && h->cur && h->cur->head_part &&
                      n * hh <= h->cur->head_rows && q.n_groups >= 1 && q.n_groups <= 2;
        for (int s = 0; s < q.n_groups && scalar; ++s) scalar = h->feat[q.qkeys[s]] == 1 && q.qkeys[s] != M3PC_ACTIONS;
        if (scalar) {
            HeadFusedP hp;
            memset(&hp, 0, sizeof(hp));
            hp.X = Y1;
            hp.ldx = d;
            hp.rows = n * hh;
            hp.d = d;
            hp.n_heads = q.n_groups;
            hp.grp = hh;
            hp.row_mod = nq;
            hp.gA = W(h, "decoder.norm.weight").f;
            hp.bA = W(h, "decoder.norm.bias").f;
            for (int s = 0; s < q.n_groups; ++s) {
                const std::string hn = std::string("output_head_dict.") + KEYN[q.qkeys[s]];
                hp.gB[s] = W(h, hn + ".0.weight").f;
                hp.bB[s] = W(h, hn + ".0.bias").f;
                hp.W1[s] = W(h, hn + ".1.weight").f;
                hp.b1[s] = W(h, hn + ".1.bias").f;
                hp.w2[s] = W(h, hn + ".3.weight").f;
                hp.b2[s] = W(h, hn + ".3.bias").f;
                if (h->tok_norm[q.qkeys[s]]) {
                    hp.mean[s] = h->tok_mean[q.qkeys[s]];
                    hp.stdv[s] = h->tok_std[q.qkeys[s]];
                }
                hp.out[s] = h->pred[s];
            }
            hp.part = h->cur->head_part;
            hp.ticket = h->cur->head_ticket;
            GemmTimer t(h, st, q.n_groups * 2.0 * n * hh * (double)d * d, dt);
            heads_done = launch_head_f32_fused(hp, st);
        }
    }
    for (int s = 0; s < q.n_groups && tail == TAIL_HEADS && !heads_done; ++s) {
        RowMap xm{hh, nq, s * hh};
        CHK(run_head(h, q.qkeys[s], Y1, xm, n * hh, h->pred[s], h->feat[q.qkeys[s]], true, dt, st));
    }
    }
    const int rc = check_launch("pruned_decoder");
    if (rc != 0 && h->cur && h->cur->head_ticket)  // (an error behind the fused heads' launch: no count may survive in its tickets)
        (void)hipMemsetAsync(h->cur->head_ticket, 0, (size_t)2 * ((h->cur->head_rows + 31) / 32) * sizeof(int), st);
    return rc;
}

// ---------------------------------------------------------------------------------- candidate pass
// widx (optional, device (n,)): candidate c belongs to history window widx[c] of states (., T, S) / rewards (., T, 1);
// without it all candidates share window 0 and the history tokens are computed once (first-layer sharing).
// stage_from / stage_to / ln_state: the pass can be enqueued in pieces -- stage k < n_enc_layer is encoder layer k (the
// embedding goes with stage 0), stage n_enc_layer everything behind the encoder -- so that the pieces of two candidate halves
// can be enqueued alternately (m3pc_plan_step)
int candidate_pass(m3pc_handle* h, const m3pc_plan_args* a, const float* states, const float* rewards, int n,
                   const float* sample_actions, float* expect_return, float* pred_rewards, float* pred_boot, int dt,
                   hipStream_t st, const int* widx, int stage_from, int stage_to, PieceState* ln_state) {
    const int T = h->T, d = h->d, hh = a->horizon, idx = T - hh;
    const size_t es = dtype_size(dt);
    struct ScaleScope {
        m3pc_handle* h;
        ~ScaleScope() { h->pass_scale = 1.0; }
    } scale_scope{h};
    h->pass_scale = !widx && a->n_total > n ? (double)a->n_total / (double)n : 1.0;
    Plan* pl = nullptr;
    CHK(get_mask_plan(h, 1, idx, &pl));  // fd mask (finetune_omtm/masks.py:30-44)
    const int qi = a->mode == M3PC_MODE_RTG ? 0 : 1;
    CHK(build_query(h, pl, qi, hh));
    CHK(build_tables(h, pl, qi, dt, st));
    Plan::Query& q = pl->query[qi];
    SharedTables& tb = q.tab[dt];
    const int Le = pl->Le, nq = q.nq;
    if ((long long)n * Le > h->R || (long long)n * nq > h->R) return fail(M3PC_ENOMEM, "n_count %d exceeds workspace", n);

    TokIn in;
    memset(&in, 0, sizeof(in));
    in.ptr[M3PC_STATES] = states;
    in.normalize[M3PC_STATES] = h->tok_norm[M3PC_STATES];
    in.ptr[M3PC_ACTIONS] = h->cand;
    in.bstride[M3PC_ACTIONS] = (long long)T * h->A;
    in.normalize[M3PC_ACTIONS] = h->tok_norm[M3PC_ACTIONS];
    in.ptr[M3PC_REWARDS] = rewards;
    in.ptr[M3PC_RETURNS] = h->rtok;
    in.widx = widx;
    in.wstride[M3PC_STATES] = (long long)T * h->S;
    in.wstride[M3PC_REWARDS] = T;
    // encoder order is states 0..idx, actions 0..T-1: everything before actions[idx] is history, shared by all candidates
    // of one window
    const int nl_enc = h->dm.n_enc_layer;
    if (stage_from < nl_enc) CHK(run_encoder(h, pl, in, n, dt, st, dt == DT_BF16, widx ? 0 : (idx + 1) + idx, stage_from, stage_to, ln_state));
    if (stage_to <= nl_enc) return check_launch("candidate_pass");

    CHK(pruned_decoder(h, pl, q, tb, n, dt, st, TAIL_HEADS, nullptr));
    const float* rw;
    const float* boot;
    float boot_scale;
    if (a->mode == M3PC_MODE_RTG) {
        rw = h->pred[0];
        boot = h->pred[1];
        boot_scale = 1000.0f;  // learner.py:305
    } else {
        if (!h->critic_set) return fail(M3PC_ESTATE, "critic weights not set");
        CriticP c;
        memset(&c, 0, sizeof(c));
        c.states = h->pred[0];
        c.actions = sample_actions;
        c.rows = n * hh;
        c.S = h->S;
        c.A = h->A;
        c.hidden = h->dm.critic_hidden;
        c.om = h->c_om;
        c.os = h->c_os;
        for (int i = 0; i < 2; ++i) {
            c.W1T[i] = h->cW1T[i];
            c.b1[i] = h->cb1[i];
            c.W2T[i] = h->cW2T[i];
            c.b2[i] = h->cb2[i];
            c.W3[i] = h->cW3[i];
            c.b3[i] = h->cb3[i];
            c.W1F[i] = h->cW1F[i];
            c.W2F[i] = h->cW2F[i];
        }
        c.q = h->qv;
        launch_critic(c, st);
        rw = h->pred[1];
        boot = h->qv;
        boot_scale = 1.0f;
    }
    ScoreP sc;
    memset(&sc, 0, sizeof(sc));
    sc.rewards = rw;
    sc.boot = boot;
    sc.n = n;
    sc.h = hh;
    sc.boot_scale = boot_scale;
    sc.gamma = (float)a->discount;
    sc.lmbda = a->lmbda;
    sc.expect_return = expect_return;
    sc.boot_out = pred_boot;
    sc.scatter_index = h->score_scatter_index;  // (m3pc_rescore_listed: the scores also go straight to their candidates' slots)
    sc.scatter_out = h->score_scatter_out;
    launch_score(sc, st);
    if (pred_rewards) HIPCHK(hipMemcpyAsync(pred_rewards, rw, (size_t)n * hh * sizeof(float), hipMemcpyDeviceToDevice, st));
    return check_launch("candidate_pass");
}


}  // namespace m3pc

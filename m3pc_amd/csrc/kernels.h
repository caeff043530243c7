// Internal launcher interface of libm3pc_hip.so (gfx950 only).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Environment A/B switches (M3PC_NO_*, M3PC_GEMM_VARIANT, ...) exist in the LAB build only (libm3pc_hip_lab.so,
// `python -m m3pc_amd.build --lab`): the product library never reads the environment.
#ifdef M3PC_LAB
#define M3PC_ENV(name) getenv(name)
#else
#define M3PC_ENV(name) ((const char*)nullptr)
#endif

namespace m3pc {

typedef __bf16 bf16_t;

enum DType { DT_F32 = 0, DT_BF16 = 1 };

static inline size_t dtype_size(int dt) { return dt == DT_F32 ? 4 : 2; }

// Row remap used by GEMM / LayerNorm operands whose logical rows are a strided subset of a
// larger tensor: physical_row = (r / rpg) * gstride + (r % rpg) + off.   rpg == 0: identity.
// Cache policy of activation rows that are written once and read once (the residual stream, Q|K|V, K|V, attention outputs: ~2 GB
// of a plan step's HBM traffic): 0 = default (what ships), 2 = nt (non-temporal).  Round 6 measured nt, same box, three runs each
// (tools/ab_libs.sh on -DM3PC_STREAM_AUX builds, profiles/r06_ab_nt_streams.txt): on the fused kernels' residual pieces and row
// stores +0.6 %; on those plus the attention kernels' own-row pieces / stores and the embedding's stores +0.4 % (one pair of three
// the other way); with the fused tail's O / Z fragment LOADS and LayerNorm-row stores too -4.5 % (a tile alone 149 -> 165 us: rows
// another kernel has just written are in the caches, and an nt load does not take them from there).  Inside the noise: not adopted.
#ifndef M3PC_STREAM_AUX
#define M3PC_STREAM_AUX 0
#endif
template <typename V>
__device__ __forceinline__ V stream_load(const V* p) { return M3PC_STREAM_AUX ? __builtin_nontemporal_load(p) : *p; }
template <typename V>
__device__ __forceinline__ void stream_store(V v, V* p) {
    if (M3PC_STREAM_AUX) __builtin_nontemporal_store(v, p);
    else *p = v;
}

struct RowMap {
    int rpg;
    int gstride;
    int off;
};
static inline RowMap rowmap_identity() { return RowMap{0, 0, 0}; }

// C[M,N] = epilogue(A[M,K] * W[N,K]^T)
//   v = acc + bias[col] + rowtab[(r % rt_mod) * rt_ld + col];  v = gelu(v);  v += res[row_c * ldr + col]
// A and W share the operand dtype (fp32 -> v_mfma_f32_32x32x2_f32, bf16 -> v_mfma_f32_32x32x16_bf16),
// accumulation is fp32.  N % 64 == 0, K*sizeof(T) % 128 == 0.
struct GemmP {
    const void* A;
    int lda;
    RowMap amap;
    const void* W;
    int ldw;
    int M, N, K;
    const float* bias;
    const float* rowtab;
    int rt_mod, rt_ld;
    int gelu;
    const float* res;
    int ldr;
    float* Cf;     // fp32 output (optional)
    bf16_t* Cb;    // bf16 output (optional)
    int ldc;
    RowMap cmap;   // applies to res, Cf, Cb
    float* ws;     // split-K slab workspace (optional): few-row fp32 GEMMs split K over blocks
    long long ws_bytes;
    int a_padded;  // != 0: at least 127 rows (lda each) of readable memory follow the last row of A (the handle's
                   // workspaces: bf16 rows in fp32-sized buffers).  gemm_line.hip then runs a ragged last tile on its
                   // persistent path, whose DMA reads rows past M (never stored); 0: ragged M takes the clamped path
    int variant;   // 0 = default kernel selection; other values pick experimental configurations
    int peel;      // set by launch_gemm_glds: rows >= peel are covered by small tiles (0 = off); callers leave it 0
    // optional: the LayerNorm that consumes C (fp32, N % 256 == 0, identity cmap).  When the launcher takes the
    // split-K route its row-wise reduce also writes ln_out = LayerNorm(C row) and launch_gemm returns 1; otherwise
    // it returns 0 and the caller launches the LayerNorm itself.
    const float* ln_g;
    const float* ln_b;
    float* ln_out;
    // optional (few-row fp32 kernel only, K == the row length, K % 256 == 0): A is LayerNorm(A rows) * a_ln_g + a_ln_b,
    // normalised on the fly while the operand is loaded (the arithmetic of layernorm_vec_kernel, bit for bit)
    const float* a_ln_g;
    const float* a_ln_b;
};
// The scalar output heads (D_k == 1) of a few-row fp32 pass in one launch (gemm_f32_direct.hip: head_f32_fused_kernel):
// out[s][r] = w2[s] . gelu(W1[s] LN_B[s](LN_A(x)) + b1[s]) + b2[s] [* stdv + mean], x = row (r / grp) * row_mod + s * grp + r % grp of X
struct HeadFusedP {
    const float* X;
    int ldx;
    int rows, d;          // rows per head; d = model width (512)
    int n_heads;          // 1 or 2 (blockIdx.y)
    int grp, row_mod;     // physical row of (head s, row r): (r / grp) * row_mod + s * grp + r % grp
    const float *gA, *bA; // decoder.norm
    const float *gB[2], *bB[2];  // the head's LayerNorm
    const float *W1[2], *b1[2];  // Linear(d, d)
    const float *w2[2], *b2[2];  // Linear(d, 1)
    const float *mean[2], *stdv[2];  // optional: de-tokenise
    float* out[2];        // (rows,)
    float* part;          // scratch: n_heads * rows * (d / 32) floats
    int* ticket;          // scratch: n_heads * ceil(rows / 32) ints, zero before the first launch (the kernel leaves them zero)
};
bool launch_head_f32_fused(const HeadFusedP& p, hipStream_t st);  // false: not covered
int launch_gemm(const GemmP& p, int dtype, hipStream_t st);
bool gemm_f32_direct_covers(const GemmP& p);             // would launch_gemm_f32_direct take this problem (incl. a_ln_*)?
bool launch_gemm_f32_direct_group(const GemmP* ps, int n, hipStream_t st);  // n <= 4 covered problems in ONE launch
bool launch_gemm_ring(const GemmP& p, hipStream_t st);  // bf16, many rows: 256x256 tile, 4-slot LDS-DMA ring (gemm_ring.hip)
bool launch_gemm_persist(const GemmP& p, hipStream_t st);  // bf16, many rows: persistent 256x256 tiles (gemm_persist.hip)
bool launch_gemm_rs(const GemmP& p, hipStream_t st);    // bf16, many rows: register-staged 128x128 ring (gemm_rs.hip)
bool launch_gemm_big(const GemmP& p, hipStream_t st);   // bf16, many rows, K >= 1024: 256x256 tile, one wave per SIMD (gemm_big.hip)
bool launch_gemm_line(const GemmP& p, int bt, hipStream_t st);  // bf16, many rows: whole-cache-line DMA pieces, 5-unit ring (gemm_line.hip)
bool launch_gemm_glds(const GemmP& p, hipStream_t st);  // bf16, many rows: direct-to-LDS staging (gemm_glds.hip)
void read_big_probe(long long out[4]);                   // debug: gemm_big.hip phase timers
void read_clock_probe(long long out[2]);                // debug: {shader clocks, 100-MHz ticks} of one ring workgroup
bool launch_gemm_f32_direct(const GemmP& p, hipStream_t st);  // fp32, few rows: in-workgroup K split (gemm_f32_direct.hip)

// The tail of one pre-LN transformer layer after its attention, as one launch (block_fused.hip; d = 512, ff = 2048,
// bf16 operands):  X' = res + bo + O Wo^T;  X'' = X' + b2 + gelu(LN2(X') W1^T + b1) W2^T;
// Xout = X'' (fp32, optional);  Hout = bf16(LN_B[sel]?(LN_A(X''))).
struct BlockP {
    const bf16_t* O;        // (M, d) attention output rows
    int ldo;
    int M;
    const float* res;       // (M, d) residual rows (fp32), or
    int ldr;
    int res_L, res_nshared; // res_L > 0: rows come in sequences of res_L; the first res_nshared rows of every sequence are the
                            // same and are read from sequence 0 (the history tokens of a candidate pass: stored once)
    const float* rowtab;    // (rt_mod, d): row r takes rowtab[r % rt_mod] as its residual (res unused)
    int rt_mod;
    int res_nu;             // > 0 (with rowtab): rows with w = r % rt_mod < res_nu have residual rows of their own, stored behind the
                            // table: rowtab[rt_mod + (r / rt_mod) * res_nu + w]
    const bf16_t* wstream;  // packed weight fragments of the layer (launch_pack_block_stream)
    const float* bo;        // out_proj.bias (d)
    const float* b1;        // linear1.bias (ff)
    const float* b2;        // linear2.bias (d)
    const float* ln2_g;     // norm2
    const float* ln2_b;
    float* Xout;            // optional; may alias res
    int ldx;
    int x_bf16;             // round 6: res and Xout are bf16 rows (ldr / ldx in elements; res / Xout are bf16_t* behind the casts): the
                            // residual stream crosses HBM between the encoder layers in bf16 -- half of a tile's residual bytes in,
                            // half of its X'' bytes out (oracle/lowprec_study.py "bf16_res": delta x 0.95-1.25).  Plain and
                            // next-Q|K|V forms only (no rowtab, no split, no heads, no shared leading rows)
    const float* lnA_g;     // LayerNorm of the block output (next block's norm1 / the stack's final norm)
    const float* lnA_b;
    const float* lnB_g[2];  // optional second LayerNorm on top (an output head's norm), per row group
    const float* lnB_b[2];
    int out_mod, out_grp;   // out_mod > 0: row r belongs to group s = (r % out_mod) / out_grp (out_mod == 2 out_grp); its
                            // Hout row is s * (M / out_mod) * out_grp + (r / out_mod) * out_grp + r % out_grp and LN_B[s] applies
    bf16_t* Hout;           // optional
    int ldh;
    bf16_t* QKVout;         // optional, instead of Hout: (M, ldq) rows of LN_A(X'') Wqkv^T + bqkv -- the NEXT layer's in_proj; the
    int ldq;                // stream then carries that layer's in_proj fragments behind the layer's own (launch_pack_block_qkv)
    unsigned qkv_bytes;     // size of the QKVout buffer from its base (< 2 GiB: stores go through a buffer resource)
    const float* bqkv;      // in_proj_bias of the next layer (3 d)
    // optional, instead of Hout (out_mod / out_grp and lnB as for Hout): the two row groups' scalar output heads,
    //     head_out[s][i] = detok(w2_s . gelu(W1_s LN_B[s](LN_A(X'')) + b1_s) + b2_s)   for the i-th row of group s;
    // W1_s rides in the stream behind the layer's own fragments (launch_pack_block_heads)
    float* head_out[2];
    const float* hb1[2];    // Linear(512,512) bias (d)
    const float* hw2[2];    // Linear(512,1) weight row (d)
    const float* hb2[2];    // its bias (1)
    const float* hmean[2];  // tokenizer mean / std of the key (1 each), or null: no de-tokenisation
    const float* hstd[2];
    int split;              // 1: four workgroups per 128-row tile, each a quarter of the FFN's hidden units; fp32 partials go to the
                            // four slabs of M rows behind Xout (M * 512 floats each), LayerNorms / Hout are launch_block_split_reduce's
    int variant;            // 0 = product kernel; 1, 2: timing experiments (tools/block_bench.py)
    long long* stamps;      // optional (4 waves, 16) shader-clock phase stamps of workgroup stamp_block
    int stamp_block;
};
bool block_fused_supported(int d, int ff);
size_t block_stream_bytes();
void launch_pack_block_stream(const bf16_t* Wo, const bf16_t* W1, const bf16_t* W2, bf16_t* out, hipStream_t st);
void launch_pack_block_qkv(const bf16_t* Wqkv_next, bf16_t* out, hipStream_t st);
void launch_pack_block_heads(const bf16_t* Wh0, const bf16_t* Wh1, bf16_t* out, hipStream_t st);  // or: the two scalar heads' Linear(512,512)  // behind them: the next layer's in_proj
bool launch_block_fused(const BlockP& p, hipStream_t st);  // false: arguments not covered (caller takes the unfused path)
bool block_fused_accepts(const BlockP& p);                 // the same checks without the launch
struct SplitReduceP {       // behind launch_block_fused(split = 1): x = sum of the slabs -> Xout; LN_B?(LN_A(x)) -> Hout
    const float* slabs;     // (block_split_n(), M, 512) fp32
    int M;
    float* Xout;            // optional
    int ldx;
    const float* lnA_g;
    const float* lnA_b;
    const float* lnB_g[2];  // optional, per row group (out_mod / out_grp as BlockP)
    const float* lnB_b[2];
    int out_mod, out_grp;
    bf16_t* Hout;           // optional
    int ldh;
};
void launch_block_split_reduce(const SplitReduceP& p, hipStream_t st);
int block_split_n();

// Decoder input of the un-masked tokens as one launch (block_fused.hip: kv_fused_kernel; d = 512, bf16 operands):
//     y = Z W_k^T + rowtab_k[r % rt_mod]      decoder embedding of key k (mtm_model.py:665-676)
//     K|V = LayerNorm(y) Wkv^T + bkv          norm1 + the K|V rows of the decoder layer's in_proj (bf16 out)
// Two row groups (keys) per launch, each with its own embedding weights; tiles are key-pure.
struct KvFusedP {
    const bf16_t* Z;          // encoder output rows
    int ldz;
    int M[2];                 // rows of each group (0: group absent)
    RowMap map[2];            // group row -> row of Z and of KV
    const float* rowtab[2];   // (rt_mod, d) fp32
    int rt_mod[2];
    const bf16_t* wstream[2]; // launch_pack_kv_stream: embedding fragments, then the K|V fragments
    const float* ln_g;
    const float* ln_b;
    const float* bkv;         // (2 d)
    bf16_t* KV;               // (., 2 d) rows
    int ldkv;
    unsigned kv_bytes;        // size of the KV buffer in bytes (< 2 GiB): stores past it are dropped
    long long* stamps;        // optional (4 waves, 16) shader-clock phase stamps of workgroup stamp_block
    int stamp_block;
};
size_t kv_stream_bytes();
void launch_pack_kv_stream(const bf16_t* Wemb, const bf16_t* Wkv, bf16_t* out, hipStream_t st);
bool launch_kv_fused(const KvFusedP& p, hipStream_t st);  // false: arguments not covered

// LayerNorm over the last dim (eps 1e-5), one wave per row; optional second LayerNorm applied to
// the result (decoder.norm followed by an output head's LayerNorm).  d <= 1024, d % 64 == 0.
struct LnP {
    const float* X;
    int ldx;
    RowMap xmap;
    int rows, d;
    const float* g1;
    const float* b1;
    const float* g2;  // optional
    const float* b2;
    float* Yf;        // optional fp32 out (contiguous rows, ld = d)
    bf16_t* Yb;       // optional bf16 out
};
void launch_layernorm(const LnP& p, hipStream_t st);

// softmax(Q K^T * scale) V for one (batch, head) per block-column.  Keys/values come from up to two
// segments: seg 1 is per batch element, seg 2 is shared by the batch (bstride 0 allowed anywhere).
struct AttnP {
    const void* Q;
    long long q_bstride;
    int ldq;
    const void* K1;
    const void* V1;
    long long kv1_bstride;
    int ldkv1;
    int L1;
    const void* K2;
    const void* V2;
    int ldkv2;
    int L2;
    void* O;
    long long o_bstride;
    int ldo;
    int batch, n_head, hd, Lq;
    float scale;
    // optional second query segment shared by the batch (bf16 kernel only): query slots [0, Lq) come from Q,
    // slots [ceil32(Lq), ceil32(Lq) + Lq2) from Q2; outputs go to rows orow1 + i / orow2 + i of O[b]
    const void* Q2;
    int ldq2, Lq2;
    int orow1, orow2;
    // optional pre-reduced key block (bf16 kernel, batch-shared queries only): for every (head, query) the running
    // max pre_m, the sum pre_l = sum_j exp(s_j - pre_m) and pre_O = sum_j exp(s_j - pre_m) V_j over a set of keys
    // that is the same for the whole batch (launch_attention_prestats).  The kernel merges it with the softmax
    // over its own keys exactly as two blocks of a streaming softmax are merged.
    const float* pre_m;  // (n_head, Lq)
    const float* pre_l;  // (n_head, Lq)
    const float* pre_O;  // (n_head, Lq, hd)
    long long* stamps;   // optional (lab): 2 x 8 shader-clock stamps of workgroup 37's third item in attn_bf16_pipe_kernel
    int no_pipe;         // 1: never the pipelined kernel (attn_bf16_pipe_kernel) -- kernel-level A/B and bit-identity tests
};
// pre_m / pre_l / pre_O of queries Q (Lq rows) against keys K2/V2 (L2 rows); uses Q, ldq, K2, V2, ldkv2, L2, Lq,
// n_head, hd, scale of `p` (bf16 operands) and writes the three arrays
void launch_attention_prestats(const AttnP& p, float* pre_m, float* pre_l, float* pre_O, hipStream_t st);
void launch_attention(const AttnP& p, int dtype, hipStream_t st);
void launch_attention_bf16(const AttnP& p, hipStream_t st);

// Encoder token embedding (mtm_model.py:546-557) with the tokenizer affine folded in and the
// mask-drop gather (mtm_model.py:534-544) applied: X[b, j, :] for the kept tokens only.
struct EmbedP {
    const float* tok[4];     // per key: (.., T, D_k) inputs
    long long bstride[4];    // batch stride in floats (0 = shared by the batch)
    const int* widx;         // optional (batch,): batch element b reads window widx[b] of the inputs ...
    long long wstride[4];    // ... at tok[k] + widx[b] * wstride[k] (+ b * bstride[k])
    int normalize[4];        // apply (x - mean) / std
    const float* mean[4];
    const float* stdv[4];
    const float* WT[4];      // (D_k, d) transposed encoder_embed weight
    const float* E[4];       // (T, d)  bias + per-dim encoding + pos_embed[t]
    int feat[4];
    const int2* tokmap;      // (L,) {key, t} of every kept token, encoder order
    int batch, L, d, T;
    float* X;                // (batch, L, d)
    bf16_t* Xb;              // instead of X: the same rows rounded to bf16 (the residual stream of a pass whose layer tails take it so)
    const float* ln_g;       // optional: also emit LayerNorm(X row) (the first block's norm1) ...
    const float* ln_b;
    float* Hf;               // ... as fp32 and/or
    bf16_t* Hb;              // ... bf16 rows (batch*L, d)
    int n_indep;             // the first n_indep tokens do not depend on the batch index (computed once per wave)
    int x_first_only;        // != 0: the X rows of those tokens are stored for batch element 0 only (the consumer reads them there)
    int n_sh;                // > 0: LayerNorm rows of tokens j < n_sh go once to Hb_sh[j], those of tokens j >= n_sh
    bf16_t* Hb_sh;           //      compactly to Hb[b * (L - n_sh) + j - n_sh]  (first-layer pruning, see run_block)
};
void launch_embed(const EmbedP& p, hipStream_t st);

// out[r, :] = src_row(r), where src is an encoder-output row (per batch element) or a shared table row.
//   rowsrc[i] >= 0: row of Xe[b, rowsrc[i], :];   rowsrc[i] < 0: row (-rowsrc[i]-1) of table
struct GatherP {
    const float* Xe;
    long long xe_bstride;
    const float* table;
    const int* rowsrc;  // (rows_per_batch,)
    int rows_per_batch, batch, d;
    float* out;         // (batch, rows_per_batch, d), optional
    bf16_t* outb;       // same rows as bf16, optional
};
void launch_gather_rows(const GatherP& p, hipStream_t st);

// y[r, f] = (x[r,:] . W[f,:] + b[f]) * std[f] + mean[f]   for f < D (D <= 32): output heads' last Linear
// with the de-tokenizer folded in (tokenizers/continuous.py:81-94).
struct HeadOutP {
    const float* X;
    const bf16_t* Xb;  // instead of X: the rows in bf16 (many-row bf16 passes: head_out_mfma_kernel, round 6)
    int ldx;
    int rows, d, D;
    const float* W;    // (D, d)
    const float* b;
    const float* mean; // optional (normalize)
    const float* stdv;
    float* Y;
    RowMap ymap;       // physical output row
    int ldy;
};
void launch_head_out(const HeadOutP& p, hipStream_t st);
bool head_out_mfma_covers(int rows, int d, int D);  // Xb rows: d == 512, D <= 32, enough rows to fill the chip

// DiagGaussianActor (mtm_model.py:313-321): mu = x.Wmu + bmu ; std = exp(-5 + 3.5*(tanh(x.Wls + bls)+1))
struct ActorP {
    const float* X;
    int ldx;
    RowMap xmap;
    int rows, d, A;
    const float* Wmu;
    const float* bmu;
    const float* Wls;
    const float* bls;
    float* mu;   // (rows, A)
    float* sd;   // (rows, A)
    const float* ln_g;  // optional: LayerNorm of the row first (decoder.norm); only when actor_head_fuses_ln(d, A)
    const float* ln_b;
};
void launch_actor_head(const ActorP& p, hipStream_t st);
bool actor_head_fuses_ln(int d, int A);

// Candidate construction (learner.py:285-288 / 156-168): cand[n, t, :] = hist actions for t < idx,
// tanh(loc + std * eps) for t >= idx (mode 0) or clamp(tanh(loc) + 0.09 * eps, +-0.99999) (mode 1).
struct SampleP {
    const float* hist_actions;  // (T, A)
    const float* loc;           // (T, A)
    const float* sd;            // (T, A)
    const float* eps;           // mode 0: (n_total, T, A); mode 1: (n_total, h, A); mode 2: the candidates themselves (n_total, h, A)
    int mode, T, A, idx, h, n_begin, n_count;
    const int* widx;            // optional (n_count,): candidate n takes its history rows from window widx[n] of hist_actions (., T, A)
    const int* index;           // optional (n_count,): candidate n reads eps row index[n] instead of n_begin+n
    float* cand;                // (n_count, T, A)
    float* sample_actions;      // (n_count, h, A)
    float* loc_out;             // optional (T, A) copies of loc / sd for the caller (saves two D2D copy launches)
    float* sd_out;
};
void launch_sample(const SampleP& p, hipStream_t st);

// TwinQ (finetune_omtm/model.py:146-171): q[r] = min(q1, q2)( (s - om)/os , a )
struct CriticP {
    const float* states;  // (rows, S)
    const float* actions; // (rows, A)   row r = cand * h + t
    int rows, S, A, hidden;
    const float* om;
    const float* os;
    const float* W1T[2];  // (S+A, hidden) transposed
    const float* b1[2];
    const float* W2T[2];  // (hidden, hidden) transposed
    const float* b2[2];
    const float* W3[2];   // (hidden,)
    const float* b3[2];
    const float* W1F[2];  // the two weight matrices in MFMA operand order (critic_pack), or null: the scalar kernel runs
    const float* W2F[2];
    float* q;             // (rows,)
};
void launch_critic(const CriticP& p, hipStream_t st);
bool critic_mfma_covers(int S, int A, int hidden);
size_t critic_w1f_floats(int hidden);
size_t critic_w2f_floats(int hidden);
void critic_pack(const float* W1, const float* W2, int SA, int hidden, float* w1f, float* w2f);

// TD(lambda) scoring (learner.py:300-316)
struct ScoreP {
    const float* rewards;  // (n, h)
    const float* boot;     // (n, h)
    int n, h;
    float boot_scale;      // 1000 for rtg (learner.py:305), 1 for critic
    float gamma;           // (float)cfg.discount
    double lmbda;
    float* expect_return;  // (n,)
    float* boot_out;       // optional: scaled bootstrap written back (n,h)
    const int* scatter_index;  // optional (n,): the score of row i also goes to scatter_out[scatter_index[i]]
    float* scatter_out;
};
void launch_score(const ScoreP& p, hipStream_t st);

// softmax / weighted mean / argmax / multinomial draw over all candidates (learner.py:318-325)
struct SelectP {
    const float* er;
    const float* a0;
    long long a0_stride;
    int n, A;
    float temperature;
    const float* expo;     // optional (n,) Exp(1) variates: sample_idx = argmax(p / expo), torch.multinomial's draw
    float* p;
    float* eval_action;
    int* argmax;
    int* sample_idx;
    float* sample_action;  // (A,) = a0[sample_idx]
};
void launch_select(const SelectP& p, hipStream_t st);

// indices of the k largest values (descending; ties -> lower index first), n <= 16384, k <= n
void launch_topk(const float* v, int n, int k, int* idx_out, hipStream_t st);
// re-score window statistics: idx = the kk best entries of v (best first); stats = {n, margin to the best entry outside the n,
// max, raw count over all n_total entries}; top_scores (optional) = v[idx[i]] for i in [-rr, kk): rr race entries sit in front of idx
void launch_window_stats(const float* v, int n_total, const int* idx, int kk, int kmin, int kmax, float window, float* stats,
                         float* host_stats, float seq, float* top_scores, hipStream_t st, int rr = 0);
// list[rmax + i] = the i-th best entry of v (i < kk); list[rmax - 1 - i] = the i-th best by race key tau v - log expo (i < rr);
// list_scores (optional, same layout) = v[list[i]]: returns true when the launch wrote them itself (n <= 2048)
bool launch_topk_race(const float* v, const float* expo, float tau, int n, int kk, int rr, int rmax, int* list, float* list_scores,
                      hipStream_t st);
// list = r race entries then n score entries (the n best of b, best first), list_scores = b[list[i]], f = their fp32 re-scores:
// out = b - median(list_scores - f), listed entries replaced by f;  stats (8 floats) = {shift, max deviation,
// need = #{b > best f + shift - delta}, margin of that threshold over the best un-listed b, -, need_race (expo given: the number
// of candidates whose race key can still reach the best listed fp32 race key), that key, its threshold on the bf16 key scale}
void launch_rescore_merge(const float* b, int n_total, const int* list, int r, int n, const float* list_scores, const float* f,
                          float delta, const float* expo, float tau, float* out, float* stats, float* host_stats, float seq,
                          hipStream_t st);
// launch_rescore_merge + launch_select in ONE launch (the select runs on `out`)
void launch_merge_select(const float* b, int n_total, const int* list, int r, int n, const float* list_scores, const float* f,
                         float delta, const float* expo, float tau, float* out, float* stats, float* host_stats, float seq,
                         const SelectP& sp, hipStream_t st);
// dst[index[i]] = src[i]
void launch_scatter(const float* src, const int* index, int n, float* dst, int* index_copy, hipStream_t st);  // + index_copy[i] = index[i]

void launch_tokenize(const void* in, int in_f64, float* out, long long rows, int D, const float* mean,
                     const float* stdv, int normalize, hipStream_t st);
void launch_detokenize(const float* in, float* out, long long rows, int D, const float* mean, const float* stdv,
                       int normalize, hipStream_t st);
void launch_goal_overlay(float* pred, const float* states_in, float* states_out, long long rows, int T, int D, int idx,
                         const float* mean, const float* stdv, int normalize, hipStream_t st);
// pruned hand-over (m3pc_goal_step_batch): pred (windows * nq, D) de-tokenised rows of the window rows t <= idx, idx+2 .. T-2
void launch_goal_overlay_rows(const float* pred, const float* states_in, float* states_out, long long windows, int T, int D, int idx,
                              int nq, hipStream_t st);
void launch_f32_to_bf16(const float* in, bf16_t* out, long long n, hipStream_t st);
void launch_fill(float* out, float value, long long n, hipStream_t st);
void launch_transpose_f32(const float* in, float* out, int rows, int cols, hipStream_t st);  // out (cols, rows) = in (rows, cols)^T
void launch_embed_table(const float* bias, const float* per_dim, const float* pos, float* out, int T, int d, hipStream_t st);

}  // namespace m3pc

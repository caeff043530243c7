// Internals of libm3pc_hip.so shared by its host-side translation units (m3pc.hip: the C ABI; m3pc_plans.hip: mask plans and
// the candidate-independent decoder tables; m3pc_passes.hip: the forward passes).  Not part of the C ABI.
#pragma once
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <memory>
#include <string>
#include <vector>

#include "../../include/m3pc_hip.h"
#ifdef M3PC_LAB
#include "../../include/m3pc_hip_debug.h"
#endif
#include "kernels.h"

using namespace m3pc;

namespace m3pc {
extern thread_local char g_err[512];  // m3pc_last_error()
int fail(int code, const char* fmt, ...);
}  // namespace m3pc
#define HIPCHK(x)                                                                                     \
    do {                                                                                              \
        hipError_t e_ = (x);                                                                          \
        if (e_ != hipSuccess) return fail(M3PC_EHIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define CHK(x)               \
    do {                     \
        int rc_ = (x);       \
        if (rc_ != 0) return rc_; \
    } while (0)

static const char* KEYN[4] = {"states", "actions", "rewards", "returns"};

namespace m3pc {


struct Tensor {
    float* f = nullptr;    // fp32 device
    bf16_t* b = nullptr;   // bf16 copy (GEMM weights only)
    long long numel = 0;
    bool gemm = false;
    bool loaded = false;
};

struct SharedTables {  // candidate-independent decoder quantities of one plan, one precision
    bool valid = false;
    float* Yall = nullptr;   // (4T, d)  decoder inputs with mask tokens everywhere a token is masked
    void* QKVm = nullptr;    // (Lm, 3d) q|k|v of masked tokens (operand dtype)
    void* QKVq = nullptr;    // (nq, 3d) rows of the scored tokens (valid when they are all masked)
    float* Yq = nullptr;     // (nq, d)  decoder inputs of the scored tokens
    // softmax block of the (shared) queries against the masked tokens' keys, pre-reduced (AttnP::pre_m/l/O); bf16 only
    float *pre_m = nullptr, *pre_l = nullptr, *pre_O = nullptr;
};

constexpr int N_QUERY = 5;
struct Plan {
    std::string key;
    int T = 0, Le = 0, Lm = 0;
    int kept[4] = {0, 0, 0, 0}, enc_off[4] = {0, 0, 0, 0};
    bool prefix[4] = {true, true, true, true};
    std::vector<int> dec_src;    // (4T) encoder index or -1
    std::vector<int> masked;     // decoder indices of masked tokens
    int2* d_tokmap = nullptr;    // (Le)
    int* d_dec_rowsrc = nullptr; // (4T): enc row, or -(key)-1 -> mask token table
    int* d_masked_rowsrc = nullptr;  // (Lm): -(i)-1 rows of a (4T, *) table
    // decoder position tables of the kept tokens of key k, (kept[k], d): h->Edec[k] itself when the kept set is the prefix
    // 0..kept-1 (every finetune mask), else the kept rows gathered (the zero-shot pi mask keeps states 0..idx and T-1)
    const float* edec_kept[4] = {nullptr, nullptr, nullptr, nullptr};
    float* edec_own[4] = {nullptr, nullptr, nullptr, nullptr};
    bool edec_valid = false;
    // query sets of the pruned decoder: the decoder tokens whose outputs the caller reads, n_groups groups of grp tokens
    // per batch element, group s = tokens of key qkeys[s]
    struct Query {
        bool built = false;
        int nq = 0, h = 0;
        int n_groups = 2, grp = 0;
        int qkeys[2] = {0, 0};
        bool all_masked = true;
        // nu > 0: the un-masked query tokens are exactly the first nu queries, consecutive tokens of ONE key whose encoder rows
        // are consecutive too (critic_lambda_guiding: states[idx]; goal path inference: states[0..idx]) -- the decoder then
        // builds per-sequence rows for those nu queries only and takes the others from the shared tables
        int nu = 0, nu_key = 0, nu_enc0 = 0, nu_kept0 = 0;
        int* d_q_rowsrc_tab = nullptr;  // (nq): -(i)-1 rows of (4T,*) tables
        int* d_q_rowsrc_mix = nullptr;  // (nq): enc row or -(i)-1 rows of Yall
        SharedTables tab[2];
    } query[N_QUERY];  // index: 0 rtg (rewards, returns), 1 critic (states, rewards), 2 goal path inference (the state rows the
                       // overlay reads), 3 goal inverse dynamics (the action token at idx), 4 policy pass (the action tokens idx .. T-1)
};

struct EventPair {
    hipEvent_t a, b;
    double flops;
    int dt;
    int kind;  // 0: GEMM launch, 1: fused layer tail (block_fused_kernel), 2: fused decoder input (kv_fused_kernel)
};

}  // namespace m3pc


struct m3pc_handle {
    m3pc_dims dm;
    int device = 0;
    int d = 0, nh = 0, hd = 0, T = 0, S = 0, A = 0, ff = 0, feat[4] = {0, 0, 0, 0};
    std::map<std::string, Tensor> w;
    bool weights_loaded = false;
    long long load_stats[4] = {0, 0, 0, 0};  // last m3pc_load_weights: tensors copied, layer-tail streams packed, kv streams packed, tables invalidated
    // derived tables
    float* WT[4] = {nullptr, nullptr, nullptr, nullptr};
    float* Eenc[4] = {nullptr, nullptr, nullptr, nullptr};
    float* Edec[4] = {nullptr, nullptr, nullptr, nullptr};
    float* mask_tokens = nullptr;  // (4, d)
    // tokenizer
    bool tok_set[4] = {false, false, false, false};
    int tok_norm[4] = {0, 0, 0, 0};
    float* tok_mean[4] = {nullptr, nullptr, nullptr, nullptr};
    float* tok_std[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<float> h_mean[4], h_std[4];
    // critic
    bool critic_set = false;
    float *cW1T[2] = {nullptr, nullptr}, *cb1[2] = {nullptr, nullptr}, *cW2T[2] = {nullptr, nullptr},
          *cb2[2] = {nullptr, nullptr}, *cW3[2] = {nullptr, nullptr}, *cb3[2] = {nullptr, nullptr};
    float *cW1F[2] = {nullptr, nullptr}, *cW2F[2] = {nullptr, nullptr};  // MFMA operand order (critic_pack)
    float *c_om = nullptr, *c_os = nullptr;
    // workspace
    long long R = 0;
    float *X = nullptr, *Y = nullptr, *EncOut = nullptr, *G = nullptr;
    void *Hn = nullptr, *QKV = nullptr, *O = nullptr, *F = nullptr, *Z = nullptr;
    float *cand = nullptr, *loc = nullptr, *sd = nullptr, *rtok = nullptr, *pred[2] = {nullptr, nullptr}, *qv = nullptr;
    float* sel_scratch = nullptr;
    float* goal_ws = nullptr;     // (max_goal_batch, T, S) the window rows the second forward of m3pc_goal_step_batch sees
    int* d_topk = nullptr;        // (1024,) candidate ids of the last top-k
    float* er_top = nullptr;      // (1024,) their fp32 re-scores
    float* sa_buf = nullptr;      // (max(max_candidates, max_rescore), h, A) scratch for m3pc_rescore
    float* sa_chain[2] = {nullptr, nullptr};  // the same for re-scores that run in the chain workspaces (one per slot parity)
    float* splitk_ws = nullptr;   // raw split-K slabs of the few-row fp32 GEMMs
    long long splitk_ws_bytes = 0;
    // Step slots: the per-step state a plan step leaves behind its policy pass (loc / sd of the policy head, the normalised
    // returns tokens).  A pipelined caller (m3pc_policy_pass of step t+1 on one stream beside m3pc_candidate_pass of step t on
    // another, the fp32 re-score of step t after it) gives every step in flight its own slot (m3pc_plan_args::slot);
    // loc / sd / rtok above are VIEWS of the slot bound last (bind_slot).
    struct Slot {
        float *loc = nullptr, *sd = nullptr, *rtok = nullptr;
        bool policy_valid = false;  // loc / sd / rtok hold a single-window policy pass (what m3pc_rescore needs)
        int n_windows = 0;          // policy heads the slot holds (m3pc_policy_pass: 1, m3pc_policy_pass_batch: E)
    } slot[M3PC_SLOTS];
    int cur_slot = 0;
    // Workspaces.  The pointers above (X ... splitk_ws) are VIEWS of the workspace bound last (bind_ws), re-based per candidate
    // half by set_view().  `base` is the candidate workspace (max_candidates); the few-row fp32 chains run in two small ones of
    // their own -- `pchain` the policy pass (batch <= max_batch), `chain` the re-score (<= max_rescore candidates) -- so that a
    // policy pass, a re-score and a candidate pass of three different steps can be enqueued on three streams at the same time
    // without sharing a buffer.  There are TWO of each, picked by the parity of the step slot (m3pc_plan_args::slot & 1): the
    // chains of consecutive steps may then run on two streams at the same time (m3pc_amd/planner.py: the policy pass and the
    // re-score of a step on the stream of its parity) -- in the pipelined step the re-score chain of one stream was the bottleneck.
    struct Base {
        float *X = nullptr, *Y = nullptr, *EncOut = nullptr, *G = nullptr, *cand = nullptr, *pred[2] = {nullptr, nullptr},
              *qv = nullptr, *splitk_ws = nullptr;
        char *Hn = nullptr, *QKV = nullptr, *O = nullptr, *F = nullptr, *Z = nullptr;
        long long splitk_ws_bytes = 0;
        float* head_part = nullptr;   // scratch of the fused fp32 scalar heads (launch_head_f32_fused): 2 * head_rows * 64 floats
        int* head_ticket = nullptr;   // its row-tile tickets (zero between launches)
        int head_rows = 0;            // rows per head the scratch holds
        long long R = 0;       // token rows
        int max_cand = 0;      // candidates (rows of cand / pred / qv)
    } base, chain[2], pchain[2];  // chain / pchain: one per step-slot parity (see below)
    Base* cur = nullptr;
    bool two_stream = true;       // candidate halves on two streams (M3PC_TWO_STREAM=0: one stream); measured -2.5 % step time on C2
    bool allow_splitk = true;     // see gemm(): off while sharded candidates are scored
    double pass_scale = 1.0;      // candidates of the whole plan step / candidates of the launch being enqueued (>= 1; FUSED_MIN_ROWS)
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    const int* score_scatter_index = nullptr;  // set around a pass whose scores also go to scatter_out[index[i]] (score_kernel)
    float* score_scatter_out = nullptr;
    std::vector<hipStream_t> auxs;   // auxs[0] == aux
    std::vector<hipEvent_t> ev_joins;
    std::vector<int> stream_split;
    // Deferred joins (M3PC_PLAN_DEFER_JOIN): the parts of a candidate pass that ran on the handle's own streams are joined by
    // the consumer of the step (m3pc_candidate_join) instead of the caller's stream.  slot_join[s][i]: recorded behind part
    // i + 1 of the last deferred pass of slot s; aux_unjoined[i]: stream auxs[i] holds candidate-workspace work nobody
    // waited for on behalf of the workspace (ws_sync); defer_parts: the part sizes of the passes in flight -- a pass with the
    // same sizes touches, per stream, the very rows that stream's earlier work touched, and needs no cross-stream order.
    hipEvent_t slot_join[M3PC_SLOTS][3] = {};
    int slot_join_n[M3PC_SLOTS] = {};
    hipEvent_t aux_tail[3] = {};
    bool aux_unjoined[3] = {false, false, false};
    std::vector<int> defer_parts;
    std::map<std::string, std::unique_ptr<Plan>> plans;
    Plan* mask_plan[4][65] = {};  // get_mask_plan cache: [rcbc | fd | pi = gid | fid][idx]
    // packed MFMA-fragment weight streams of the fused layer tails (block_fused.hip), by block prefix
    std::map<std::string, bf16_t*> wstream;
    // packed streams of the fused decoder input (kv_fused_kernel), by key: embedding of key k + K|V rows of decoder layer 0
    bf16_t* kvstream[4] = {nullptr, nullptr, nullptr, nullptr};
    // lab: in-kernel phase stamps of one workgroup of every fused-tail launch, as the step runs (m3pc_debug_stamp_log)
    long long* stamp_log = nullptr;
    int stamp_cap = 0, stamp_i = 0;
    // profiling
    bool prof = false;
    bool prof_serial = false;     // m3pc_profile_enable(h, 2): the candidate halves run one after the other on the caller's stream
    std::vector<EventPair> ev;
    size_t ev_used = 0;
};

namespace m3pc {


template <typename T>
inline int dmalloc(T** p, size_t n) {
    hipError_t e = hipMalloc((void**)p, n * sizeof(T) > 0 ? n * sizeof(T) : 16);
    if (e != hipSuccess) return fail(M3PC_EHIP, "hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e));
    return 0;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(M3PC_EHIP, "kernel launch failed in %s: %s", what, hipGetErrorString(e));
    return 0;
}

// Re-base the workspace views at candidate c0 (each candidate owns 2T workspace rows).
inline void set_view(m3pc_handle* h, int c0, int /*n*/) {
    const size_t rows = (size_t)c0 * 2 * h->T, d = (size_t)h->d;
    const m3pc_handle::Base& b = *h->cur;
    h->X = b.X + rows * d;
    h->Y = b.Y + rows * d;
    h->EncOut = b.EncOut + rows * d;
    h->G = b.G + rows * d;
    h->Hn = b.Hn + rows * d * 4;
    h->QKV = b.QKV + rows * 3 * d * 4;
    h->O = b.O + rows * d * 4;
    h->F = b.F + rows * 4 * d * 4;
    h->Z = b.Z + rows * d * 4;
    h->cand = b.cand + (size_t)c0 * h->T * h->A;
    h->pred[0] = b.pred[0] + (size_t)c0 * h->T * 32;
    h->pred[1] = b.pred[1] + (size_t)c0 * h->T * 32;
    h->qv = b.qv + (size_t)c0 * h->T;
    const long long half = b.splitk_ws_bytes / 2;
    h->splitk_ws = c0 == 0 ? b.splitk_ws : b.splitk_ws + half / 4;
    h->splitk_ws_bytes = half;
}

// Bind a workspace (the candidate one or the chain one): every launcher below reads the views.  Host-side state only: what
// was enqueued before keeps the pointers it was enqueued with.
inline void bind_ws(m3pc_handle* h, m3pc_handle::Base* b) {
    h->cur = b;
    h->R = b->R;
    set_view(h, 0, b->max_cand);
}
struct WsScope {  // binds a chain workspace for the duration of a few-row fp32 pass
    m3pc_handle* h;
    WsScope(m3pc_handle* h_, bool chain, bool policy = false, int slot = 0) : h(h_) {
        if (chain) bind_ws(h, policy ? &h->pchain[slot & 1] : &h->chain[slot & 1]);
    }
    ~WsScope() { bind_ws(h, &h->base); }
};
inline void bind_slot(m3pc_handle* h, int s) {
    h->cur_slot = s;
    h->loc = h->slot[s].loc;
    h->sd = h->slot[s].sd;
    h->rtok = h->slot[s].rtok;
}

// Orders `st` behind the deferred parts of earlier candidate passes: every call that touches the candidate workspace
// in any other shape than those passes starts with it.
inline int ws_sync(m3pc_handle* h, hipStream_t st) {
    for (int i = 0; i < 3; ++i)
        if (h->aux_unjoined[i]) {
            HIPCHK(hipStreamWaitEvent(st, h->aux_tail[i], 0));
            h->aux_unjoined[i] = false;
        }
    h->defer_parts.clear();
    return 0;
}

inline int alloc_ws(m3pc_handle* h, m3pc_handle::Base& b, long long R_, int max_cand, long long splitk_bytes) {
    const size_t R = (size_t)R_, d = (size_t)h->d, T = (size_t)h->T;
    b.R = R_;
    b.max_cand = max_cand;
    CHK(dmalloc(&b.X, R * d));
    CHK(dmalloc(&b.Y, R * d));
    CHK(dmalloc(&b.EncOut, R * d));
    CHK(dmalloc(&b.G, R * d));
    CHK(dmalloc((float**)&b.Hn, R * d));
    CHK(dmalloc((float**)&b.QKV, R * 3 * d));
    CHK(dmalloc((float**)&b.O, R * d));
    CHK(dmalloc((float**)&b.F, R * 4 * d));
    CHK(dmalloc((float**)&b.Z, R * d));
    CHK(dmalloc(&b.cand, (size_t)max_cand * T * h->A));
    CHK(dmalloc(&b.pred[0], (size_t)max_cand * T * 32));
    CHK(dmalloc(&b.pred[1], (size_t)max_cand * T * 32));
    CHK(dmalloc(&b.qv, (size_t)max_cand * T));
    b.splitk_ws_bytes = splitk_bytes;
    CHK(dmalloc(&b.splitk_ws, (size_t)(splitk_bytes / 4)));
    if (max_cand <= 4096) {  // (the few-row workspaces: the fused fp32 heads of a re-score; at most T scored rows per candidate and head)
        b.head_rows = (int)((long long)max_cand * h->T < 4096 ? (long long)max_cand * h->T : 4096);
        CHK(dmalloc(&b.head_part, (size_t)2 * b.head_rows * 64));
        CHK(dmalloc(&b.head_ticket, (size_t)2 * ((b.head_rows + 31) / 32)));
        HIPCHK(hipMemset(b.head_ticket, 0, (size_t)2 * ((b.head_rows + 31) / 32) * sizeof(int)));
    }
    return 0;
}
inline void free_ws(m3pc_handle::Base& b) {
    void* bufs[] = {b.X, b.Y, b.EncOut, b.G, b.Hn, b.QKV, b.O, b.F, b.Z, b.cand, b.pred[0], b.pred[1], b.qv, b.splitk_ws, b.head_part, b.head_ticket};
    for (void* p : bufs)
        if (p) hipFree(p);
    b = m3pc_handle::Base();
}

inline Tensor& W(m3pc_handle* h, const std::string& n) { return h->w.at(n); }
inline const void* Wop(m3pc_handle* h, const std::string& n, int dt) {
    Tensor& t = h->w.at(n);
    return dt == DT_BF16 ? (const void*)t.b : (const void*)t.f;
}

inline void add_tensor(m3pc_handle* h, const std::string& n, long long numel, bool gemm = false) {
    Tensor t;
    t.numel = numel;
    t.gemm = gemm;
    h->w[n] = t;
}

inline void declare_weights(m3pc_handle* h) {
    const int d = h->d, ff = h->ff;
    for (int k = 0; k < 4; ++k) {
        const std::string kn = KEYN[k];
        add_tensor(h, "encoder_embed_dict." + kn + ".weight", (long long)d * h->feat[k]);
        add_tensor(h, "encoder_embed_dict." + kn + ".bias", d);
        add_tensor(h, "decoder_embed_dict." + kn + ".weight", (long long)d * d, true);
        add_tensor(h, "decoder_embed_dict." + kn + ".bias", d);
        add_tensor(h, "mask_token_dict." + kn, d);
        add_tensor(h, "encoder_per_dim_encoding." + kn, d);
        add_tensor(h, "decoder_per_dim_encoding." + kn, d);
        if (k == M3PC_ACTIONS) {
            add_tensor(h, "output_head_dict.actions.mu.weight", (long long)h->A * d);
            add_tensor(h, "output_head_dict.actions.mu.bias", h->A);
            add_tensor(h, "output_head_dict.actions.log_std.weight", (long long)h->A * d);
            add_tensor(h, "output_head_dict.actions.log_std.bias", h->A);
        } else {
            add_tensor(h, "output_head_dict." + kn + ".0.weight", d);
            add_tensor(h, "output_head_dict." + kn + ".0.bias", d);
            add_tensor(h, "output_head_dict." + kn + ".1.weight", (long long)d * d, true);
            add_tensor(h, "output_head_dict." + kn + ".1.bias", d);
            add_tensor(h, "output_head_dict." + kn + ".3.weight", (long long)h->feat[k] * d);
            add_tensor(h, "output_head_dict." + kn + ".3.bias", h->feat[k]);
        }
    }
    auto block = [&](const std::string& p) {
        add_tensor(h, p + ".self_attn.in_proj_weight", 3LL * d * d, true);
        add_tensor(h, p + ".self_attn.in_proj_bias", 3 * d);
        add_tensor(h, p + ".self_attn.out_proj.weight", (long long)d * d, true);
        add_tensor(h, p + ".self_attn.out_proj.bias", d);
        add_tensor(h, p + ".linear1.weight", (long long)ff * d, true);
        add_tensor(h, p + ".linear1.bias", ff);
        add_tensor(h, p + ".linear2.weight", (long long)d * ff, true);
        add_tensor(h, p + ".linear2.bias", d);
        add_tensor(h, p + ".norm1.weight", d);
        add_tensor(h, p + ".norm1.bias", d);
        add_tensor(h, p + ".norm2.weight", d);
        add_tensor(h, p + ".norm2.bias", d);
    };
    for (int i = 0; i < h->dm.n_enc_layer; ++i) block("encoder.layers." + std::to_string(i));
    for (int i = 0; i < h->dm.n_dec_layer; ++i) block("decoder.layers." + std::to_string(i));
    add_tensor(h, "encoder.norm.weight", d);
    add_tensor(h, "encoder.norm.bias", d);
    add_tensor(h, "decoder.norm.weight", d);
    add_tensor(h, "decoder.norm.bias", d);
    add_tensor(h, "pos_embed", (long long)h->T * d);
}

// ---------------------------------------------------------------------------------- profiling
struct GemmTimer {
    m3pc_handle* h;
    hipStream_t st;
    EventPair* e = nullptr;
    GemmTimer(m3pc_handle* h_, hipStream_t st_, double flops, int dt, int kind = 0) : h(h_), st(st_) {
        if (!h->prof) return;
        if (h->ev_used == h->ev.size()) {
            EventPair n;
            hipEventCreate(&n.a);
            hipEventCreate(&n.b);
            h->ev.push_back(n);
        }
        e = &h->ev[h->ev_used++];
        e->flops = flops;
        e->dt = dt;
        e->kind = kind;
        hipEventRecord(e->a, st);
    }
    ~GemmTimer() {
        if (e) hipEventRecord(e->b, st);
    }
};

// returns 1 when the LayerNorm named by p_in.ln_* was fused into the launch (GemmP::ln_g)
inline int gemm(m3pc_handle* h, const GemmP& p_in, int dt, hipStream_t st) {
    GemmP p = p_in;
    // split-K changes the association of the K sum, so it is only allowed where every rank / shard runs the
    // same row count (policy pass, generic forward, top-k re-score): sharded candidate scores stay bit-identical
    p.ws = h->allow_splitk ? h->splitk_ws : nullptr;
    p.ws_bytes = h->splitk_ws_bytes;
#ifdef M3PC_LAB  // (the lab build only: an environment variable must not change which kernels the product runs)
    static const int env_variant = M3PC_ENV("M3PC_GEMM_VARIANT") ? atoi(M3PC_ENV("M3PC_GEMM_VARIANT")) : 0;  // A/B runs
    if (env_variant) p.variant = env_variant;
#endif
    GemmTimer t(h, st, 2.0 * p.M * (double)p.N * p.K, dt);
    return launch_gemm(p, dt, st);
}

// Few-row fp32 passes: can the LayerNorm in front of GEMM `p` ride on its operand load (gemm_f32_direct.hip: a_ln_*)?
// `p` must already read the un-normalised rows (A = X, lda = d) and carry a_ln_g / a_ln_b.
// Measured on the policy pass: the folded GEMM takes 11-12.5 us where GEMM 6.8 + LayerNorm launch 4.3-5
// took 11-12 -- every workgroup recomputes its rows' statistics behind a barrier before its K loop starts, which costs what
// the launch cost.  Neutral, so OFF by default (M3PC_LN_FOLD=1 turns it on for A/B runs); the GPU tests pass either way.
inline bool can_fold_ln(m3pc_handle* h, const GemmP& p, int dt) {
    static const bool on = M3PC_ENV("M3PC_LN_FOLD") != nullptr && M3PC_ENV("M3PC_NO_F32_DIRECT") == nullptr &&
                           M3PC_ENV("M3PC_GEMM_VARIANT") == nullptr;  // A/B switch
    return dt == DT_F32 && on && h->allow_splitk && h->splitk_ws && gemm_f32_direct_covers(p);
}

inline GemmP gemm_basic(const void* A, int lda, const void* Wp, int ldw, int M, int N, int K, const float* bias) {
    GemmP p;
    memset(&p, 0, sizeof(p));
    p.A = A;
    p.lda = lda;
    p.W = Wp;
    p.ldw = ldw;
    p.M = M;
    p.N = N;
    p.K = K;
    p.bias = bias;
    p.rt_mod = 1;
    // every A operand built here is one of the handle's workspaces (fp32-sized, R rows): a many-row bf16 operand leaves at
    // least as many bytes behind its last row as it occupies (see GemmP::a_padded)
    p.a_padded = 1;
    return p;
}
inline void gemm_out(GemmP& p, int dt_out, void* C, int ldc) {
    if (dt_out == DT_BF16)
        p.Cb = (bf16_t*)C;
    else
        p.Cf = (float*)C;
    p.ldc = ldc;
}

// ---------------------------------------------------------------------------------- transformer block
// One pre-LN layer (mtm_model.py:379-409) over `batch` sequences of L rows, in place on X (fp32).
// next_ln: the LayerNorm that follows this block on X (next block's norm1 or the stack's final norm); when the
// FFN2 GEMM can apply it in its split-K reduce, *next_ln_done is set and the caller skips that launch.
// x_dead: nothing reads X after this block except through next_ln (lets the fused tail skip the fp32 store).
// Xnext / res_nshared (fused tail only): the block output goes to Xnext instead of X, and the first res_nshared rows of
// every sequence of X are read from sequence 0 (the embedding kernel stored the history rows once, EmbedP::x_first_only).
// The fused layer tail (block_fused.hip) works in 128-row tiles, one per CU, and a tile takes its ~130-170 us whatever the row
// count: below ~96 tiles most of the chip idles for that long and the GEMM chain, whose tiles spread over all CUs, is faster
// (the reference's shipped N=625 / T=8 config, 64 + 40 tiles: 1.13 -> 1.00 ms per closed-loop call).
// The choice goes by the size of the WHOLE step (m3pc_handle::pass_scale = n_total / candidates of this launch), not by the
// rows of a shard or a candidate part: a candidate's score must not depend on how the candidates were cut.
constexpr long long FUSED_MIN_ROWS = 96 * 128;
// Between 16 and 96 tiles the tail still runs fused, four workgroups per tile (each a quarter of the FFN's hidden units, fp32
// partials to four slabs in the F buffer) with a row-wise reduce + LayerNorm launch behind it (block_fused_kernel<0, 3>).
constexpr long long SPLIT_MIN_ROWS = 16 * 128;

struct TokIn {
    const float* ptr[4];
    long long bstride[4];
    int normalize[4];
    const int* widx = nullptr;  // optional per-batch-element window index into ptr[k] (stride wstride[k])
    long long wstride[4] = {0, 0, 0, 0};
};

// embed + encoder stack + encoder.norm -> EncOut (fp32) [and bf16 copy in Z when dt == bf16 and want_b]
// n_indep: number of leading encoder tokens that are identical for every batch element (candidate pass: history)
// layer_from / layer_to / ln_state: the pass can be enqueued in pieces (the embedding goes with layer 0, encoder.norm with the
// last layer); *ln_state carries "norm1 of the next layer is already in Hn" (ln) / "its Q|K|V rows are already in QKV" (qkv)
// from one piece to the next
struct PieceState {
    bool ln = true, qkv = false;
};

enum { TAIL_HEADS = 0, TAIL_X = 1 };

// m3pc_plans.hip
int get_plan(m3pc_handle* h, const unsigned char* const masks[4], Plan** out);
int get_mask_plan(m3pc_handle* h, int kind, int idx, Plan** out);
int ensure_edec(m3pc_handle* h, Plan* pl, hipStream_t st);
void free_tables(SharedTables& t);
void invalidate_tables(m3pc_handle* h);
int build_query_list(m3pc_handle* h, Plan* pl, int qi, int hh, const std::vector<int>& toks, int n_groups, int key0, int key1);
int build_query(m3pc_handle* h, Plan* pl, int qi, int hh);
int build_tables(m3pc_handle* h, Plan* pl, int qi, int dt, hipStream_t st);
// m3pc_passes.hip
int run_block(m3pc_handle* h, const std::string& pfx, float* X, int batch, int L, int dt, hipStream_t st, bool ln1_done = false,
              int n_sh = 0, const LnP* next_ln = nullptr, bool* next_ln_done = nullptr, bool x_dead = false,
              float* Xnext = nullptr, int res_nshared = 0, bool qkv_done = false, const std::string* next_qkv = nullptr,
              bool* next_qkv_done = nullptr, bool x_bf16 = false);
int run_encoder(m3pc_handle* h, Plan* pl, const TokIn& in, int batch, int dt, hipStream_t st, bool bf16_out_only = false,
                int n_indep = 0, int layer_from = 0, int layer_to = 1 << 30, PieceState* ln_state = nullptr);
void dec_embed(m3pc_handle* h, int k, const void* Zop, RowMap amap, float* Yout, RowMap cmap, int M, int mod, int dt,
               hipStream_t st, const float* table = nullptr);
int run_decoder_full(m3pc_handle* h, const void* Zop, int batch, int dt, hipStream_t st);
int run_head(m3pc_handle* h, int k, const float* Ysrc, RowMap xmap, int rows, float* out, int ldy, bool detok, int dt,
             hipStream_t st);
int run_head_tail(m3pc_handle* h, int k, const void* ln_rows, int rows, float* out, int ldy, bool detok, int dt, hipStream_t st);
int forward_impl(m3pc_handle* h, Plan* pl, const TokIn& in, int batch, float* out_states, float* out_rewards,
                 float* out_returns, float* out_mu, float* out_std, int dt, hipStream_t st);
int pruned_decoder(m3pc_handle* h, Plan* pl, Plan::Query& q, SharedTables& tb, int n, int dt, hipStream_t st, int tail,
                   float** xrows);
int candidate_pass(m3pc_handle* h, const m3pc_plan_args* a, const float* states, const float* rewards, int n,
                   const float* sample_actions, float* expect_return, float* pred_rewards, float* pred_boot, int dt,
                   hipStream_t st, const int* widx = nullptr, int stage_from = 0, int stage_to = 1 << 30, PieceState* ln_state = nullptr);

}  // namespace m3pc

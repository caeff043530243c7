// libm3pc_hip.so -- host side of the C ABI declared in include/m3pc_hip.h.
//
// Data layout in HBM (one handle = one GPU):
//   weights   fp32 arena in state_dict layout + bf16 copies of every GEMM weight ([N][K], K contiguous)
//   tables    per key: transposed encoder-embed weight (D_k,d); E_enc/E_dec (T,d) = bias + per-dim + pos
//   plans     per mask pattern: token maps, and -- for the candidate pass -- the candidate-independent
//             part of the decoder (inputs, K/V and Q of every masked token), computed once per weight load
//   workspace activations for R = max(max_candidates*2T, max_batch*4T) token rows:
//             X, Y (fp32 residual streams), Hn, QKV, O, F (operand dtype), EncOut (fp32)
//
// Candidate pass ("pass 2", learner.py:288-293) is exactly pruned: decoder rows of masked tokens do not
// depend on the candidate, so only the 2h scored tokens are pushed through out-proj/FFN/heads and only the
// un-masked tokens through the K/V projection.
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

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
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

namespace {

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

constexpr int N_QUERY = 4;
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
                       // overlay reads), 3 goal inverse dynamics (the action token at idx)
};

struct EventPair {
    hipEvent_t a, b;
    double flops;
    int dt;
    int kind;  // 0: GEMM launch, 1: fused layer tail (block_fused_kernel), 2: fused decoder input (kv_fused_kernel)
};

}  // namespace

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

namespace {

template <typename T>
int dmalloc(T** p, size_t n) {
    hipError_t e = hipMalloc((void**)p, n * sizeof(T) > 0 ? n * sizeof(T) : 16);
    if (e != hipSuccess) return fail(M3PC_EHIP, "hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e));
    return 0;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(M3PC_EHIP, "kernel launch failed in %s: %s", what, hipGetErrorString(e));
    return 0;
}

// Re-base the workspace views at candidate c0 (each candidate owns 2T workspace rows).
void set_view(m3pc_handle* h, int c0, int /*n*/) {
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
void bind_ws(m3pc_handle* h, m3pc_handle::Base* b) {
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
void bind_slot(m3pc_handle* h, int s) {
    h->cur_slot = s;
    h->loc = h->slot[s].loc;
    h->sd = h->slot[s].sd;
    h->rtok = h->slot[s].rtok;
}

// Orders `st` behind the deferred parts of earlier candidate passes: every call that touches the candidate workspace
// in any other shape than those passes starts with it.
int ws_sync(m3pc_handle* h, hipStream_t st) {
    for (int i = 0; i < 3; ++i)
        if (h->aux_unjoined[i]) {
            HIPCHK(hipStreamWaitEvent(st, h->aux_tail[i], 0));
            h->aux_unjoined[i] = false;
        }
    h->defer_parts.clear();
    return 0;
}

int alloc_ws(m3pc_handle* h, m3pc_handle::Base& b, long long R_, int max_cand, long long splitk_bytes) {
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
    return 0;
}
void free_ws(m3pc_handle::Base& b) {
    void* bufs[] = {b.X, b.Y, b.EncOut, b.G, b.Hn, b.QKV, b.O, b.F, b.Z, b.cand, b.pred[0], b.pred[1], b.qv, b.splitk_ws};
    for (void* p : bufs)
        if (p) hipFree(p);
    b = m3pc_handle::Base();
}

Tensor& W(m3pc_handle* h, const std::string& n) { return h->w.at(n); }
const void* Wop(m3pc_handle* h, const std::string& n, int dt) {
    Tensor& t = h->w.at(n);
    return dt == DT_BF16 ? (const void*)t.b : (const void*)t.f;
}

void add_tensor(m3pc_handle* h, const std::string& n, long long numel, bool gemm = false) {
    Tensor t;
    t.numel = numel;
    t.gemm = gemm;
    h->w[n] = t;
}

void declare_weights(m3pc_handle* h) {
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
int gemm(m3pc_handle* h, const GemmP& p_in, int dt, hipStream_t st) {
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
bool can_fold_ln(m3pc_handle* h, const GemmP& p, int dt) {
    static const bool on = M3PC_ENV("M3PC_LN_FOLD") != nullptr && M3PC_ENV("M3PC_NO_F32_DIRECT") == nullptr &&
                           M3PC_ENV("M3PC_GEMM_VARIANT") == nullptr;  // A/B switch
    return dt == DT_F32 && on && h->allow_splitk && h->splitk_ws && gemm_f32_direct_covers(p);
}

GemmP gemm_basic(const void* A, int lda, const void* Wp, int ldw, int M, int N, int K, const float* bias) {
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
void gemm_out(GemmP& p, int dt_out, void* C, int ldc) {
    if (dt_out == DT_BF16)
        p.Cb = (bf16_t*)C;
    else
        p.Cf = (float*)C;
    p.ldc = ldc;
}

// ---------------------------------------------------------------------------------- plans
int get_plan(m3pc_handle* h, const unsigned char* const masks[4], Plan** out) {
    const int T = h->T;
    std::string key(4 * T, '0');
    for (int k = 0; k < 4; ++k)
        for (int t = 0; t < T; ++t) key[k * T + t] = masks[k][t] ? '1' : '0';
    auto it = h->plans.find(key);
    if (it != h->plans.end()) {
        *out = it->second.get();
        return 0;
    }
    std::unique_ptr<Plan> pl(new Plan());
    pl->key = key;
    pl->T = T;
    std::vector<int2> tokmap;
    pl->dec_src.assign(4 * T, -1);
    for (int k = 0; k < 4; ++k) {
        pl->enc_off[k] = (int)tokmap.size();
        bool seen_zero = false;
        for (int t = 0; t < T; ++t) {
            if (masks[k][t]) {
                if (seen_zero) pl->prefix[k] = false;
                pl->dec_src[k * T + t] = (int)tokmap.size();
                tokmap.push_back(make_int2(k, t));
                pl->kept[k]++;
            } else {
                seen_zero = true;
            }
        }
    }
    pl->Le = (int)tokmap.size();
    if (pl->Le == 0) return fail(M3PC_EINVAL, "mask keeps no token");
    std::vector<int> dec_rowsrc(4 * T), masked_rowsrc;
    for (int i = 0; i < 4 * T; ++i) {
        if (pl->dec_src[i] >= 0) {
            dec_rowsrc[i] = pl->dec_src[i];
        } else {
            dec_rowsrc[i] = -(i / T) - 1;
            pl->masked.push_back(i);
            masked_rowsrc.push_back(-i - 1);
        }
    }
    pl->Lm = (int)pl->masked.size();
    CHK(dmalloc(&pl->d_tokmap, tokmap.size()));
    HIPCHK(hipMemcpy(pl->d_tokmap, tokmap.data(), tokmap.size() * sizeof(int2), hipMemcpyHostToDevice));
    CHK(dmalloc(&pl->d_dec_rowsrc, dec_rowsrc.size()));
    HIPCHK(hipMemcpy(pl->d_dec_rowsrc, dec_rowsrc.data(), dec_rowsrc.size() * sizeof(int), hipMemcpyHostToDevice));
    CHK(dmalloc(&pl->d_masked_rowsrc, masked_rowsrc.size() + 1));
    if (!masked_rowsrc.empty())
        HIPCHK(hipMemcpy(pl->d_masked_rowsrc, masked_rowsrc.data(), masked_rowsrc.size() * sizeof(int), hipMemcpyHostToDevice));
    *out = pl.get();
    h->plans[key] = std::move(pl);
    return 0;
}

// The deterministic test-time masks, cached per (kind, idx): kind 0 = rcbc (finetune_omtm/masks.py:7-27: states[:idx+1],
// actions[:idx], all returns), kind 1 = fd (masks.py:30-44: states[:idx+1], all actions), kind 2 = pi = gid
// (zeroshot_omtm/masks.py:72-91 / 50-69: all states but idx+1 .. T-2 when idx > 0, actions[:idx]), kind 3 = fid
// (zeroshot_omtm/masks.py:30-47: all states, actions[:idx]).  No host-side mask work after the first call with a given idx.
int get_mask_plan(m3pc_handle* h, int kind, int idx, Plan** out) {
    Plan*& slot = h->mask_plan[kind][idx];
    if (!slot) {
        const int T = h->T;
        std::vector<unsigned char> m[4];
        for (int k = 0; k < 4; ++k) m[k].assign(T, 0);
        if (kind <= 1) {
            for (int t = 0; t <= idx && t < T; ++t) m[M3PC_STATES][t] = 1;
            for (int t = 0; t < (kind == 0 ? idx : T); ++t) m[M3PC_ACTIONS][t] = 1;
            if (kind == 0)
                for (int t = 0; t < T; ++t) m[M3PC_RETURNS][t] = 1;
        } else {
            for (int t = 0; t < T; ++t) m[M3PC_STATES][t] = 1;
            if (kind == 2 && idx > 0)
                for (int t = idx + 1; t < T - 1; ++t) m[M3PC_STATES][t] = 0;  // state_mask[idx + 1 : -1] = 0
            for (int t = 0; t < idx; ++t) m[M3PC_ACTIONS][t] = 1;
        }
        const unsigned char* mp[4] = {m[0].data(), m[1].data(), m[2].data(), m[3].data()};
        CHK(get_plan(h, mp, &slot));
    }
    *out = slot;
    return 0;
}

// Plan::edec_kept: the decoder position table rows of the kept tokens of each key, in encoder order
int ensure_edec(m3pc_handle* h, Plan* pl, hipStream_t st) {
    if (pl->edec_valid) return 0;
    const int T = h->T, d = h->d;
    for (int k = 0; k < 4; ++k) {
        if (pl->prefix[k] || pl->kept[k] == 0) {
            pl->edec_kept[k] = h->Edec[k];
            continue;
        }
        if (!pl->edec_own[k]) CHK(dmalloc(&pl->edec_own[k], (size_t)T * d));
        int j = 0;
        for (int t = 0; t < T; ++t)
            if (pl->dec_src[k * T + t] >= 0) {
                HIPCHK(hipMemcpyAsync(pl->edec_own[k] + (size_t)j * d, h->Edec[k] + (size_t)t * d, (size_t)d * sizeof(float),
                                      hipMemcpyDeviceToDevice, st));
                ++j;
            }
        pl->edec_kept[k] = pl->edec_own[k];
    }
    pl->edec_valid = true;
    return 0;
}

void free_tables(SharedTables& t) {
    if (t.Yall) hipFree(t.Yall);
    if (t.QKVm) hipFree(t.QKVm);
    if (t.QKVq) hipFree(t.QKVq);
    if (t.Yq) hipFree(t.Yq);
    if (t.pre_m) hipFree(t.pre_m);
    if (t.pre_l) hipFree(t.pre_l);
    if (t.pre_O) hipFree(t.pre_O);
    t = SharedTables();
}

void invalidate_tables(m3pc_handle* h) {
    for (auto& kv : h->plans)
    {
        for (int q = 0; q < N_QUERY; ++q)
            for (int pr = 0; pr < 2; ++pr) kv.second->query[q].tab[pr].valid = false;
        kv.second->edec_valid = false;
    }
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

// qkv_done: the previous layer's fused tail already wrote this layer's Q|K|V rows; next_qkv: prefix of the layer whose Q|K|V
// projection this layer's fused tail may compute (-> *next_qkv_done)
int run_block(m3pc_handle* h, const std::string& pfx, float* X, int batch, int L, int dt, hipStream_t st, bool ln1_done = false,
              int n_sh = 0, const LnP* next_ln = nullptr, bool* next_ln_done = nullptr, bool x_dead = false,
              float* Xnext = nullptr, int res_nshared = 0, bool qkv_done = false, const std::string* next_qkv = nullptr,
              bool* next_qkv_done = nullptr) {
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
    if (dt == DT_BF16 && !no_fused && !no_split && step_rows < (double)FUSED_MIN_ROWS && step_rows >= (double)SPLIT_MIN_ROWS &&
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
        if (launch_block_fused(b, st)) {
            if (next_ln_done) *next_ln_done = fuse_ln;
            return check_launch(pfx.c_str());
        }
        if (next_qkv_done) *next_qkv_done = false;
    }
    if (Xnext || res_nshared) return fail(M3PC_EINVAL, "%s: the fused layer tail did not take a pass set up for it", pfx.c_str());
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
int run_encoder(m3pc_handle* h, Plan* pl, const TokIn& in, int batch, int dt, hipStream_t st, bool bf16_out_only = false,
                int n_indep = 0, int layer_from = 0, int layer_to = 1 << 30, PieceState* ln_state = nullptr) {
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
    e.x_first_only = shared_res ? 1 : 0;
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
                      i + 1 == nl && bf16_out_only, Xn, Xn ? n_indep : 0, q1, i + 1 < nl ? &nq : nullptr, &qkv_done));
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
               hipStream_t st, const float* table = nullptr) {
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
    gemm_out(p, DT_F32, h->G, d);
    gemm(h, p, dt, st);
    HeadOutP ho;
    memset(&ho, 0, sizeof(ho));
    ho.X = h->G;
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
        ActorP a;
        memset(&a, 0, sizeof(a));
        a.X = h->G;
        a.ldx = d;
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

// ---------------------------------------------------------------------------------- shared decoder tables
// toks: the decoder tokens (key * T + t) of the query set, group by group; hh: what the cached set is keyed on
int build_query_list(m3pc_handle* h, Plan* pl, int qi, int hh, const std::vector<int>& toks, int n_groups, int key0, int key1) {
    Plan::Query& q = pl->query[qi];
    if (q.built && q.h == hh) return 0;
    const int T = h->T;
    q.h = hh;
    q.nq = (int)toks.size();
    q.n_groups = n_groups;
    q.grp = q.nq / n_groups;
    q.qkeys[0] = key0;
    q.qkeys[1] = key1;
    if (q.nq < 1 || q.nq > 2 * T || q.grp * n_groups != q.nq) return fail(M3PC_EINVAL, "bad query set (%d tokens, %d groups)", q.nq, n_groups);
    std::vector<int> tab(q.nq), mix(q.nq);
    q.all_masked = true;
    for (int j = 0; j < q.nq; ++j) {
        const int i = toks[j];
        tab[j] = -i - 1;
        if (pl->dec_src[i] >= 0) {
            q.all_masked = false;
            mix[j] = pl->dec_src[i];
        } else {
            mix[j] = -i - 1;
        }
    }
    q.nu = 0;
    if (!q.all_masked) {
        int nu = 0;
        while (nu < q.nq && pl->dec_src[toks[nu]] >= 0) ++nu;
        bool ok = nu > 0;
        for (int j = nu; j < q.nq && ok; ++j) ok = pl->dec_src[toks[j]] < 0;                       // a prefix, nothing behind it
        for (int j = 1; j < nu && ok; ++j)                                                           // consecutive tokens / rows of one key
            ok = toks[j] == toks[0] + j && toks[j] / T == toks[0] / T && pl->dec_src[toks[j]] == pl->dec_src[toks[0]] + j;
        if (ok) {
            q.nu = nu;
            q.nu_key = toks[0] / T;
            q.nu_enc0 = pl->dec_src[toks[0]];
            q.nu_kept0 = q.nu_enc0 - pl->enc_off[q.nu_key];  // index among the key's kept tokens (the compact position table's row)
        }
    }
    if (!q.d_q_rowsrc_tab) {
        CHK(dmalloc(&q.d_q_rowsrc_tab, (size_t)2 * T));
        CHK(dmalloc(&q.d_q_rowsrc_mix, (size_t)2 * T));
    }
    HIPCHK(hipMemcpy(q.d_q_rowsrc_tab, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(q.d_q_rowsrc_mix, mix.data(), mix.size() * sizeof(int), hipMemcpyHostToDevice));
    for (int pr = 0; pr < 2; ++pr) q.tab[pr].valid = false;
    q.built = true;
    return 0;
}

// the two scored keys of a plan step at positions idx .. T-1: qi 0 = rtg (rewards, returns), 1 = critic (states, rewards)
int build_query(m3pc_handle* h, Plan* pl, int qi, int hh) {
    if (pl->query[qi].built && pl->query[qi].h == hh) return 0;
    const int T = h->T, idx = T - hh;
    const int k0 = qi == 0 ? M3PC_REWARDS : M3PC_STATES, k1 = qi == 0 ? M3PC_RETURNS : M3PC_REWARDS;
    std::vector<int> toks;
    for (int s = 0; s < 2; ++s)
        for (int t = 0; t < hh; ++t) toks.push_back((s == 0 ? k0 : k1) * T + idx + t);
    return build_query_list(h, pl, qi, hh, toks, 2, k0, k1);
}

// Candidate-independent decoder rows for plan `pl`: run decoder-embed, LN1 and the QKV projection on a
// single sequence whose un-masked slots are zero (never read) and masked slots hold the mask tokens.
int build_tables(m3pc_handle* h, Plan* pl, int qi, int dt, hipStream_t st) {
    Plan::Query& q = pl->query[qi];
    SharedTables& tb = q.tab[dt];
    if (tb.valid) return 0;
    const int T = h->T, d = h->d;
    const size_t es = dtype_size(dt);
    if (!tb.Yall) {
        CHK(dmalloc(&tb.Yall, (size_t)4 * T * d));
        CHK(dmalloc((char**)&tb.QKVm, (size_t)4 * T * 3 * d * es));
        CHK(dmalloc((char**)&tb.QKVq, (size_t)2 * T * 3 * d * es));
        CHK(dmalloc(&tb.Yq, (size_t)2 * T * d));
    }
    // Z: mask tokens everywhere (un-masked rows are ignored downstream)
    std::vector<int> rs(4 * T);
    for (int i = 0; i < 4 * T; ++i) rs[i] = -(i / T) - 1;
    int* d_rs = nullptr;
    CHK(dmalloc(&d_rs, rs.size()));
    HIPCHK(hipMemcpyAsync(d_rs, rs.data(), rs.size() * sizeof(int), hipMemcpyHostToDevice, st));
    GatherP g;
    memset(&g, 0, sizeof(g));
    g.table = h->mask_tokens;
    g.rowsrc = d_rs;
    g.rows_per_batch = 4 * T;
    g.batch = 1;
    g.d = d;
    if (dt == DT_BF16)
        g.outb = (bf16_t*)h->Z;
    else
        g.out = (float*)h->Z;
    launch_gather_rows(g, st);
    for (int k = 0; k < 4; ++k) {
        RowMap m{T, 4 * T, k * T};
        dec_embed(h, k, h->Z, m, tb.Yall, m, T, T, dt, st);
    }
    const std::string pfx = "decoder.layers.0";
    LnP ln;
    memset(&ln, 0, sizeof(ln));
    ln.X = tb.Yall;
    ln.ldx = d;
    ln.rows = 4 * T;
    ln.d = d;
    ln.g1 = W(h, pfx + ".norm1.weight").f;
    ln.b1 = W(h, pfx + ".norm1.bias").f;
    if (dt == DT_BF16)
        ln.Yb = (bf16_t*)h->Hn;
    else
        ln.Yf = (float*)h->Hn;
    launch_layernorm(ln, st);
    // full q|k|v rows of the 4T-token sequence go to h->QKV (fp32 copy for the gather), then compacted
    {
        GemmP p = gemm_basic(h->Hn, d, Wop(h, pfx + ".self_attn.in_proj_weight", dt), d, 4 * T, 3 * d, d,
                             W(h, pfx + ".self_attn.in_proj_bias").f);
        gemm_out(p, DT_F32, h->QKV, 3 * d);
        gemm(h, p, dt, st);
    }
    g.table = (const float*)h->QKV;
    g.d = 3 * d;
    g.rowsrc = pl->d_masked_rowsrc;
    g.rows_per_batch = pl->Lm;
    g.out = dt == DT_F32 ? (float*)tb.QKVm : nullptr;
    g.outb = dt == DT_BF16 ? (bf16_t*)tb.QKVm : nullptr;
    launch_gather_rows(g, st);
    g.rowsrc = q.d_q_rowsrc_tab;
    g.rows_per_batch = q.nq;
    g.out = dt == DT_F32 ? (float*)tb.QKVq : nullptr;
    g.outb = dt == DT_BF16 ? (bf16_t*)tb.QKVq : nullptr;
    launch_gather_rows(g, st);
    g.table = tb.Yall;
    g.d = d;
    g.out = tb.Yq;
    g.outb = nullptr;
    launch_gather_rows(g, st);
    static const bool no_prestats = M3PC_ENV("M3PC_NO_PRESTATS") != nullptr;  // A/B switch
    if (dt == DT_BF16 && q.all_masked && pl->Lm > 0 && pl->Lm <= 256 && !no_prestats) {
        // queries and masked-token keys are both candidate-independent: reduce that block of the softmax once
        if (!tb.pre_m) {
            CHK(dmalloc(&tb.pre_m, (size_t)h->nh * q.nq));
            CHK(dmalloc(&tb.pre_l, (size_t)h->nh * q.nq));
            CHK(dmalloc(&tb.pre_O, (size_t)h->nh * q.nq * h->hd));
        }
        AttnP at;
        memset(&at, 0, sizeof(at));
        at.Q = tb.QKVq;
        at.ldq = 3 * d;
        at.Lq = q.nq;
        at.K2 = (const char*)tb.QKVm + (size_t)d * es;
        at.V2 = (const char*)tb.QKVm + (size_t)2 * d * es;
        at.ldkv2 = 3 * d;
        at.L2 = pl->Lm;
        at.n_head = h->nh;
        at.hd = h->hd;
        at.scale = 1.0f / sqrtf((float)h->hd);
        launch_attention_prestats(at, tb.pre_m, tb.pre_l, tb.pre_O, st);
    }
    HIPCHK(hipStreamSynchronize(st));
    hipFree(d_rs);
    tb.valid = true;
    return check_launch("tables");
}

enum { TAIL_HEADS = 0, TAIL_X = 1 };
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
                      (double)n * nq * h->pass_scale >= (double)FUSED_MIN_ROWS && (long long)nq + (long long)n * q.nu <= h->R;
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
    for (int k = 0; k < 4; ++k) {
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
    for (int s = 0; s < q.n_groups && tail == TAIL_HEADS; ++s) {
        RowMap xm{hh, nq, s * hh};
        CHK(run_head(h, q.qkeys[s], Y1, xm, n * hh, h->pred[s], h->feat[q.qkeys[s]], true, dt, st));
    }
    }
    return check_launch("pruned_decoder");
}

// ---------------------------------------------------------------------------------- candidate pass
// widx (optional, device (n,)): candidate c belongs to history window widx[c] of states (., T, S) / rewards (., T, 1);
// without it all candidates share window 0 and the history tokens are computed once (first-layer sharing).
// stage_from / stage_to / ln_state: the pass can be enqueued in pieces -- stage k < n_enc_layer is encoder layer k (the
// embedding goes with stage 0), stage n_enc_layer everything behind the encoder -- so that the pieces of two candidate halves
// can be enqueued alternately (m3pc_plan_step)
int candidate_pass(m3pc_handle* h, const m3pc_plan_args* a, const float* states, const float* rewards, int n,
                   const float* sample_actions, float* expect_return, float* pred_rewards, float* pred_boot, int dt,
                   hipStream_t st, const int* widx = nullptr, int stage_from = 0, int stage_to = 1 << 30, PieceState* ln_state = nullptr) {
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

int find_tensor(const m3pc_named_tensor* list, int n, const std::string& name) {
    for (int i = 0; i < n; ++i)
        if (list[i].name && name == list[i].name) return i;
    return -1;
}

}  // namespace

// =====================================================================================================
static int fill_rtok(m3pc_handle* h, const double* rtg, int n_windows, hipStream_t st);

// the library is built with -fvisibility=hidden: the C ABI below (include/m3pc_hip.h, and in the lab build m3pc_hip_debug.h) is
// everything it exports
#pragma GCC visibility push(default)
extern "C" {

const char* m3pc_last_error(void) { return g_err; }
int m3pc_abi_version(void) { return M3PC_ABI_VERSION; }

int m3pc_create(const m3pc_dims* dims, int device, m3pc_handle** out) {
    if (!dims || !out) return fail(M3PC_EINVAL, "null argument");
    const m3pc_dims& D = *dims;
    if (D.n_embd % 64 || D.n_embd > 1024 || D.n_head <= 0 || D.n_embd % D.n_head)
        return fail(M3PC_EINVAL, "n_embd must be a multiple of 64 (<=1024) and divisible by n_head");
    const int hd = D.n_embd / D.n_head;
    if (hd != 32 && hd != 64 && hd != 128) return fail(M3PC_EINVAL, "head_dim %d unsupported (32, 64, 128)", hd);
    if (D.state_dim < 1 || D.state_dim > 32 || D.action_dim < 1 || D.action_dim > 32)
        return fail(M3PC_EINVAL, "state_dim/action_dim must be in [1,32]");
    if (D.traj_length < 1 || D.traj_length > 64) return fail(M3PC_EINVAL, "traj_length must be in [1,64]");
    if (D.n_dec_layer != 1) return fail(M3PC_EINVAL, "n_dec_layer must be 1 (every shipped m3pc config)");
    if (D.n_enc_layer < 1 || D.max_candidates < 1 || D.max_batch < 1) return fail(M3PC_EINVAL, "bad sizes");
    if (D.critic_hidden < 0 || D.critic_hidden > 256) return fail(M3PC_EINVAL, "critic_hidden must be <= 256");
    if (D.max_goal_batch < 0) return fail(M3PC_EINVAL, "max_goal_batch must be >= 0");
    HIPCHK(hipSetDevice(device));
    std::unique_ptr<m3pc_handle> h(new m3pc_handle());
    h->dm = D;
    h->device = device;
    h->d = D.n_embd;
    h->nh = D.n_head;
    h->hd = hd;
    h->T = D.traj_length;
    h->S = D.state_dim;
    h->A = D.action_dim;
    h->ff = 4 * D.n_embd;
    h->feat[0] = D.state_dim;
    h->feat[1] = D.action_dim;
    h->feat[2] = 1;
    h->feat[3] = 1;
    declare_weights(h.get());
    for (auto& kv : h->w) {
        CHK(dmalloc(&kv.second.f, (size_t)kv.second.numel));
        if (kv.second.gemm) CHK(dmalloc(&kv.second.b, (size_t)kv.second.numel));
    }
    const int d = h->d, T = h->T;
    for (int k = 0; k < 4; ++k) {
        CHK(dmalloc(&h->WT[k], (size_t)h->feat[k] * d));
        CHK(dmalloc(&h->Eenc[k], (size_t)T * d));
        CHK(dmalloc(&h->Edec[k], (size_t)T * d));
        CHK(dmalloc(&h->tok_mean[k], 32));
        CHK(dmalloc(&h->tok_std[k], 32));
    }
    CHK(dmalloc(&h->mask_tokens, (size_t)4 * d));
    // candidate workspace: max_candidates candidates of 2T token rows (or max_batch generic forwards of 4T)
    {
        const long long r1 = (long long)D.max_candidates * 2 * T, r2 = (long long)D.max_batch * 4 * T;
        const long long r3 = (long long)D.max_goal_batch * 2 * T;  // (a goal window keeps at most 2T - 1 tokens, reads at most T)
        long long R = r1 > r2 ? r1 : r2;
        if (r3 > R) R = r3;
        if (R < 4 * T) R = 4 * T;
        CHK(alloc_ws(h.get(), h->base, R, D.max_candidates > D.max_goal_batch ? D.max_candidates : D.max_goal_batch, 64LL << 20));
        if (D.max_goal_batch > 0) CHK(dmalloc(&h->goal_ws, (size_t)D.max_goal_batch * T * h->S));
    }
    // chain workspaces: fp32 re-scores (<= max_rescore candidates) / policy passes (batch <= max_batch)
    {
        const int mr = D.max_rescore > 0 ? D.max_rescore : 64;
        long long R = (long long)mr * 2 * T;
        if (R < 4 * T) R = 4 * T;
        for (int par = 0; par < 2; ++par) {
            CHK(alloc_ws(h.get(), h->chain[par], R, mr, 32LL << 20));
            CHK(alloc_ws(h.get(), h->pchain[par], (long long)D.max_batch * 4 * T, 1, 32LL << 20));
        }
    }
    for (int s = 0; s < M3PC_SLOTS; ++s) {
        CHK(dmalloc(&h->slot[s].loc, (size_t)D.max_batch * T * h->A + 64));
        CHK(dmalloc(&h->slot[s].sd, (size_t)D.max_batch * T * h->A + 64));
        CHK(dmalloc(&h->slot[s].rtok, (size_t)D.max_batch * T));
    }
    CHK(dmalloc(&h->sel_scratch, 64));
    CHK(dmalloc(&h->d_topk, 1024));
    CHK(dmalloc(&h->er_top, 1024));
    CHK(dmalloc(&h->sa_buf, (size_t)(D.max_candidates > h->chain[0].max_cand ? D.max_candidates : h->chain[0].max_cand) * T * h->A));
    CHK(dmalloc(&h->sa_chain[0], (size_t)h->chain[0].max_cand * T * h->A));
    CHK(dmalloc(&h->sa_chain[1], (size_t)h->chain[0].max_cand * T * h->A));
    bind_ws(h.get(), &h->base);
    bind_slot(h.get(), 0);
    // ONE extra stream per device for all handles of the process: a process has four hardware queues, and with more
    // streams than that two of them share a queue and stop overlapping (a second planner must not cost the first its halves)
    {
        static std::map<int, hipStream_t> shared_aux;
        static std::mutex shared_aux_mutex;  // (handles of different threads may be created at the same time)
        std::lock_guard<std::mutex> lock(shared_aux_mutex);
        hipStream_t& sa = shared_aux[device];
        if (!sa) HIPCHK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
        h->aux = sa;
    }
    HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    for (int s = 0; s < M3PC_SLOTS; ++s)
        for (int i = 0; i < 3; ++i) HIPCHK(hipEventCreateWithFlags(&h->slot_join[s][i], hipEventDisableTiming));
    for (int i = 0; i < 3; ++i) HIPCHK(hipEventCreateWithFlags(&h->aux_tail[i], hipEventDisableTiming));
    if (const char* e = M3PC_ENV("M3PC_TWO_STREAM")) h->two_stream = atoi(e) != 0;
    h->auxs.push_back(h->aux);
    h->ev_joins.push_back(h->ev_join);
    // (lab: more than two candidate parts.  Streams are created only when asked for: a process has four hardware queues,
    // with more streams than that two of them share a queue and the halves no longer overlap)
    if (const char* e = M3PC_ENV("M3PC_STREAM_SPLIT")) {
        for (const char* q = e; *q;) {
            h->stream_split.push_back(atoi(q));
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
        }
        for (size_t i = 1; i < h->stream_split.size() && i < 3; ++i) {
            hipStream_t s2;
            hipEvent_t e2;
            HIPCHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
            h->auxs.push_back(s2);
            h->ev_joins.push_back(e2);
        }
    }
    if (D.critic_hidden > 0) {
        const int Hd = D.critic_hidden, SA = h->S + h->A;
        for (int i = 0; i < 2; ++i) {
            CHK(dmalloc(&h->cW1T[i], (size_t)SA * Hd));
            CHK(dmalloc(&h->cb1[i], Hd));
            CHK(dmalloc(&h->cW2T[i], (size_t)Hd * Hd));
            CHK(dmalloc(&h->cb2[i], Hd));
            CHK(dmalloc(&h->cW3[i], Hd));
            CHK(dmalloc(&h->cb3[i], 4));
            if (critic_mfma_covers(h->S, h->A, Hd)) {
                CHK(dmalloc(&h->cW1F[i], critic_w1f_floats(Hd)));
                CHK(dmalloc(&h->cW2F[i], critic_w2f_floats(Hd)));
            }
        }
        CHK(dmalloc(&h->c_om, 32));
        CHK(dmalloc(&h->c_os, 32));
    }
    *out = h.release();
    return 0;
}

int m3pc_destroy(m3pc_handle* h) {
    if (!h) return 0;
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    for (auto& kv : h->w) {
        hipFree(kv.second.f);
        if (kv.second.b) hipFree(kv.second.b);
    }
    for (int k = 0; k < 4; ++k) {
        hipFree(h->WT[k]);
        hipFree(h->Eenc[k]);
        hipFree(h->Edec[k]);
        hipFree(h->tok_mean[k]);
        hipFree(h->tok_std[k]);
    }
    hipFree(h->mask_tokens);
    free_ws(h->base);
    for (int par = 0; par < 2; ++par) {
        free_ws(h->chain[par]);
        free_ws(h->pchain[par]);
    }
    for (int s = 0; s < M3PC_SLOTS; ++s) {
        hipFree(h->slot[s].loc);
        hipFree(h->slot[s].sd);
        hipFree(h->slot[s].rtok);
    }
    void* bufs[] = {h->sel_scratch, h->d_topk, h->er_top, h->sa_buf, h->sa_chain[0], h->sa_chain[1], h->c_om, h->c_os, h->goal_ws};
    for (size_t i = 1; i < h->auxs.size(); ++i) {
        hipStreamDestroy(h->auxs[i]);
        hipEventDestroy(h->ev_joins[i]);
    }
    for (int s = 0; s < M3PC_SLOTS; ++s)
        for (int i = 0; i < 3; ++i)
            if (h->slot_join[s][i]) hipEventDestroy(h->slot_join[s][i]);
    for (int i = 0; i < 3; ++i)
        if (h->aux_tail[i]) hipEventDestroy(h->aux_tail[i]);
    if (h->aux) hipStreamSynchronize(h->aux);  // (shared by the handles of the device: not destroyed)
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    for (void* b : bufs)
        if (b) hipFree(b);
    for (int i = 0; i < 2; ++i) {
        void* cb[] = {h->cW1T[i], h->cb1[i], h->cW2T[i], h->cb2[i], h->cW3[i], h->cb3[i], h->cW1F[i], h->cW2F[i]};
        for (void* b : cb)
            if (b) hipFree(b);
    }
    for (auto& kv : h->plans) {
        Plan* pl = kv.second.get();
        hipFree(pl->d_tokmap);
        hipFree(pl->d_dec_rowsrc);
        hipFree(pl->d_masked_rowsrc);
        for (int k = 0; k < 4; ++k)
            if (pl->edec_own[k]) hipFree(pl->edec_own[k]);
        for (int q = 0; q < N_QUERY; ++q) {
            if (pl->query[q].d_q_rowsrc_tab) hipFree(pl->query[q].d_q_rowsrc_tab);
            if (pl->query[q].d_q_rowsrc_mix) hipFree(pl->query[q].d_q_rowsrc_mix);
            for (int pr = 0; pr < 2; ++pr) free_tables(pl->query[q].tab[pr]);
        }
    }
    for (auto& e : h->ev) {
        hipEventDestroy(e.a);
        hipEventDestroy(e.b);
    }
    for (int k = 0; k < 4; ++k)
        if (h->kvstream[k]) hipFree(h->kvstream[k]);
    for (auto& kv : h->wstream)
        if (kv.second) hipFree(kv.second);
    delete h;
    return 0;
}

int m3pc_load_weights(m3pc_handle* h, const m3pc_named_tensor* tensors, int n, void* stream) {
    if (!h || !tensors) return fail(M3PC_EINVAL, "null argument");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(h->device));
    CHK(ws_sync(h, st));  // (deferred candidate parts still read the weights this call is about to replace)
    // The first call must bring every tensor; later calls may bring any subset (fine-tuning changes the weights between
    // rollouts, finetune.py:306): only what depends on a tensor that came is re-derived.
    const bool first = !h->weights_loaded;
    std::vector<std::string> dirty;
    for (auto& kv : h->w) {
        const int i = find_tensor(tensors, n, kv.first);
        if (i < 0) {
            if (first) return fail(M3PC_EINVAL, "state_dict is missing '%s'", kv.first.c_str());
            continue;
        }
        if (tensors[i].numel != kv.second.numel)
            return fail(M3PC_EINVAL, "'%s' has %lld elements, expected %lld", kv.first.c_str(), tensors[i].numel, kv.second.numel);
        HIPCHK(hipMemcpyAsync(kv.second.f, tensors[i].data, (size_t)kv.second.numel * sizeof(float),
                              tensors[i].on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        if (kv.second.gemm) launch_f32_to_bf16(kv.second.f, kv.second.b, kv.second.numel, st);
        kv.second.loaded = true;
        dirty.push_back(kv.first);
    }
    auto is_dirty = [&](const std::string& name) {
        for (const std::string& d : dirty)
            if (d == name) return true;
        return false;
    };
    long long* ls = h->load_stats;
    ls[0] = (long long)dirty.size();
    ls[1] = ls[2] = ls[3] = 0;
    if (block_fused_supported(h->d, h->ff)) {  // fragment streams of the fused layer tails (block_fused.hip)
        // nxt: the layer whose Q|K|V projection rides behind this layer's tail (the next encoder layer), or ""
        auto pack = [&](const std::string& pfx, const std::string& nxt) -> int {
            const bool own = is_dirty(pfx + ".self_attn.out_proj.weight") || is_dirty(pfx + ".linear1.weight") || is_dirty(pfx + ".linear2.weight");
            const bool qkv = !nxt.empty() && is_dirty(nxt + ".self_attn.in_proj_weight");
            if (!own && !qkv) return 0;
            bf16_t*& ws = h->wstream[pfx];
            if (!ws) CHK(dmalloc((char**)&ws, block_stream_bytes()));
            if (own)
                launch_pack_block_stream(W(h, pfx + ".self_attn.out_proj.weight").b, W(h, pfx + ".linear1.weight").b,
                                         W(h, pfx + ".linear2.weight").b, ws, st);
            if (qkv) launch_pack_block_qkv(W(h, nxt + ".self_attn.in_proj_weight").b, ws, st);
            ++ls[1];
            return 0;
        };
        for (int i = 0; i < h->dm.n_enc_layer; ++i)
            CHK(pack("encoder.layers." + std::to_string(i), i + 1 < h->dm.n_enc_layer ? "encoder.layers." + std::to_string(i + 1) : ""));
        for (int i = 0; i < h->dm.n_dec_layer; ++i) CHK(pack("decoder.layers." + std::to_string(i), ""));
        // behind the decoder layer's own fragments: the first Linear of the two scalar output heads rtg_guiding scores
        // (rewards, returns: learner.py:294-305), consumed by the fused tail's head phases
        if (h->dm.n_dec_layer >= 1 && h->wstream.count("decoder.layers.0")) {
            const std::string w0 = std::string("output_head_dict.") + KEYN[M3PC_REWARDS] + ".1.weight";
            const std::string w1 = std::string("output_head_dict.") + KEYN[M3PC_RETURNS] + ".1.weight";
            if (is_dirty(w0) || is_dirty(w1)) {
                launch_pack_block_heads(W(h, w0).b, W(h, w1).b, h->wstream["decoder.layers.0"], st);
                ++ls[1];
            }
        }
        if (h->dm.n_dec_layer >= 1)
            for (int k = 0; k < 4; ++k) {
                const std::string we = std::string("decoder_embed_dict.") + KEYN[k] + ".weight";
                if (!(is_dirty(we) || is_dirty("decoder.layers.0.self_attn.in_proj_weight"))) continue;
                if (!h->kvstream[k]) CHK(dmalloc((char**)&h->kvstream[k], kv_stream_bytes()));
                launch_pack_kv_stream(W(h, we).b, W(h, "decoder.layers.0.self_attn.in_proj_weight").b + (size_t)h->d * h->d,
                                      h->kvstream[k], st);
                ++ls[2];
            }
    }
    // small derived tables, on the device: transposed encoder-embed weights, E_enc / E_dec = (bias + per-dim) + pos, mask tokens
    const int d = h->d, T = h->T;
    const bool pos_dirty = is_dirty("pos_embed");
    for (int k = 0; k < 4; ++k) {
        const std::string kn = KEYN[k];
        if (is_dirty("encoder_embed_dict." + kn + ".weight"))
            launch_transpose_f32(W(h, "encoder_embed_dict." + kn + ".weight").f, h->WT[k], d, h->feat[k], st);
        for (int pass = 0; pass < 2; ++pass) {
            const std::string side = pass == 0 ? "encoder" : "decoder";
            if (pos_dirty || is_dirty(side + "_embed_dict." + kn + ".bias") || is_dirty(side + "_per_dim_encoding." + kn))
                launch_embed_table(W(h, side + "_embed_dict." + kn + ".bias").f, W(h, side + "_per_dim_encoding." + kn).f,
                                   W(h, "pos_embed").f, pass == 0 ? h->Eenc[k] : h->Edec[k], T, d, st);
        }
        if (is_dirty("mask_token_dict." + kn))
            HIPCHK(hipMemcpyAsync(h->mask_tokens + (size_t)k * d, W(h, "mask_token_dict." + kn).f, d * sizeof(float),
                                  hipMemcpyDeviceToDevice, st));
    }
    // the candidate-independent decoder tables of every cached plan hang on the decoder side of the model
    bool dec_dirty = pos_dirty;
    for (const std::string& nme : dirty)
        if (nme.rfind("decoder", 0) == 0 || nme.rfind("mask_token_dict", 0) == 0) dec_dirty = true;
    if (dec_dirty) {
        invalidate_tables(h);
        ls[3] = 1;
    }
    for (int sl = 0; sl < M3PC_SLOTS; ++sl) {
        h->slot[sl].policy_valid = false;
        h->slot[sl].n_windows = 0;
    }
    HIPCHK(hipStreamSynchronize(st));  // the caller's tensors may go away when the call returns
    h->weights_loaded = true;
    return check_launch("load_weights");
}

int m3pc_load_stats(m3pc_handle* h, long long* out4) {
    if (!h || !out4) return fail(M3PC_EINVAL, "null argument");
    for (int i = 0; i < 4; ++i) out4[i] = h->load_stats[i];
    return 0;
}

int m3pc_set_tokenizer(m3pc_handle* h, int key, const float* mean, const float* std_, int dim, int normalize) {
    if (!h || key < 0 || key > 3 || !mean || !std_) return fail(M3PC_EINVAL, "bad argument");
    if (dim != h->feat[key]) return fail(M3PC_EINVAL, "tokenizer '%s' has dim %d, expected %d", KEYN[key], dim, h->feat[key]);
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpy(h->tok_mean[key], mean, dim * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->tok_std[key], std_, dim * sizeof(float), hipMemcpyHostToDevice));
    h->h_mean[key].assign(mean, mean + dim);
    h->h_std[key].assign(std_, std_ + dim);
    h->tok_norm[key] = normalize ? 1 : 0;
    h->tok_set[key] = true;
    return 0;
}

int m3pc_set_critic(m3pc_handle* h, const m3pc_named_tensor* tensors, int n, const float* obs_mean, const float* obs_std,
                    void* stream) {
    if (!h || !tensors || !obs_mean || !obs_std) return fail(M3PC_EINVAL, "null argument");
    if (h->dm.critic_hidden <= 0) return fail(M3PC_EINVAL, "handle was created without a critic");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(h->device));
    // device tensors are read after everything already queued on the caller's stream (an optimizer step that has just
    // updated qf, finetune.py:288-290): the blocking copies below are not ordered behind a non-blocking stream by themselves
    HIPCHK(hipStreamSynchronize(st));
    const int Hd = h->dm.critic_hidden, SA = h->S + h->A;
    auto fetch = [&](const std::string& name, long long numel, std::vector<float>& out) -> int {
        const int i = find_tensor(tensors, n, name);
        if (i < 0) return fail(M3PC_EINVAL, "critic state_dict is missing '%s'", name.c_str());
        if (tensors[i].numel != numel) return fail(M3PC_EINVAL, "'%s' has %lld elements, expected %lld", name.c_str(), tensors[i].numel, numel);
        out.resize((size_t)numel);
        HIPCHK(hipMemcpy(out.data(), tensors[i].data, (size_t)numel * sizeof(float),
                         tensors[i].on_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
        return 0;
    };
    for (int qn = 0; qn < 2; ++qn) {
        const std::string q = qn == 0 ? "q1" : "q2";
        std::vector<float> w1, b1, w2, b2, w3, b3;
        CHK(fetch(q + ".net.0.weight", (long long)Hd * SA, w1));
        CHK(fetch(q + ".net.0.bias", Hd, b1));
        CHK(fetch(q + ".net.2.weight", (long long)Hd * Hd, w2));
        CHK(fetch(q + ".net.2.bias", Hd, b2));
        CHK(fetch(q + ".net.4.weight", Hd, w3));
        CHK(fetch(q + ".net.4.bias", 1, b3));
        std::vector<float> w1t((size_t)SA * Hd), w2t((size_t)Hd * Hd);
        for (int c = 0; c < Hd; ++c)
            for (int f = 0; f < SA; ++f) w1t[(size_t)f * Hd + c] = w1[(size_t)c * SA + f];
        for (int c = 0; c < Hd; ++c)
            for (int k = 0; k < Hd; ++k) w2t[(size_t)k * Hd + c] = w2[(size_t)c * Hd + k];
        HIPCHK(hipMemcpy(h->cW1T[qn], w1t.data(), w1t.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->cb1[qn], b1.data(), b1.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->cW2T[qn], w2t.data(), w2t.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->cb2[qn], b2.data(), b2.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->cW3[qn], w3.data(), w3.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->cb3[qn], b3.data(), sizeof(float), hipMemcpyHostToDevice));
        if (h->cW1F[qn]) {
            std::vector<float> w1f(critic_w1f_floats(Hd)), w2f(critic_w2f_floats(Hd));
            critic_pack(w1.data(), w2.data(), SA, Hd, w1f.data(), w2f.data());
            HIPCHK(hipMemcpy(h->cW1F[qn], w1f.data(), w1f.size() * sizeof(float), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(h->cW2F[qn], w2f.data(), w2f.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    HIPCHK(hipMemcpy(h->c_om, obs_mean, h->S * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->c_os, obs_std, h->S * sizeof(float), hipMemcpyHostToDevice));
    h->critic_set = true;
    return 0;
}

int m3pc_tokenize(m3pc_handle* h, int key, const void* in, int in_f64, float* out, long long rows, void* stream) {
    if (!h || key < 0 || key > 3 || !in || !out) return fail(M3PC_EINVAL, "bad argument");
    if (!h->tok_set[key]) return fail(M3PC_ESTATE, "tokenizer '%s' not set", KEYN[key]);
    HIPCHK(hipSetDevice(h->device));
    launch_tokenize(in, in_f64, out, rows, h->feat[key], h->tok_mean[key], h->tok_std[key], h->tok_norm[key], (hipStream_t)stream);
    return check_launch("tokenize");
}

int m3pc_detokenize(m3pc_handle* h, int key, const float* in, float* out, long long rows, void* stream) {
    if (!h || key < 0 || key > 3 || !in || !out) return fail(M3PC_EINVAL, "bad argument");
    if (!h->tok_set[key]) return fail(M3PC_ESTATE, "tokenizer '%s' not set", KEYN[key]);
    HIPCHK(hipSetDevice(h->device));
    launch_detokenize(in, out, rows, h->feat[key], h->tok_mean[key], h->tok_std[key], h->tok_norm[key], (hipStream_t)stream);
    return check_launch("detokenize");
}

int m3pc_forward(m3pc_handle* h, int batch, const float* const tokens[4], const unsigned char* const masks[4],
                 float* out_states, float* out_rewards, float* out_returns, float* out_mu, float* out_std, int precision,
                 void* stream) {
    if (!h || !tokens || !masks) return fail(M3PC_EINVAL, "null argument");
    if (!h->weights_loaded) return fail(M3PC_ESTATE, "weights not loaded");
    if (batch < 1 || batch > h->dm.max_batch) return fail(M3PC_EINVAL, "batch %d outside [1, max_batch=%d]", batch, h->dm.max_batch);
    if (precision != M3PC_PREC_FP32 && precision != M3PC_PREC_BF16) return fail(M3PC_EINVAL, "bad precision");
    if ((out_mu == nullptr) != (out_std == nullptr)) return fail(M3PC_EINVAL, "out_mu and out_std go together");
    HIPCHK(hipSetDevice(h->device));
    Plan* pl = nullptr;
    CHK(get_plan(h, masks, &pl));
    TokIn in;
    memset(&in, 0, sizeof(in));
    for (int k = 0; k < 4; ++k) {
        if (pl->kept[k] && !tokens[k]) return fail(M3PC_EINVAL, "tokens[%s] is null but its mask keeps tokens", KEYN[k]);
        in.ptr[k] = tokens[k];
        in.bstride[k] = (long long)h->T * h->feat[k];
    }
    h->allow_splitk = true;
    CHK(ws_sync(h, (hipStream_t)stream));
    return forward_impl(h, pl, in, batch, out_states, out_rewards, out_returns, out_mu, out_std,
                        precision == M3PC_PREC_BF16 ? DT_BF16 : DT_F32, (hipStream_t)stream);
}

// Zero-shot goal reaching, both forwards of action_piid_sample (zeroshot_omtm/learner.py:151-261) in one call on RAW windows:
// path inference under the pi mask -> the inferred observations over the window rows [0, idx] and [idx+2, T-2] (240-246) ->
// inverse dynamics under the fid mask -> the action distribution.  fp32, in the policy workspace.
int m3pc_goal_step(m3pc_handle* h, int batch, const float* states, const float* actions, const float* rewards, const double* rtg,
                   const unsigned char* const masks_pi[4], const unsigned char* const masks_fid[4], int idx, float* inferred,
                   float* window_states, float* out_mu, float* out_std, void* stream) {
    if (!h || !states || !actions || !rewards || !rtg || !masks_pi || !masks_fid || !inferred || !window_states || !out_mu || !out_std)
        return fail(M3PC_EINVAL, "null argument");
    if (!h->weights_loaded) return fail(M3PC_ESTATE, "weights not loaded");
    for (int k = 0; k < 4; ++k)
        if (!h->tok_set[k]) return fail(M3PC_ESTATE, "tokenizer '%s' not set", KEYN[k]);
    if (batch < 1 || batch > h->dm.max_batch) return fail(M3PC_EINVAL, "batch %d outside [1, max_batch=%d]", batch, h->dm.max_batch);
    const int T = h->T;
    if (idx < 0 || idx >= T) return fail(M3PC_EINVAL, "idx %d outside [0, T=%d)", idx, T);
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(h->device));
    Plan *pi = nullptr, *fid = nullptr;
    CHK(get_plan(h, masks_pi, &pi));
    CHK(get_plan(h, masks_fid, &fid));
    bind_slot(h, 0);
    h->slot[0].policy_valid = false;
    h->slot[0].n_windows = 0;
    CHK(fill_rtok(h, rtg, batch, st));
    TokIn in;
    memset(&in, 0, sizeof(in));
    const float* src[4] = {states, actions, rewards, h->rtok};
    for (int k = 0; k < 4; ++k) {
        in.ptr[k] = src[k];
        in.bstride[k] = (long long)T * h->feat[k];
        in.normalize[k] = k == M3PC_RETURNS ? 0 : h->tok_norm[k];
    }
    h->allow_splitk = true;
    WsScope ws(h, true, true);
    CHK(forward_impl(h, pi, in, batch, inferred, nullptr, nullptr, nullptr, nullptr, DT_F32, st));
    launch_goal_overlay(inferred, states, window_states, (long long)batch * T, T, h->S, idx, h->tok_mean[M3PC_STATES],
                        h->tok_std[M3PC_STATES], h->tok_norm[M3PC_STATES], st);
    in.ptr[M3PC_STATES] = window_states;
    CHK(forward_impl(h, fid, in, batch, nullptr, nullptr, nullptr, out_mu, out_std, DT_F32, st));
    return check_launch("goal_step");
}

// m3pc_goal_step for many windows: the same two forwards, exactly pruned to what the reference reads of them, in the
// arithmetic of the candidate pass (bf16 MFMA or fp32), in the candidate workspace.
//   path inference (pi mask): the states head is read at the window rows t <= idx and idx+2 <= t <= T-2 only
//   (zeroshot_omtm/learner.py:240-246) -- those decoder tokens are the queries; inverse dynamics (fid mask): the action
//   distribution is read at token idx only (learner.py:250-256) -- ONE query per window, a masked token, so its query row is
//   shared by the batch.  Neither mask keeps a rewards or returns token (zeroshot_omtm/masks.py:30-47, 72-91): those rows
//   of the window never enter, which is why the call does not take them.
// goal_mode M3PC_GOAL_ID: action_id_sample (learner.py:60-149) -- the second forward alone, under the gid mask (= pi mask).
// plan tables, query set and candidate-independent decoder rows of one goal forward (cached per idx / weights)
static int goal_prepare(m3pc_handle* h, int kind, int qi, int idx, int dt, hipStream_t st, Plan** pl_out, Plan::Query** q_out) {
    const int T = h->T;
    Plan* pl = nullptr;
    CHK(get_mask_plan(h, kind, idx, &pl));
    std::vector<int> toks;
    if (qi == 2) {
        for (int t = 0; t < T; ++t)
            if (t <= idx || (t >= idx + 2 && t < T - 1)) toks.push_back(M3PC_STATES * T + t);
    } else {
        toks.push_back(M3PC_ACTIONS * T + idx);
    }
    CHK(build_query_list(h, pl, qi, T - idx, toks, 1, qi == 2 ? M3PC_STATES : M3PC_ACTIONS, 0));
    CHK(build_tables(h, pl, qi, dt, st));
    CHK(ensure_edec(h, pl, st));
    if (pl_out) *pl_out = pl;
    if (q_out) *q_out = &pl->query[qi];
    return 0;
}

static int goal_forward(m3pc_handle* h, int kind, int qi, int idx, const float* states, const float* actions, int n, int dt,
                        hipStream_t st, int tail, float** xrows) {
    const int T = h->T;
    Plan* pl = nullptr;
    Plan::Query* qp = nullptr;
    CHK(goal_prepare(h, kind, qi, idx, dt, st, &pl, &qp));
    Plan::Query& q = *qp;
    if ((long long)n * pl->Le > h->R || (long long)n * q.nq > h->R) return fail(M3PC_ENOMEM, "batch %d exceeds workspace", n);
    TokIn in;
    memset(&in, 0, sizeof(in));
    in.ptr[M3PC_STATES] = states;
    in.bstride[M3PC_STATES] = (long long)T * h->S;
    in.normalize[M3PC_STATES] = h->tok_norm[M3PC_STATES];
    in.ptr[M3PC_ACTIONS] = actions;
    in.bstride[M3PC_ACTIONS] = (long long)T * h->A;
    in.normalize[M3PC_ACTIONS] = h->tok_norm[M3PC_ACTIONS];
    CHK(run_encoder(h, pl, in, n, dt, st, dt == DT_BF16, 0));
    return pruned_decoder(h, pl, q, q.tab[dt], n, dt, st, tail, xrows);
}

int m3pc_goal_step_batch(m3pc_handle* h, int batch, const float* states, const float* actions, int idx, int goal_mode,
                         int precision, float* window_states, float* out_mu, float* out_std, void* stream) {
    if (!h || !states || !actions || !out_mu || !out_std) return fail(M3PC_EINVAL, "null argument");
    if (!h->weights_loaded) return fail(M3PC_ESTATE, "weights not loaded");
    for (int k = 0; k < 2; ++k)
        if (!h->tok_set[k]) return fail(M3PC_ESTATE, "tokenizer '%s' not set", KEYN[k]);
    if (batch < 1 || batch > h->dm.max_goal_batch) return fail(M3PC_EINVAL, "batch %d outside [1, max_goal_batch=%d]", batch, h->dm.max_goal_batch);
    if (precision != M3PC_PREC_FP32 && precision != M3PC_PREC_BF16) return fail(M3PC_EINVAL, "bad precision");
    if (goal_mode != M3PC_GOAL_PIID && goal_mode != M3PC_GOAL_ID) return fail(M3PC_EINVAL, "bad goal_mode %d", goal_mode);
    const int T = h->T, d = h->d;
    if (idx < 0 || idx >= T) return fail(M3PC_EINVAL, "idx %d outside [0, T=%d)", idx, T);
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(h->device));
    const int dt = precision == M3PC_PREC_BF16 ? DT_BF16 : DT_F32;
    bind_ws(h, &h->base);
    CHK(ws_sync(h, st));
    // kernels chosen by the row count (split-K, the few-row fp32 kernels) stay off: a window's result must not depend on
    // which other windows share its call (environment sharding, m3pc_amd/dist.py)
    h->allow_splitk = false;
    h->pass_scale = 1.0;
    int nq_a = 0;
    for (int t = 0; t < T; ++t) nq_a += (t <= idx || (t >= idx + 2 && t < T - 1)) ? 1 : 0;
    float* const ws_all = window_states ? window_states : h->goal_ws;
    // windows [c0, c0 + cnt) on stream s, in the workspace rows of those windows (set_view: 2T rows per window)
    auto run_part = [&](int c0, int cnt, hipStream_t s) -> int {
        set_view(h, c0, cnt);
        const float* st_p = states + (size_t)c0 * T * h->S;
        const float* ac_p = actions + (size_t)c0 * T * h->A;
        const float* second = st_p;
        if (goal_mode == M3PC_GOAL_PIID) {
            CHK(goal_forward(h, 2, 2, idx, st_p, ac_p, cnt, dt, s, TAIL_HEADS, nullptr));
            float* ws = ws_all + (size_t)c0 * T * h->S;
            launch_goal_overlay_rows(h->pred[0], st_p, ws, cnt, T, h->S, idx, nq_a, s);
            second = ws;
        } else if (window_states) {
            HIPCHK(hipMemcpyAsync(window_states + (size_t)c0 * T * h->S, st_p, (size_t)cnt * T * h->S * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
        float* xr = nullptr;
        CHK(goal_forward(h, goal_mode == M3PC_GOAL_PIID ? 3 : 2, 3, idx, second, ac_p, cnt, dt, s, TAIL_X, &xr));
        // decoder.norm of the one query row per window, then the action head (mtm_model.py:705, 313-321), fp32
        LnP ln;
        memset(&ln, 0, sizeof(ln));
        ln.X = xr;
        ln.ldx = d;
        ln.rows = cnt;
        ln.d = d;
        ln.g1 = W(h, "decoder.norm.weight").f;
        ln.b1 = W(h, "decoder.norm.bias").f;
        ln.Yf = h->G;
        launch_layernorm(ln, s);
        ActorP ac;
        memset(&ac, 0, sizeof(ac));
        ac.X = h->G;
        ac.ldx = d;
        ac.rows = cnt;
        ac.d = d;
        ac.A = h->A;
        ac.Wmu = W(h, "output_head_dict.actions.mu.weight").f;
        ac.bmu = W(h, "output_head_dict.actions.mu.bias").f;
        ac.Wls = W(h, "output_head_dict.actions.log_std.weight").f;
        ac.bls = W(h, "output_head_dict.actions.log_std.bias").f;
        ac.mu = out_mu + (size_t)c0 * h->A;
        ac.sd = out_std + (size_t)c0 * h->A;
        launch_actor_head(ac, s);
        return 0;
    };
    // Many windows in bf16: two halves on two streams, as the candidate pass runs its halves -- the encoder launches of the
    // whole call are 2.5 and 3 rounds of fused-tail tiles, and the other half's attention / embedding kernels (HBM-bound) run
    // beside a half's tiles.  The kernel choice goes by the size of the whole call (pass_scale): same bits either way.
    int rc = 0;
    if (h->two_stream && dt == DT_BF16 && batch >= 2048 && !h->prof_serial) {
        const int n0 = ((batch / 2 + 63) / 64) * 64;
        // what is built once per (weights, idx) -- plan tables, query sets, the shared decoder rows -- before the streams fork
        if (goal_mode == M3PC_GOAL_PIID) CHK(goal_prepare(h, 2, 2, idx, dt, st, nullptr, nullptr));
        CHK(goal_prepare(h, goal_mode == M3PC_GOAL_PIID ? 3 : 2, 3, idx, dt, st, nullptr, nullptr));
        HIPCHK(hipEventRecord(h->ev_fork, st));
        HIPCHK(hipStreamWaitEvent(h->aux, h->ev_fork, 0));
        h->pass_scale = (double)batch / (double)n0;
        rc = run_part(0, n0, st);
        h->pass_scale = (double)batch / (double)(batch - n0);
        if (rc == 0) rc = run_part(n0, batch - n0, h->aux);
        HIPCHK(hipEventRecord(h->ev_join, h->aux));
        HIPCHK(hipStreamWaitEvent(st, h->ev_join, 0));
    } else {
        rc = run_part(0, batch, st);
    }
    h->pass_scale = 1.0;
    set_view(h, 0, h->base.max_cand);
    if (rc) return rc;
    return check_launch("goal_step_batch");
}

// common argument checks of the plan-step entry points; binds the step's slot
static int plan_check(m3pc_handle* h, const m3pc_plan_args* a, bool need_critic) {
    if (!h->weights_loaded) return fail(M3PC_ESTATE, "weights not loaded");
    for (int k = 0; k < 4; ++k)
        if (!h->tok_set[k]) return fail(M3PC_ESTATE, "tokenizer '%s' not set", KEYN[k]);
    const int T = h->T;
    if (a->horizon < 1 || a->horizon > T) return fail(M3PC_EINVAL, "horizon %d outside [1, T=%d]", a->horizon, T);
    if (a->mode < 0 || a->mode > 2) return fail(M3PC_EINVAL, "bad mode %d", a->mode);
    if (a->precision != M3PC_PREC_FP32 && a->precision != M3PC_PREC_BF16) return fail(M3PC_EINVAL, "bad precision");
    if (a->slot < 0 || a->slot >= M3PC_SLOTS) return fail(M3PC_EINVAL, "slot %d outside [0, %d)", a->slot, M3PC_SLOTS);
    if (a->flags & ~M3PC_PLAN_DEFER_JOIN)  // (also what a caller built against the shorter ABI v2 structure would hand over)
        return fail(M3PC_EINVAL, "unknown m3pc_plan_args::flags 0x%x (is the caller's structure the ABI v%d one?)", a->flags, M3PC_ABI_VERSION);
    if (need_critic && a->mode != M3PC_MODE_RTG && !h->critic_set) return fail(M3PC_ESTATE, "critic weights not set");
    HIPCHK(hipSetDevice(h->device));
    bind_slot(h, a->slot);
    return 0;
}

// PASS 1 of a plan step (learner.py:278-284): the returns tokens of the window, then the return-conditioned policy at
// batch 1 under the rcbc mask (finetune_omtm/masks.py:7-27), always fp32, in the chain workspace; leaves loc / sd / rtok in
// the step's slot.
int m3pc_policy_pass(m3pc_handle* h, const m3pc_plan_args* a, const float* states, const float* actions, const float* rewards,
                     float* loc, float* std_, void* stream) {
    if (!h || !a || !states || !actions || !rewards) return fail(M3PC_EINVAL, "null argument");
    CHK(plan_check(h, a, false));
    hipStream_t st = (hipStream_t)stream;
    const int T = h->T, hh = a->horizon, idx = T - hh;
    if (a->returns) {
        // the caller's returns row (learner.py:272-293 consumes whatever trajectory["returns"] holds): tokenised as
        // ContinuousTokenizer.encode does, in the row's own dtype, then cast (continuous.py:74-79)
        launch_tokenize(a->returns, a->returns_f64, h->rtok, T, 1, h->tok_mean[M3PC_RETURNS], h->tok_std[M3PC_RETURNS],
                        h->tok_norm[M3PC_RETURNS], st);
    } else {
        // constant return-to-go: float64 normalisation then cast (learner.py:371-374, continuous.py:74-79)
        double rt = a->rtg;
        if (h->tok_norm[M3PC_RETURNS]) rt = (rt - (double)h->h_mean[M3PC_RETURNS][0]) / (double)h->h_std[M3PC_RETURNS][0];
        launch_fill(h->rtok, (float)rt, T, st);
    }
    Plan* pl = nullptr;
    CHK(get_mask_plan(h, 0, idx, &pl));
    TokIn in;
    memset(&in, 0, sizeof(in));
    in.ptr[M3PC_STATES] = states;
    in.normalize[M3PC_STATES] = h->tok_norm[M3PC_STATES];
    in.ptr[M3PC_ACTIONS] = actions;
    in.normalize[M3PC_ACTIONS] = h->tok_norm[M3PC_ACTIONS];
    in.ptr[M3PC_REWARDS] = rewards;
    in.normalize[M3PC_REWARDS] = h->tok_norm[M3PC_REWARDS];
    in.ptr[M3PC_RETURNS] = h->rtok;
    h->allow_splitk = true;
    int rc;
    {
        WsScope ws(h, true, true, a->slot);
        rc = forward_impl(h, pl, in, 1, nullptr, nullptr, nullptr, h->loc, h->sd, DT_F32, st);
    }
    h->allow_splitk = false;
    if (rc) return rc;
    h->slot[a->slot].policy_valid = true;
    h->slot[a->slot].n_windows = 1;
    if (loc) HIPCHK(hipMemcpyAsync(loc, h->loc, (size_t)T * h->A * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (std_) HIPCHK(hipMemcpyAsync(std_, h->sd, (size_t)T * h->A * sizeof(float), hipMemcpyDeviceToDevice, st));
    return check_launch("policy_pass");
}

// Candidates + PASS 2 + scoring of a plan step (learner.py:285-316) from the slot's policy head, in the candidate workspace.
int m3pc_candidate_pass(m3pc_handle* h, const m3pc_plan_args* a, const float* states, const float* actions, const float* rewards,
                        const float* eps, float* loc, float* std_, float* sample_actions, float* expect_return,
                        float* pred_rewards, float* pred_boot, void* stream) {
    if (!h || !a || !states || !actions || !rewards || !eps || !sample_actions || !expect_return)
        return fail(M3PC_EINVAL, "null argument");
    CHK(plan_check(h, a, true));
    if (a->window < 0 || a->window >= h->slot[a->slot].n_windows)
        return fail(M3PC_ESTATE, "m3pc_candidate_pass: slot %d holds %d policy pass(es), window %d asked for (m3pc_policy_pass[_batch] first)",
                    a->slot, h->slot[a->slot].n_windows, a->window);
    const int T = h->T;
    if (a->n_count < 1 || a->n_begin < 0 || a->n_begin + a->n_count > a->n_total)
        return fail(M3PC_EINVAL, "candidate range [%d,+%d) outside n_total=%d", a->n_begin, a->n_count, a->n_total);
    if (a->n_count > h->dm.max_candidates) return fail(M3PC_ENOMEM, "n_count %d > max_candidates %d", a->n_count, h->dm.max_candidates);
    hipStream_t st = (hipStream_t)stream;
    const int hh = a->horizon, idx = T - hh;
    bind_ws(h, &h->base);
    h->allow_splitk = false;

    // candidates of [c0, c0 + cnt) (relative to n_begin), on the stream of the part that scores them
    auto sample = [&](int c0, int cnt, hipStream_t s) {
        SampleP sp;
        memset(&sp, 0, sizeof(sp));
        sp.hist_actions = actions;
        sp.loc = h->loc + (size_t)a->window * T * h->A;
        sp.sd = h->sd + (size_t)a->window * T * h->A;
        sp.eps = eps;
        sp.mode = a->mode == M3PC_MODE_NOISE ? 1 : 0;
        sp.T = T;
        sp.A = h->A;
        sp.idx = idx;
        sp.h = hh;
        sp.n_begin = a->n_begin + c0;
        sp.n_count = cnt;
        sp.cand = h->base.cand + (size_t)c0 * T * h->A;
        sp.sample_actions = sample_actions + (size_t)c0 * hh * h->A;
        if (c0 == 0) {  // the caller's copies of the policy head ride on the first launch
            sp.loc_out = loc;
            sp.sd_out = std_;
        }
        launch_sample(sp, s);
    };

    // PASS 2 + scoring.  Large bf16 batches are cut into two candidate halves that run the same kernel chain
    // on two HIP streams over disjoint workspace halves.  The fused layer tails work in 128-row tiles, one per CU:
    // 1024 candidates are 392 tiles = two rounds on 256 CUs with the second round half empty, a half is 196 tiles =
    // one round, and the other half's attention / projection kernels run on the CUs it leaves free.  Candidates are
    // independent, results are identical to the one-stream order.
    const int dt = a->precision == M3PC_PREC_BF16 ? DT_BF16 : DT_F32;
    const int n = a->n_count;
    h->slot_join_n[a->slot] = 0;
    // (only when one half alone fills the chip with fused-tail tiles: more than 256 tiles of 128 rows in the whole pass)
    if (h->two_stream && dt == DT_BF16 && n >= 512 && (long long)n * (2 * T - hh + 1) > 256 * 128) {
        // part sizes: M3PC_STREAM_SPLIT=a,b,c (lab) or two halves
        std::vector<int> parts;
        if (!h->stream_split.empty()) {
            int left = n;
            for (int v : h->stream_split)
                if (v > 0 && v < left && parts.size() + 1 < h->auxs.size() + 1) {
                    parts.push_back(v);
                    left -= v;
                }
            parts.push_back(left);
        } else {
            int n0 = ((n / 2 + 127) / 128) * 128;
            parts = {n0, n - n0};
        }
        // M3PC_PLAN_DEFER_JOIN: the caller's stream does not wait for the other parts (m3pc_candidate_join does, for the
        // consumer), and a pass of the same part sizes as the deferred ones before it starts without waiting for them either:
        // per stream it touches the rows that stream's own earlier work touched.
        const bool defer = (a->flags & M3PC_PLAN_DEFER_JOIN) != 0 && !h->prof_serial && parts.size() <= 4;
        if (!(defer && parts == h->defer_parts)) CHK(ws_sync(h, st));
        HIPCHK(hipEventRecord(h->ev_fork, st));
        int rc = 0;
        // enqueued stage by stage, alternating between the parts: the host needs ~1.5 us per launch, and a part whose 45
        // launches are all enqueued behind the other part's starts that much later on the device -- and ends that much later,
        // with the other stream idle.  (With the profiling brackets in serial mode: one part after the other.)
        const int n_stage = h->dm.n_enc_layer + 1;
        PieceState lnst[4];
        for (int stg = 0; stg < (h->prof_serial ? 1 : n_stage) && rc == 0; ++stg) {
            int c0 = 0;
            for (size_t i = 0; i < parts.size() && rc == 0; ++i) {
                hipStream_t s = i == 0 || h->prof_serial ? st : h->auxs[i - 1];
                if (stg == 0) {
                    if (s != st) HIPCHK(hipStreamWaitEvent(s, h->ev_fork, 0));
                    sample(c0, parts[i], s);
                }
                set_view(h, c0, parts[i]);
                rc = candidate_pass(h, a, states, rewards, parts[i], sample_actions + (size_t)c0 * hh * h->A, expect_return + c0,
                                    pred_rewards ? pred_rewards + (size_t)c0 * hh : nullptr,
                                    pred_boot ? pred_boot + (size_t)c0 * hh : nullptr, dt, s, nullptr,
                                    h->prof_serial ? 0 : stg, h->prof_serial ? 1 << 30 : stg + 1, &lnst[i]);
                if (s != st && stg == n_stage - 1 && rc == 0) {
                    if (defer) {
                        HIPCHK(hipEventRecord(h->slot_join[a->slot][i - 1], s));
                        HIPCHK(hipEventRecord(h->aux_tail[i - 1], s));
                        h->aux_unjoined[i - 1] = true;
                        h->slot_join_n[a->slot] = (int)i;
                    } else {
                        HIPCHK(hipEventRecord(h->ev_joins[i - 1], s));
                        HIPCHK(hipStreamWaitEvent(st, h->ev_joins[i - 1], 0));
                    }
                }
                c0 += parts[i];
            }
        }
        if (defer && rc == 0) h->defer_parts = parts;
        set_view(h, 0, h->dm.max_candidates);
        return rc;
    }
    CHK(ws_sync(h, st));
    sample(0, n, st);
    return candidate_pass(h, a, states, rewards, n, sample_actions, expect_return, pred_rewards, pred_boot, dt, st);
}

int m3pc_candidate_join(m3pc_handle* h, int slot, void* stream) {
    if (!h) return fail(M3PC_EINVAL, "null handle");
    if (slot < 0 || slot >= M3PC_SLOTS) return fail(M3PC_EINVAL, "slot %d outside [0, %d)", slot, M3PC_SLOTS);
    for (int i = 0; i < h->slot_join_n[slot]; ++i) HIPCHK(hipStreamWaitEvent((hipStream_t)stream, h->slot_join[slot][i], 0));
    h->slot_join_n[slot] = 0;
    return 0;
}

// m3pc_policy_pass + m3pc_candidate_pass on one stream
int m3pc_plan_step(m3pc_handle* h, const m3pc_plan_args* a, const float* states, const float* actions, const float* rewards,
                   const float* eps, float* loc, float* std_, float* sample_actions, float* expect_return,
                   float* pred_rewards, float* pred_boot, void* stream) {
    if (!h || !a || !states || !actions || !rewards || !eps || !sample_actions || !expect_return)
        return fail(M3PC_EINVAL, "null argument");
    if (a->mode >= 0 && a->mode <= 2 && a->mode != M3PC_MODE_RTG && !h->critic_set) return fail(M3PC_ESTATE, "critic weights not set");
    if (a->n_count < 1 || a->n_begin < 0 || a->n_begin + a->n_count > a->n_total)
        return fail(M3PC_EINVAL, "candidate range [%d,+%d) outside n_total=%d", a->n_begin, a->n_count, a->n_total);
    if (a->n_count > h->dm.max_candidates) return fail(M3PC_ENOMEM, "n_count %d > max_candidates %d", a->n_count, h->dm.max_candidates);
    CHK(m3pc_policy_pass(h, a, states, actions, rewards, nullptr, nullptr, stream));
    return m3pc_candidate_pass(h, a, states, actions, rewards, eps, loc, std_, sample_actions, expect_return, pred_rewards,
                               pred_boot, stream);
}

// fills h->rtok[w * T + t] with window w's normalised return-to-go (float64 normalisation then cast: learner.py:371-374,
// continuous.py:74-79)
static int fill_rtok(m3pc_handle* h, const double* rtg, int n_windows, hipStream_t st) {
    for (int w = 0; w < n_windows; ++w) {
        double rt = rtg[w];
        if (h->tok_norm[M3PC_RETURNS]) rt = (rt - (double)h->h_mean[M3PC_RETURNS][0]) / (double)h->h_std[M3PC_RETURNS][0];
        launch_fill(h->rtok + (size_t)w * h->T, (float)rt, h->T, st);
    }
    return 0;
}

int m3pc_score_actions(m3pc_handle* h, const m3pc_plan_args* a, int n_windows, const float* states, const float* actions,
                       const float* rewards, const float* cand, const int* window_index, float* expect_return,
                       float* pred_rewards, float* pred_boot, void* stream) {
    if (!h || !a || !states || !actions || !rewards || !cand || !expect_return) return fail(M3PC_EINVAL, "null argument");
    if (!h->weights_loaded) return fail(M3PC_ESTATE, "weights not loaded");
    for (int k = 0; k < 4; ++k)
        if (!h->tok_set[k]) return fail(M3PC_ESTATE, "tokenizer '%s' not set", KEYN[k]);
    const int T = h->T, n = a->n_count;
    if (a->horizon < 1 || a->horizon > T) return fail(M3PC_EINVAL, "horizon %d outside [1, T=%d]", a->horizon, T);
    if (a->mode != M3PC_MODE_RTG && a->mode != M3PC_MODE_CRITIC) return fail(M3PC_EINVAL, "mode must be RTG or CRITIC scoring");
    if (a->precision != M3PC_PREC_FP32 && a->precision != M3PC_PREC_BF16) return fail(M3PC_EINVAL, "bad precision");
    if (n < 1 || (n > h->dm.max_candidates && !(a->precision == M3PC_PREC_FP32 && n <= h->chain[0].max_cand)))
        return fail(M3PC_ENOMEM, "n_count %d outside [1, max_candidates=%d]", n, h->dm.max_candidates);
    if (n_windows < 1 || (n_windows > 1 && !window_index)) return fail(M3PC_EINVAL, "n_windows > 1 needs window_index");
    if (a->mode == M3PC_MODE_CRITIC && !h->critic_set) return fail(M3PC_ESTATE, "critic weights not set");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(h->device));
    const int dt = a->precision == M3PC_PREC_BF16 ? DT_BF16 : DT_F32;
    // few-row fp32 calls (the re-score of a batched plan) run in the chain workspace, like m3pc_rescore
    WsScope ws(h, dt == DT_F32 && n <= h->chain[0].max_cand, false, a->slot);
    if (h->cur == &h->base) CHK(ws_sync(h, st));
    SampleP sp;
    memset(&sp, 0, sizeof(sp));
    sp.hist_actions = actions;
    sp.eps = cand;
    sp.mode = 2;
    sp.T = T;
    sp.A = h->A;
    sp.idx = T - a->horizon;
    sp.h = a->horizon;
    sp.n_count = n;
    sp.widx = window_index;
    sp.cand = h->cand;
    launch_sample(sp, st);
    // (split-K is row-count dependent: only where the caller does not rely on sharding exactness, i.e. the fp32 re-scores)
    h->allow_splitk = dt == DT_F32;
    const int rc = candidate_pass(h, a, states, rewards, n, cand, expect_return, pred_rewards, pred_boot, dt, st, window_index);
    h->allow_splitk = false;
    return rc;
}

// PASS 1 of E windows at once (learner.py:278-284 per window): return-conditioned policy, batch E, rcbc mask, fp32, in the
// policy workspace; the slot then holds E policy heads (loc / sd rows [w T, (w+1) T)) and E rows of returns tokens.
int m3pc_policy_pass_batch(m3pc_handle* h, const m3pc_plan_args* a, int n_windows, const float* states, const float* actions,
                           const float* rewards, const double* rtg, float* loc, float* std_, void* stream) {
    if (!h || !a || !states || !actions || !rewards || !rtg) return fail(M3PC_EINVAL, "null argument");
    if (!h->weights_loaded) return fail(M3PC_ESTATE, "weights not loaded");
    for (int k = 0; k < 4; ++k)
        if (!h->tok_set[k]) return fail(M3PC_ESTATE, "tokenizer '%s' not set", KEYN[k]);
    const int T = h->T, E = n_windows;
    if (E < 1 || E > h->dm.max_batch) return fail(M3PC_ENOMEM, "n_windows %d outside [1, max_batch=%d]", E, h->dm.max_batch);
    if (a->horizon < 1 || a->horizon > T) return fail(M3PC_EINVAL, "horizon %d outside [1, T=%d]", a->horizon, T);
    if (a->mode < 0 || a->mode > 2) return fail(M3PC_EINVAL, "bad mode %d", a->mode);
    if (a->slot < 0 || a->slot >= M3PC_SLOTS) return fail(M3PC_EINVAL, "slot %d outside [0, %d)", a->slot, M3PC_SLOTS);
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(h->device));
    const int idx = T - a->horizon, A = h->A;
    bind_slot(h, a->slot);
    CHK(fill_rtok(h, rtg, E, st));
    Plan* pl = nullptr;
    CHK(get_mask_plan(h, 0, idx, &pl));
    TokIn in;
    memset(&in, 0, sizeof(in));
    in.ptr[M3PC_STATES] = states;
    in.bstride[M3PC_STATES] = (long long)T * h->S;
    in.normalize[M3PC_STATES] = h->tok_norm[M3PC_STATES];
    in.ptr[M3PC_ACTIONS] = actions;
    in.bstride[M3PC_ACTIONS] = (long long)T * A;
    in.normalize[M3PC_ACTIONS] = h->tok_norm[M3PC_ACTIONS];
    in.ptr[M3PC_REWARDS] = rewards;
    in.bstride[M3PC_REWARDS] = T;
    in.normalize[M3PC_REWARDS] = h->tok_norm[M3PC_REWARDS];
    in.ptr[M3PC_RETURNS] = h->rtok;
    in.bstride[M3PC_RETURNS] = T;
    h->allow_splitk = true;
    {
        WsScope ws(h, true, true, a->slot);
        const int rc = forward_impl(h, pl, in, E, nullptr, nullptr, nullptr, h->loc, h->sd, DT_F32, st);
        h->allow_splitk = false;
        if (rc) return rc;
    }
    h->slot[a->slot].policy_valid = E == 1;  // (m3pc_rescore works on a single-window slot; batched callers re-score with m3pc_score_actions)
    h->slot[a->slot].n_windows = E;
    if (loc) HIPCHK(hipMemcpyAsync(loc, h->loc, (size_t)E * T * A * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (std_) HIPCHK(hipMemcpyAsync(std_, h->sd, (size_t)E * T * A * sizeof(float), hipMemcpyDeviceToDevice, st));
    return check_launch("policy_pass_batch");
}

int m3pc_plan_step_batch(m3pc_handle* h, const m3pc_plan_args* a, int n_windows, const float* states, const float* actions,
                         const float* rewards, const double* rtg, const float* eps, const int* window_index, float* loc,
                         float* std_, float* sample_actions, float* expect_return, void* stream) {
    if (!h || !a || !states || !actions || !rewards || !rtg || !eps || !window_index || !sample_actions || !expect_return)
        return fail(M3PC_EINVAL, "null argument");
    const int T = h->T, E = n_windows, N = a->n_total;
    if (a->precision != M3PC_PREC_FP32 && a->precision != M3PC_PREC_BF16) return fail(M3PC_EINVAL, "bad precision");
    if (E >= 1 && (N < 1 || (long long)E * N > h->dm.max_candidates))
        return fail(M3PC_ENOMEM, "n_windows * n_total = %lld > max_candidates %d", (long long)E * N, h->dm.max_candidates);
    if (a->mode >= 0 && a->mode <= 2 && a->mode != M3PC_MODE_RTG && !h->critic_set) return fail(M3PC_ESTATE, "critic weights not set");
    CHK(m3pc_policy_pass_batch(h, a, n_windows, states, actions, rewards, rtg, nullptr, nullptr, stream));
    hipStream_t st = (hipStream_t)stream;
    const int hh = a->horizon, idx = T - hh, A = h->A;
    h->slot[a->slot].policy_valid = false;
    CHK(ws_sync(h, st));
    // candidates of window w: rows [w N, (w+1) N) of cand / sample_actions, drawn from window w's policy head and eps block
    for (int w = 0; w < E; ++w) {
        SampleP sp;
        memset(&sp, 0, sizeof(sp));
        sp.hist_actions = actions + (size_t)w * T * A;
        sp.loc = h->loc + (size_t)w * T * A;
        sp.sd = h->sd + (size_t)w * T * A;
        const size_t per = a->mode == M3PC_MODE_NOISE ? (size_t)hh * A : (size_t)T * A;
        sp.eps = eps + (size_t)w * N * per;
        sp.mode = a->mode == M3PC_MODE_NOISE ? 1 : 0;
        sp.T = T;
        sp.A = A;
        sp.idx = idx;
        sp.h = hh;
        sp.n_count = N;
        sp.cand = h->cand + (size_t)w * N * T * A;
        sp.sample_actions = sample_actions + (size_t)w * N * hh * A;
        sp.loc_out = loc ? loc + (size_t)w * T * A : nullptr;
        sp.sd_out = std_ ? std_ + (size_t)w * T * A : nullptr;
        launch_sample(sp, st);
    }
    const int dt = a->precision == M3PC_PREC_BF16 ? DT_BF16 : DT_F32;
    return candidate_pass(h, a, states, rewards, E * N, sample_actions, expect_return, nullptr, nullptr, dt, st, window_index);
}

int m3pc_rescore(m3pc_handle* h, const m3pc_plan_args* a, const float* states, const float* actions, const float* rewards,
                 const float* eps, const int* index, int n, float* sample_actions, float* expect_return, void* stream) {
    if (!h || !a || !states || !actions || !rewards || !eps || !index || !expect_return) return fail(M3PC_EINVAL, "null argument");
    if (!h->weights_loaded) return fail(M3PC_ESTATE, "weights not loaded");
    if (a->slot < 0 || a->slot >= M3PC_SLOTS) return fail(M3PC_EINVAL, "slot %d outside [0, %d)", a->slot, M3PC_SLOTS);
    if (!h->slot[a->slot].policy_valid) return fail(M3PC_ESTATE, "m3pc_rescore needs a preceding m3pc_plan_step / m3pc_policy_pass on slot %d", a->slot);
    const int T = h->T;
    if (a->horizon < 1 || a->horizon > T || a->mode < 0 || a->mode > 2) return fail(M3PC_EINVAL, "bad horizon/mode");
    // runs in the chain workspace when it fits (so that it can be enqueued beside a candidate pass), else in the candidate one
    const bool in_chain = n <= h->chain[0].max_cand;
    if (n < 1 || (!in_chain && n > h->dm.max_candidates))
        return fail(M3PC_ENOMEM, "n %d outside [1, max(max_rescore=%d, max_candidates=%d)]", n, h->chain[0].max_cand, h->dm.max_candidates);
    if (a->mode != M3PC_MODE_RTG && !h->critic_set) return fail(M3PC_ESTATE, "critic weights not set");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(h->device));
    bind_slot(h, a->slot);
    WsScope ws(h, in_chain, false, a->slot);
    if (!in_chain) CHK(ws_sync(h, st));
    float* sa = sample_actions ? sample_actions : (in_chain ? h->sa_chain[a->slot & 1] : h->sa_buf);
    SampleP sp;
    memset(&sp, 0, sizeof(sp));
    sp.hist_actions = actions;
    sp.loc = h->loc;
    sp.sd = h->sd;
    sp.eps = eps;
    sp.mode = a->mode == M3PC_MODE_NOISE ? 1 : 0;
    sp.T = T;
    sp.A = h->A;
    sp.idx = T - a->horizon;
    sp.h = a->horizon;
    sp.n_count = n;
    sp.index = index;
    sp.cand = h->cand;
    sp.sample_actions = sa;
    launch_sample(sp, st);
    h->allow_splitk = true;
    const int rc = candidate_pass(h, a, states, rewards, n, sa, expect_return, nullptr, nullptr, DT_F32, st);
    h->allow_splitk = false;
    return rc;
}

int m3pc_rescore_topk(m3pc_handle* h, const m3pc_plan_args* a, const float* states, const float* actions,
                      const float* rewards, const float* eps, float* expect_return, int k, int* topk_index, void* stream) {
    if (!h || !a || !expect_return) return fail(M3PC_EINVAL, "null argument");
    if (a->n_total < 1 || a->n_total > 16384) return fail(M3PC_EINVAL, "top-k supports n_total <= 16384");
    if (k < 1 || k > a->n_total || (k > h->dm.max_candidates && k > h->chain[0].max_cand) || k > 1024)
        return fail(M3PC_EINVAL, "k %d outside [1, min(n_total, max(max_candidates, max_rescore), 1024)]", k);
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(h->device));
    launch_topk(expect_return, a->n_total, k, h->d_topk, st);
    CHK(m3pc_rescore(h, a, states, actions, rewards, eps, h->d_topk, k, nullptr, h->er_top, stream));
    launch_scatter(h->er_top, h->d_topk, k, expect_return, topk_index, st);
    return check_launch("rescore_topk");
}

int m3pc_topk_window(m3pc_handle* h, const float* expect_return, int n_total, int kmax, int kmin, float window, int* topk_index,
                     float* stats, float* top_scores, float* host_stats, float seq, void* stream) {
    if (!h || !expect_return || !topk_index || !stats) return fail(M3PC_EINVAL, "null argument");
    if (n_total < 1 || n_total > 16384) return fail(M3PC_EINVAL, "top-k supports n_total <= 16384");
    if (kmax < 1 || kmax > 1023 || kmin < 1 || kmin > kmax || !(window >= 0.f)) return fail(M3PC_EINVAL, "bad kmin/kmax/window");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(h->device));
    const int kk = kmax + 1 < n_total ? kmax + 1 : n_total;
    launch_topk(expect_return, n_total, kk, topk_index, st);
    launch_window_stats(expect_return, n_total, topk_index, kk, kmin, kmax, window, stats, host_stats, seq, top_scores, st);
    return check_launch("topk_window");
}

int m3pc_topk_race_window(m3pc_handle* h, const float* expect_return, const float* expo, float temperature, int n_total, int kmax,
                          int kmin, int rmax, int* list, float* stats, float* list_scores, float* host_stats, float seq, void* stream) {
    if (!h || !expect_return || !expo || !list || !stats) return fail(M3PC_EINVAL, "null argument");
    if (n_total < 1 || n_total > 16384) return fail(M3PC_EINVAL, "top-k supports n_total <= 16384");
    if (kmax < 1 || kmax > 1023 || kmin < 1 || kmin > kmax) return fail(M3PC_EINVAL, "bad kmin/kmax");
    if (rmax < 1 || rmax > 64 || rmax > n_total) return fail(M3PC_EINVAL, "rmax %d outside [1, min(64, n_total)]", rmax);
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(h->device));
    const int kk = kmax + 1 < n_total ? kmax + 1 : n_total;
    launch_topk_race(expect_return, expo, temperature, n_total, kk, rmax, rmax, list, st);
    launch_window_stats(expect_return, n_total, list + rmax, kk, kmin, kmax, 0.f, stats, host_stats, seq,
                        list_scores ? list_scores + rmax : nullptr, st, rmax);
    return check_launch("topk_race_window");
}

int m3pc_rescore_merge(m3pc_handle* h, const float* scores, int n_total, const int* index, int n, const float* top_scores,
                       const float* top_rescored, float delta, float* merged, float* stats, float* host_stats, float seq,
                       void* stream) {
    if (!h || !scores || !index || !top_scores || !top_rescored || !merged || !stats) return fail(M3PC_EINVAL, "null argument");
    if (n_total < 1 || n < 1 || n > 1024 || n > n_total) return fail(M3PC_EINVAL, "n %d outside [1, min(1024, n_total)]", n);
    if (!(delta >= 0.f)) return fail(M3PC_EINVAL, "delta must be >= 0");
    HIPCHK(hipSetDevice(h->device));
    launch_rescore_merge(scores, n_total, index, 0, n, top_scores, top_rescored, delta, nullptr, 0.f, merged, stats, host_stats, seq,
                         (hipStream_t)stream);
    return check_launch("rescore_merge");
}

int m3pc_rescore_merge_race(m3pc_handle* h, const float* scores, const float* expo, float temperature, int n_total, const int* list,
                            int r, int n, const float* list_scores, const float* list_rescored, float delta, float* merged,
                            float* stats, float* host_stats, float seq, void* stream) {
    if (!h || !scores || !expo || !list || !list_scores || !list_rescored || !merged || !stats) return fail(M3PC_EINVAL, "null argument");
    if (n_total < 1 || n < 1 || r < 0 || r + n > 1024 || n > n_total || r > n_total)
        return fail(M3PC_EINVAL, "r %d + n %d outside [1, 1024] / n_total %d", r, n, n_total);
    if (!(delta >= 0.f)) return fail(M3PC_EINVAL, "delta must be >= 0");
    HIPCHK(hipSetDevice(h->device));
    launch_rescore_merge(scores, n_total, list, r, n, list_scores, list_rescored, delta, expo, temperature, merged, stats, host_stats,
                         seq, (hipStream_t)stream);
    return check_launch("rescore_merge_race");
}

int m3pc_rescore_listed(m3pc_handle* h, const m3pc_plan_args* a, const float* states, const float* actions, const float* rewards,
                        const float* eps, const int* index, int n, float* expect_return, void* stream) {
    if (!h || !a || !expect_return || !index) return fail(M3PC_EINVAL, "null argument");
    if (n < 1 || n > 1024) return fail(M3PC_EINVAL, "n %d outside [1, 1024]", n);
    h->score_scatter_index = index;
    h->score_scatter_out = expect_return;
    const int rc = m3pc_rescore(h, a, states, actions, rewards, eps, index, n, nullptr, h->er_top, stream);
    h->score_scatter_index = nullptr;
    h->score_scatter_out = nullptr;
    if (rc) return rc;
    return check_launch("rescore_listed");
}

int m3pc_select(m3pc_handle* h, const float* expect_return, const float* a0, long long a0_stride, int n, float temperature,
                const float* expo, float* p, float* eval_action, int* argmax, int* sample_idx, float* sample_action,
                void* stream) {
    if (!h || !expect_return || n < 1) return fail(M3PC_EINVAL, "bad argument");
    if ((eval_action || sample_action) && !a0) return fail(M3PC_EINVAL, "eval_action / sample_action need a0");
    if ((sample_idx || sample_action) && !expo) return fail(M3PC_EINVAL, "the multinomial draw needs expo");
    HIPCHK(hipSetDevice(h->device));
    SelectP s;
    memset(&s, 0, sizeof(s));
    s.er = expect_return;
    s.a0 = a0;
    s.a0_stride = a0_stride;
    s.n = n;
    s.A = h->A;
    s.temperature = temperature;
    s.expo = expo;
    s.p = p;
    s.eval_action = eval_action;
    s.argmax = argmax;
    s.sample_idx = sample_idx;
    s.sample_action = sample_action;
    launch_select(s, (hipStream_t)stream);
    return check_launch("select");
}

#ifdef M3PC_LAB  // ---- kernel-level test / bench hooks: libm3pc_hip_lab.so only, declared in include/m3pc_hip_debug.h
// Lab build only (include/m3pc_hip_debug.h): lets tools/gemm_bench.py time the GEMM kernel on the plan step's shapes.
int m3pc_debug_gemm(int dtype, const void* A, const void* Wt, const float* bias, const float* res, void* C, int M, int N,
                    int K, int gelu, int f32out, int variant, void* stream) {
    static const int ldpad = M3PC_ENV("M3PC_DEBUG_LDPAD") ? atoi(M3PC_ENV("M3PC_DEBUG_LDPAD")) : 0;  // operand row padding (elements)
    GemmP p = gemm_basic(A, K + ldpad, Wt, K + ldpad, M, N, K, bias);
    p.a_padded = 0;  // (a caller's tensor: nothing is known about the memory behind it)
    p.gelu = gelu;
    p.res = res;
    p.ldr = N;
    p.variant = variant;
    if (dtype == DT_BF16 && !f32out)
        p.Cb = (bf16_t*)C;
    else
        p.Cf = (float*)C;
    p.ldc = N;
    if (dtype == DT_F32 && variant != 1) {  // split-K workspace as the handle provides it (variant 1: none)
        static float* ws = nullptr;
        if (!ws) HIPCHK(hipMalloc((void**)&ws, 64 << 20));
        p.ws = ws;
        p.ws_bytes = 64 << 20;
    }
    launch_gemm(p, dtype, (hipStream_t)stream);
    return check_launch("debug_gemm");
}

// Not part of the public header (tools/gemm_bench.py): clock counters of the last probed GEMM workgroup.
// cap > 0: from now on every fused-tail launch of the handle logs the phase stamps of its workgroup 37 into a ring of `cap`
// entries (64 int64 each: wave w at [16 w ..]); cap == 0: copy the ring to out (host, cap_prev * 64 int64), return how many
// launches were logged through *n_logged, and stop logging
int m3pc_debug_stamp_log(m3pc_handle* h, int cap, long long* out, int* n_logged) {
    if (!h) return fail(M3PC_EINVAL, "null handle");
    if (cap > 0) {
        if (h->stamp_log) hipFree(h->stamp_log);
        CHK(dmalloc(&h->stamp_log, (size_t)cap * 64));
        HIPCHK(hipMemset(h->stamp_log, 0, (size_t)cap * 64 * sizeof(long long)));
        h->stamp_cap = cap;
        h->stamp_i = 0;
        return 0;
    }
    if (!h->stamp_log) return fail(M3PC_ESTATE, "stamp log not enabled");
    HIPCHK(hipDeviceSynchronize());
    if (out) HIPCHK(hipMemcpy(out, h->stamp_log, (size_t)h->stamp_cap * 64 * sizeof(long long), hipMemcpyDeviceToHost));
    if (n_logged) *n_logged = h->stamp_i;
    hipFree(h->stamp_log);
    h->stamp_log = nullptr;
    h->stamp_cap = 0;
    return 0;
}

int m3pc_debug_clock(long long* out2) {
    HIPCHK(hipDeviceSynchronize());
    read_clock_probe(out2);
    return 0;
}

// Not part of the public header (tests/test_gemm_kernels_gpu.py): the top-k kernels on their own.
int m3pc_debug_topk(const float* v, int n, int k, int* idx_out, void* stream) {
    launch_topk(v, n, k, idx_out, (hipStream_t)stream);
    return check_launch("debug_topk");
}

// Not part of the public header (tests/test_block_fused_gpu.py, tools/block_bench.py): the fused layer tail on its own.
//   O (M,512) bf16; res (M,512) fp32 or rowtab (rt_mod,512); Wo (512,512), W1 (2048,512), W2 (512,2048) bf16 in torch
//   Linear layout; stream: scratch of m3pc_debug_block_stream_bytes() bytes (packed when pack != 0);
//   lnB_g0 / lnB_g1 optional (with out_mod / out_grp); Xout (M,512) fp32 optional; Hout (M,512) bf16 optional
long long m3pc_debug_block_stream_bytes(void) { return (long long)block_stream_bytes(); }
int m3pc_debug_block_fused(const void* O, int M, const float* res, const float* rowtab, int rt_mod, const void* Wo, const void* W1,
                           const void* W2, void* stream_buf, int pack, const float* bo, const float* b1, const float* b2,
                           const float* ln2_g, const float* ln2_b, const float* lnA_g, const float* lnA_b, const float* lnB_g0,
                           const float* lnB_b0, const float* lnB_g1, const float* lnB_b1, int out_mod, int out_grp, float* Xout,
                           void* Hout, int variant, void* stream, long long* stamps) {
    hipStream_t st = (hipStream_t)stream;
    if (pack) launch_pack_block_stream((const bf16_t*)Wo, (const bf16_t*)W1, (const bf16_t*)W2, (bf16_t*)stream_buf, st);
    BlockP b;
    memset(&b, 0, sizeof(b));
    b.O = (const bf16_t*)O;
    b.ldo = 512;
    b.M = M;
    b.res = res;
    b.ldr = 512;
    b.rowtab = rowtab;
    b.rt_mod = rt_mod;
    b.wstream = (const bf16_t*)stream_buf;
    b.bo = bo;
    b.b1 = b1;
    b.b2 = b2;
    b.ln2_g = ln2_g;
    b.ln2_b = ln2_b;
    b.Xout = Xout;
    b.ldx = 512;
    b.lnA_g = lnA_g;
    b.lnA_b = lnA_b;
    b.lnB_g[0] = lnB_g0;
    b.lnB_b[0] = lnB_b0;
    b.lnB_g[1] = lnB_g1;
    b.lnB_b[1] = lnB_b1;
    b.out_mod = out_mod;
    b.out_grp = out_grp;
    b.Hout = (bf16_t*)Hout;
    b.ldh = 512;
    b.variant = variant;
    b.stamps = stamps;
    if (!launch_block_fused(b, st)) return fail(M3PC_EINVAL, "block_fused: arguments not covered");
    return check_launch("debug_block_fused");
}

// the fused layer tail with the next layer's Q|K|V projection behind it: QKV (M, 1536) bf16 = LN_A(X'') Wqkv^T + bqkv
int m3pc_debug_block_fused_qkv(const void* O, int M, const float* res, const void* Wo, const void* W1, const void* W2, const void* Wqkv,
                               void* stream_buf, const float* bo, const float* b1, const float* b2, const float* ln2_g,
                               const float* ln2_b, const float* lnA_g, const float* lnA_b, const float* bqkv, float* Xout, void* QKV,
                               void* stream, long long* stamps) {
    hipStream_t st = (hipStream_t)stream;
    launch_pack_block_stream((const bf16_t*)Wo, (const bf16_t*)W1, (const bf16_t*)W2, (bf16_t*)stream_buf, st);
    launch_pack_block_qkv((const bf16_t*)Wqkv, (bf16_t*)stream_buf, st);
    BlockP b;
    memset(&b, 0, sizeof(b));
    b.O = (const bf16_t*)O;
    b.ldo = 512;
    b.M = M;
    b.res = res;
    b.ldr = 512;
    b.wstream = (const bf16_t*)stream_buf;
    b.bo = bo;
    b.b1 = b1;
    b.b2 = b2;
    b.ln2_g = ln2_g;
    b.ln2_b = ln2_b;
    b.Xout = Xout;
    b.ldx = 512;
    b.lnA_g = lnA_g;
    b.lnA_b = lnA_b;
    b.QKVout = (bf16_t*)QKV;
    b.ldq = 1536;
    b.qkv_bytes = (unsigned)((size_t)M * 1536 * 2);
    b.bqkv = bqkv;
    b.stamps = stamps;
    if (!launch_block_fused(b, st)) return fail(M3PC_EINVAL, "block_fused (qkv): arguments not covered");
    return check_launch("debug_block_fused_qkv");
}

// the decoder form of the fused tail with the two scalar output heads inside: rows of group s = (r % out_mod) / out_grp;
// out0 / out1 (M / 2) floats; Wh (2, 512, 512) bf16, hb1 / hw2 (2, 512), hb2 / hmean / hstd (2) floats (hmean null: no detok)
int m3pc_debug_block_fused_heads(const void* O, int M, const float* rowtab, int rt_mod, const void* Wo, const void* W1, const void* W2,
                                 const void* Wh, void* stream_buf, const float* bo, const float* b1, const float* b2, const float* ln2_g,
                                 const float* ln2_b, const float* lnA_g, const float* lnA_b, const float* lnB_g0, const float* lnB_b0,
                                 const float* lnB_g1, const float* lnB_b1, int out_mod, int out_grp, const float* hb1, const float* hw2,
                                 const float* hb2, const float* hmean, const float* hstd, float* out0, float* out1, void* stream,
                                 long long* stamps) {
    hipStream_t st = (hipStream_t)stream;
    launch_pack_block_stream((const bf16_t*)Wo, (const bf16_t*)W1, (const bf16_t*)W2, (bf16_t*)stream_buf, st);
    launch_pack_block_heads((const bf16_t*)Wh, (const bf16_t*)Wh + 512 * 512, (bf16_t*)stream_buf, st);
    BlockP b;
    memset(&b, 0, sizeof(b));
    b.O = (const bf16_t*)O;
    b.ldo = 512;
    b.M = M;
    b.rowtab = rowtab;
    b.rt_mod = rt_mod;
    b.wstream = (const bf16_t*)stream_buf;
    b.bo = bo;
    b.b1 = b1;
    b.b2 = b2;
    b.ln2_g = ln2_g;
    b.ln2_b = ln2_b;
    b.lnA_g = lnA_g;
    b.lnA_b = lnA_b;
    b.lnB_g[0] = lnB_g0;
    b.lnB_b[0] = lnB_b0;
    b.lnB_g[1] = lnB_g1;
    b.lnB_b[1] = lnB_b1;
    b.out_mod = out_mod;
    b.out_grp = out_grp;
    b.head_out[0] = out0;
    b.head_out[1] = out1;
    for (int s = 0; s < 2; ++s) {
        b.hb1[s] = hb1 + 512 * s;
        b.hw2[s] = hw2 + 512 * s;
        b.hb2[s] = hb2 + s;
        b.hmean[s] = hmean ? hmean + s : nullptr;
        b.hstd[s] = hstd ? hstd + s : nullptr;
    }
    b.stamps = stamps;
    if (!launch_block_fused(b, st)) return fail(M3PC_EINVAL, "block_fused (heads): arguments not covered");
    return check_launch("debug_block_fused_heads");
}

// kv_fused_kernel alone (tests/test_block_fused_gpu.py): n candidates of Le rows each in Z (n*Le, 512) bf16; group g holds
// the kept[g] rows at offset off[g] of every candidate, embedded with We[g] (512, 512) bf16 + rowtab[g] (kept[g], 512);
// stream_buf: 2 * m3pc_debug_kv_stream_bytes() bytes; KV (n*Le, 1024) bf16
long long m3pc_debug_kv_stream_bytes(void) { return (long long)kv_stream_bytes(); }
int m3pc_debug_attention_bf16(const void* QKV, const void* QKVs, void* O, int batch, int n_own, int n_sh, int kernel, void* stream,
                              long long* stamps) {
    const int d = 512, L = n_own + n_sh;
    AttnP a;
    memset(&a, 0, sizeof(a));
    const char* q = (const char*)QKV;
    const char* qs = (const char*)QKVs;
    a.Q = q;
    a.q_bstride = (long long)n_own * 3 * d;
    a.ldq = 3 * d;
    a.Lq = n_own;
    a.K1 = q + (size_t)d * 2;
    a.V1 = q + (size_t)2 * d * 2;
    a.kv1_bstride = (long long)n_own * 3 * d;
    a.ldkv1 = 3 * d;
    a.L1 = n_own;
    if (n_sh > 0) {  // (run_block: own rows first in the slots, shared rows first in the output)
        a.orow1 = n_sh;
        a.Q2 = qs;
        a.ldq2 = 3 * d;
        a.Lq2 = n_sh;
        a.orow2 = 0;
        a.K2 = qs + (size_t)d * 2;
        a.V2 = qs + (size_t)2 * d * 2;
        a.ldkv2 = 3 * d;
        a.L2 = n_sh;
    }
    a.O = O;
    a.o_bstride = (long long)L * d;
    a.ldo = d;
    a.batch = batch;
    a.n_head = 4;
    a.hd = 128;
    a.scale = 1.0f / sqrtf(128.0f);
    a.stamps = stamps;
    a.no_pipe = kernel;  // (2, 3: timing variants of the pipelined kernel that compute nothing / load nothing)
    launch_attention(a, DT_BF16, (hipStream_t)stream);
    return check_launch("debug_attention_bf16");
}

int m3pc_debug_attention_dec_bf16(const void* Qtab, const void* QKVm, const void* KV, void* O, float* pre, int n, int nq, int Lm, int kernel,
                                  void* stream) {
    const int d = 512, nh = 4, hd = 128, Le = 49;
    hipStream_t st = (hipStream_t)stream;
    float* pre_m = pre;
    float* pre_l = pre + nh * nq;
    float* pre_O = pre + 2 * nh * nq;
    AttnP at;
    memset(&at, 0, sizeof(at));
    at.Q = Qtab;
    at.ldq = 3 * d;
    at.Lq = nq;
    at.K2 = (const char*)QKVm + (size_t)d * 2;
    at.V2 = (const char*)QKVm + (size_t)2 * d * 2;
    at.ldkv2 = 3 * d;
    at.L2 = Lm;
    at.n_head = nh;
    at.hd = hd;
    at.scale = 1.0f / sqrtf((float)hd);
    launch_attention_prestats(at, pre_m, pre_l, pre_O, st);
    at.K2 = at.V2 = nullptr;
    at.L2 = 0;
    at.q_bstride = 0;
    at.K1 = KV;
    at.V1 = (const char*)KV + (size_t)d * 2;
    at.kv1_bstride = (long long)Le * 2 * d;
    at.ldkv1 = 2 * d;
    at.L1 = Le;
    at.O = O;
    at.o_bstride = (long long)nq * d;
    at.ldo = d;
    at.batch = n;
    at.pre_m = pre_m;
    at.pre_l = pre_l;
    at.pre_O = pre_O;
    at.no_pipe = kernel;
    launch_attention(at, DT_BF16, st);
    return check_launch("debug_attention_dec_bf16");
}

int m3pc_debug_attention_mix_bf16(const void* Qown, const void* Qsh, const void* KV, const void* QKVm, void* O, int n, int Lq, int Lq2, int kernel,
                                  void* stream) {
    const int d = 512, Le = 49, Lm = 79;
    AttnP at;
    memset(&at, 0, sizeof(at));
    at.Q = Qown;
    at.q_bstride = (long long)Lq * d;
    at.ldq = d;
    at.Lq = Lq;
    at.orow1 = 0;
    at.Q2 = Qsh;
    at.ldq2 = 3 * d;
    at.Lq2 = Lq2;
    at.orow2 = Lq;
    at.K1 = KV;
    at.V1 = (const char*)KV + (size_t)d * 2;
    at.kv1_bstride = (long long)Le * 2 * d;
    at.ldkv1 = 2 * d;
    at.L1 = Le;
    at.K2 = (const char*)QKVm + (size_t)d * 2;
    at.V2 = (const char*)QKVm + (size_t)2 * d * 2;
    at.ldkv2 = 3 * d;
    at.L2 = Lm;
    at.O = O;
    at.o_bstride = (long long)(Lq + Lq2) * d;
    at.ldo = d;
    at.batch = n;
    at.n_head = 4;
    at.hd = 128;
    at.scale = 1.0f / sqrtf(128.0f);
    at.no_pipe = kernel;
    launch_attention(at, DT_BF16, (hipStream_t)stream);
    return check_launch("debug_attention_mix_bf16");
}

int m3pc_debug_kv_fused(const void* Z, int n, int Le, int kept0, int off0, int kept1, int off1, const void* We0, const void* We1,
                        const void* Wkv, void* stream_buf, const float* rowtab0, const float* rowtab1, const float* ln_g,
                        const float* ln_b, const float* bkv, void* KV, void* stream, long long* stamps) {
    hipStream_t st = (hipStream_t)stream;
    bf16_t* s0 = (bf16_t*)stream_buf;
    bf16_t* s1 = (bf16_t*)((char*)stream_buf + kv_stream_bytes());
    launch_pack_kv_stream((const bf16_t*)We0, (const bf16_t*)Wkv, s0, st);
    if (kept1) launch_pack_kv_stream((const bf16_t*)We1, (const bf16_t*)Wkv, s1, st);
    KvFusedP p;
    memset(&p, 0, sizeof(p));
    p.Z = (const bf16_t*)Z;
    p.ldz = 512;
    p.M[0] = n * kept0;
    p.map[0] = RowMap{kept0, Le, off0};
    p.rowtab[0] = rowtab0;
    p.rt_mod[0] = kept0;
    p.wstream[0] = s0;
    p.M[1] = n * kept1;
    p.map[1] = RowMap{kept1 ? kept1 : 1, Le, off1};
    p.rowtab[1] = rowtab1;
    p.rt_mod[1] = kept1 ? kept1 : 1;
    p.wstream[1] = s1;
    p.ln_g = ln_g;
    p.ln_b = ln_b;
    p.bkv = bkv;
    p.KV = (bf16_t*)KV;
    p.ldkv = 1024;
    p.kv_bytes = (unsigned)((size_t)n * Le * 1024 * 2);
    p.stamps = stamps;
    if (!launch_kv_fused(p, st)) return fail(M3PC_EINVAL, "kv_fused: arguments not covered");
    return check_launch("debug_kv_fused");
}

// which XCD (and CU) every workgroup of a launch on `stream` lands on: out[2 i] = XCC_ID, out[2 i + 1] = HW_ID register
// (tools/xcd_probe.py: maps the bits of a hipExtStreamCreateWithCUMask mask to XCDs)
__global__ void xcc_probe_kernel(int* out) {
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = (int)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID[3:0]
        out[2 * blockIdx.x + 1] = (int)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); // HW_REG_HW_ID
    }
    // (long enough that the workgroups of the launch spread over everything the stream may use)
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < 1024; ++i) {
        if (__builtin_amdgcn_s_memrealtime() - t0 >= 300) break;
        __builtin_amdgcn_s_sleep(16);
    }
}
int m3pc_debug_xcc_probe(int* out, int n_blocks, void* stream) {
    hipLaunchKernelGGL(xcc_probe_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, out);
    return check_launch("xcc_probe");
}

int m3pc_debug_clock_big(long long* out4) {
    HIPCHK(hipDeviceSynchronize());
    read_big_probe(out4);
    return 0;
}

#endif  // M3PC_LAB

int m3pc_profile_enable(m3pc_handle* h, int enable) {
    if (!h) return fail(M3PC_EINVAL, "null handle");
    h->prof = enable == 1 || enable == 2;
    h->prof_serial = enable == 2 || enable == 3;
    return 0;
}

int m3pc_profile_read(m3pc_handle* h, int precision, long long* launches, double* gemm_ms, double* gemm_flops, int reset) {
    if (!h) return fail(M3PC_EINVAL, "null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    double ms = 0, fl = 0;
    long long cnt = 0;
    for (size_t i = 0; i < h->ev_used; ++i) {
        if (precision == M3PC_PROF_LAYER_TAIL) {
            if (h->ev[i].kind != 1) continue;
        } else if (precision >= 0 && h->ev[i].dt != (precision == M3PC_PREC_BF16 ? DT_BF16 : DT_F32)) {
            continue;
        }
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, h->ev[i].a, h->ev[i].b));
        ms += t;
        fl += h->ev[i].flops;
        ++cnt;
    }
    if (launches) *launches = cnt;
    if (gemm_ms) *gemm_ms = ms;
    if (gemm_flops) *gemm_flops = fl;
    if (reset) h->ev_used = 0;
    return 0;
}

}  // extern "C"
#pragma GCC visibility pop

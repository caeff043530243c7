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
#include "m3pc_internal.h"

namespace m3pc {
thread_local char g_err[512] = "";
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace m3pc

namespace {

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
#ifdef M3PC_LAB  // (tools/cu_mask_probe.py: the candidate passes' second stream confined to a CU mask, comma-separated hex words)
        if (!sa)
            if (const char* e = M3PC_ENV("M3PC_AUX_CU_MASK")) {
                std::vector<uint32_t> words;
                for (const char* q = e; *q;) {
                    words.push_back((uint32_t)strtoul(q, nullptr, 16));
                    while (*q && *q != ',') ++q;
                    if (*q == ',') ++q;
                }
                HIPCHK(hipExtStreamCreateWithCUMask(&sa, (uint32_t)words.size(), words.data()));
            }
#endif
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
        const bool fuse_ln = actor_head_fuses_ln(d, h->A);  // (the head kernel normalises its rows itself: one launch fewer)
        if (!fuse_ln) {
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
        }
        ActorP ac;
        memset(&ac, 0, sizeof(ac));
        ac.X = fuse_ln ? xr : h->G;
        if (fuse_ln) {
            ac.ln_g = W(h, "decoder.norm.weight").f;
            ac.ln_b = W(h, "decoder.norm.bias").f;
        }
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
    if (a->flags & ~(M3PC_PLAN_DEFER_JOIN | M3PC_PLAN_PRUNED_POLICY))  // (also what a caller built against the shorter ABI v2 structure would hand over)
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
    if ((a->flags & M3PC_PLAN_PRUNED_POLICY) && !loc && !std_ && idx > 0) {
        // the policy head at the action tokens idx .. T-1 only (masked under the rcbc mask for idx > 0): query set 4 of the plan
        WsScope ws(h, true, true, a->slot);
        std::vector<int> toks;
        for (int t = idx; t < T; ++t) toks.push_back(M3PC_ACTIONS * T + t);
        rc = build_query_list(h, pl, 4, hh, toks, 1, M3PC_ACTIONS, 0);
        if (!rc) rc = build_tables(h, pl, 4, DT_F32, st);
        if (!rc && !pl->query[4].all_masked) rc = fail(M3PC_EINVAL, "policy pass: an action token at t >= idx is not masked");
        if (!rc) rc = run_encoder(h, pl, in, 1, DT_F32, st, false, 0);
        float* xr = nullptr;
        if (!rc) rc = pruned_decoder(h, pl, pl->query[4], pl->query[4].tab[DT_F32], 1, DT_F32, st, TAIL_X, &xr);
        if (!rc) {  // decoder.norm of the h query rows, then the action head (mtm_model.py:705, 313-321)
            const int d = h->d;
            const bool fuse_ln = actor_head_fuses_ln(d, h->A);
            if (!fuse_ln) {
                LnP ln;
                memset(&ln, 0, sizeof(ln));
                ln.X = xr;
                ln.ldx = d;
                ln.rows = hh;
                ln.d = d;
                ln.g1 = W(h, "decoder.norm.weight").f;
                ln.b1 = W(h, "decoder.norm.bias").f;
                ln.Yf = h->G;
                launch_layernorm(ln, st);
            }
            ActorP ac;
            memset(&ac, 0, sizeof(ac));
            ac.X = fuse_ln ? xr : h->G;
            if (fuse_ln) {
                ac.ln_g = W(h, "decoder.norm.weight").f;
                ac.ln_b = W(h, "decoder.norm.bias").f;
            }
            ac.ldx = d;
            ac.rows = hh;
            ac.d = d;
            ac.A = h->A;
            ac.Wmu = W(h, "output_head_dict.actions.mu.weight").f;
            ac.bmu = W(h, "output_head_dict.actions.mu.bias").f;
            ac.Wls = W(h, "output_head_dict.actions.log_std.weight").f;
            ac.bls = W(h, "output_head_dict.actions.log_std.bias").f;
            ac.mu = h->loc + (size_t)idx * h->A;
            ac.sd = h->sd + (size_t)idx * h->A;
            launch_actor_head(ac, st);
            if (hipMemsetAsync(h->loc, 0, (size_t)idx * h->A * sizeof(float), st) != hipSuccess ||
                hipMemsetAsync(h->sd, 0, (size_t)idx * h->A * sizeof(float), st) != hipSuccess)
                rc = fail(M3PC_EHIP, "hipMemsetAsync failed in the pruned policy pass");
        }
    } else {
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
    if (!h || !expect_return || !expo || !list) return fail(M3PC_EINVAL, "null argument");
    if (n_total < 1 || n_total > 16384) return fail(M3PC_EINVAL, "top-k supports n_total <= 16384");
    if (kmax < 1 || kmax > 1023 || kmin < 1 || kmin > kmax) return fail(M3PC_EINVAL, "bad kmin/kmax");
    if (!stats && host_stats) return fail(M3PC_EINVAL, "host_stats needs stats");
    if (rmax < 1 || rmax > 64 || rmax > n_total) return fail(M3PC_EINVAL, "rmax %d outside [1, min(64, n_total)]", rmax);
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipSetDevice(h->device));
    const int kk = kmax + 1 < n_total ? kmax + 1 : n_total;
    const bool scored = launch_topk_race(expect_return, expo, temperature, n_total, kk, rmax, rmax, list, list_scores, st);
    // the statistics of the score part (and, beyond 2048 candidates, the listed scores) are a launch of their own: skipped when the
    // caller wants neither (the planner's certificate works on the merge's statistics)
    if (stats || (list_scores && !scored)) {
        if (!stats) stats = h->sel_scratch;
        launch_window_stats(expect_return, n_total, list + rmax, kk, kmin, kmax, 0.f, stats, host_stats, seq,
                            list_scores ? list_scores + rmax : nullptr, st, rmax);
    }
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

int m3pc_merge_race_select(m3pc_handle* h, const float* scores, const float* expo, float temperature, int n_total, const int* list,
                           int r, int n, const float* list_scores, const float* list_rescored, float delta, float* merged,
                           float* stats, float* host_stats, float seq, const float* a0, long long a0_stride, float* p,
                           float* eval_action, int* argmax, int* sample_idx, float* sample_action, void* stream) {
    if (!h || !scores || !expo || !list || !list_scores || !list_rescored || !merged || !stats) return fail(M3PC_EINVAL, "null argument");
    if (n_total < 1 || n < 1 || r < 0 || r + n > 1024 || n > n_total || r > n_total)
        return fail(M3PC_EINVAL, "r %d + n %d outside [1, 1024] / n_total %d", r, n, n_total);
    if (!(delta >= 0.f)) return fail(M3PC_EINVAL, "delta must be >= 0");
    if ((eval_action || sample_action) && !a0) return fail(M3PC_EINVAL, "eval_action / sample_action need a0");
    HIPCHK(hipSetDevice(h->device));
    SelectP s;
    memset(&s, 0, sizeof(s));
    s.er = merged;
    s.a0 = a0;
    s.a0_stride = a0_stride;
    s.n = n_total;
    s.A = h->A;
    s.temperature = temperature;
    s.expo = expo;
    s.p = p;
    s.eval_action = eval_action;
    s.argmax = argmax;
    s.sample_idx = sample_idx;
    s.sample_action = sample_action;
    launch_merge_select(scores, n_total, list, r, n, list_scores, list_rescored, delta, expo, temperature, merged, stats, host_stats,
                        seq, s, (hipStream_t)stream);
    return check_launch("merge_race_select");
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
    b.variant = variant & 15;
    b.x_bf16 = (variant & 16) ? 1 : 0;  // (res and Xout are then (M, 512) bf16 rows)
    b.stamps = stamps;
    if (!launch_block_fused(b, st)) return fail(M3PC_EINVAL, "block_fused: arguments not covered");
    return check_launch("debug_block_fused");
}

// the fused layer tail with the next layer's Q|K|V projection behind it: QKV (M, 1536) bf16 = LN_A(X'') Wqkv^T + bqkv
int m3pc_debug_block_fused_qkv(const void* O, int M, const float* res, const void* Wo, const void* W1, const void* W2, const void* Wqkv,
                               void* stream_buf, const float* bo, const float* b1, const float* b2, const float* ln2_g,
                               const float* ln2_b, const float* lnA_g, const float* lnA_b, const float* bqkv, float* Xout, void* QKV,
                               void* stream, long long* stamps, int x_bf16) {
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
    b.x_bf16 = x_bf16 ? 1 : 0;  // (res and Xout are then (M, 512) bf16 rows)
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
    return m3pc_debug_attention_dec_le_bf16(Qtab, QKVm, KV, O, pre, n, nq, Lm, 49, kernel, stream);
}

int m3pc_debug_attention_dec_le_bf16(const void* Qtab, const void* QKVm, const void* KV, void* O, float* pre, int n, int nq, int Lm, int Le, int kernel,
                                     void* stream) {
    const int d = 512, nh = 4, hd = 128;
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

// Mask plans (which tokens a mask keeps, where the decoder takes each row from) and the candidate-independent decoder tables
// of the pruned candidate pass (decoder inputs / K|V / Q of the masked tokens, the pre-reduced softmax block), built once per
// (weights, mask pattern, mode, precision).  Host side of libm3pc_hip.so; see m3pc_internal.h.
#include "m3pc_internal.h"

namespace m3pc {


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


}  // namespace m3pc

// HBM-streaming and small reduction kernels of the plan step (gfx950): token embedding, LayerNorm,
// output heads, candidate sampling, critic, TD(lambda) scoring and the cross-candidate select.
// All of them are bandwidth- or latency-bound: one wave (64 lanes) per row, 64-lane shuffles for the
// reductions, rows laid out so that lanes touch consecutive addresses.
#include "kernels.h"

namespace m3pc {

__device__ __forceinline__ int map_row(const RowMap& m, int r) {
    if (m.rpg == 0) return r;
    return (r / m.rpg) * m.gstride + (r % m.rpg) + m.off;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ------------------------------------------------------------------------------------------ embed
__global__ __launch_bounds__(256) void embed_kernel(EmbedP p) {
    const long long row = blockIdx.x;  // b * L + j
    const int b = (int)(row / p.L), j = (int)(row % p.L);
    const int2 kt = p.tokmap[j];
    const int key = kt.x, t = kt.y;
    const int D = p.feat[key];
    const float* x = p.tok[key] + (long long)b * p.bstride[key] + (p.widx ? (long long)p.widx[b] * p.wstride[key] : 0) + (long long)t * D;
    __shared__ float xs[32];
    if (threadIdx.x < D) {
        float v = x[threadIdx.x];
        if (p.normalize[key]) v = (v - p.mean[key][threadIdx.x]) / p.stdv[key][threadIdx.x];
        xs[threadIdx.x] = v;
    }
    __syncthreads();
    const float* WT = p.WT[key];
    const float* E = p.E[key] + (long long)t * p.d;
    float* out = p.X + row * p.d;
    float vals[4];  // d <= 1024, blockDim = min(d, 256): at most 4 columns per thread (static indices)
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = threadIdx.x + i * blockDim.x;
        vals[i] = 0.f;
        if (c < p.d) {
            float acc = 0.f;
            for (int f = 0; f < D; ++f) acc = fmaf(xs[f], WT[f * p.d + c], acc);
            acc += E[c];
            out[c] = acc;
            vals[i] = acc;
            s += acc;
        }
    }
    if (!p.ln_g) return;
    // fused LayerNorm of the row (norm1 of the first encoder layer): block-wide mean / variance
    __shared__ float red[8];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    s = wave_sum(s);
    if (lane == 0) red[wid] = s;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < nwv; ++w) tot += red[w];
    const float mean = tot / (float)p.d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (threadIdx.x + i * blockDim.x < p.d) {
            const float c = vals[i] - mean;
            q += c * c;
        }
    q = wave_sum(q);
    __syncthreads();
    if (lane == 0) red[wid] = q;
    __syncthreads();
    tot = 0.f;
    for (int w = 0; w < nwv; ++w) tot += red[w];
    const float rstd = rsqrtf(tot / (float)p.d + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = threadIdx.x + i * blockDim.x;
        if (c < p.d) {
            const float y = (vals[i] - mean) * rstd * p.ln_g[c] + p.ln_b[c];
            if (p.Hf) p.Hf[row * p.d + c] = y;
            if (p.Hb) p.Hb[row * p.d + c] = (bf16_t)y;
        }
    }
}
// Wave-per-row variant (d % 256 == 0): one wave owns token j for a chunk of CB batch elements; lane l holds the
// NV = d/256 float4 column groups (i*64 + l)*4.  The row is built, normalised (LayerNorm = two 64-lane butterflies,
// no LDS, no barrier) and stored with 16-byte accesses.  Tokens j < n_indep do not depend on the batch index: the
// row is computed for the first element of the chunk and only re-stored for the others, which turns the kernel
// into a pure HBM writer (the candidate pass shares 33 of its 49 tokens between candidates).
template <int NV>
__global__ __launch_bounds__(256) void embed_rows_kernel(EmbedP p, int CB, int nchunk) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= p.L * nchunk) return;
    const int j = w / nchunk, ch = w % nchunk;
    const int2 kt = p.tokmap[j];
    const int key = kt.x, t = kt.y;
    const int D = p.feat[key], d = p.d;
    const float* WT = p.WT[key];
    const float* E = p.E[key] + (long long)t * d;
    const bool indep = j < p.n_indep;
    const int b0 = ch * CB, b1 = b0 + CB < p.batch ? b0 + CB : p.batch;
    if (indep && p.x_first_only && b0 > 0 && p.Hb && !p.Hf && j < p.n_sh) return;  // (nothing of this token is stored past element 0)
    const int n_own = p.L - p.n_sh;
    float4 ev[NV], gv[NV], bv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        ev[i] = *(const float4*)(E + c);
        if (p.ln_g) {
            gv[i] = *(const float4*)(p.ln_g + c);
            bv[i] = *(const float4*)(p.ln_b + c);
        }
    }
    float4 x[NV], y[NV];
    for (int b = b0; b < b1; ++b) {
        if (b == b0 || !indep) {
            const float* xin = p.tok[key] + (long long)b * p.bstride[key] + (p.widx ? (long long)p.widx[b] * p.wstride[key] : 0) + (long long)t * D;
#pragma unroll
            for (int i = 0; i < NV; ++i) x[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int f = 0; f < D; ++f) {
                float xf = xin[f];
                if (p.normalize[key]) xf = (xf - p.mean[key][f]) / p.stdv[key][f];
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const float4 wv = *(const float4*)(WT + (long long)f * d + (i * 64 + lane) * 4);
                    x[i].x = fmaf(xf, wv.x, x[i].x);
                    x[i].y = fmaf(xf, wv.y, x[i].y);
                    x[i].z = fmaf(xf, wv.z, x[i].z);
                    x[i].w = fmaf(xf, wv.w, x[i].w);
                }
            }
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                x[i].x += ev[i].x;
                x[i].y += ev[i].y;
                x[i].z += ev[i].z;
                x[i].w += ev[i].w;
                s += (x[i].x + x[i].y) + (x[i].z + x[i].w);
            }
            if (p.ln_g) {
                const float mean = wave_sum(s) / (float)d;
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const float c0 = x[i].x - mean, c1 = x[i].y - mean, c2 = x[i].z - mean, c3 = x[i].w - mean;
                    q += (c0 * c0 + c1 * c1) + (c2 * c2 + c3 * c3);
                }
                const float rstd = rsqrtf(wave_sum(q) / (float)d + 1e-5f);
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    y[i].x = (x[i].x - mean) * rstd * gv[i].x + bv[i].x;
                    y[i].y = (x[i].y - mean) * rstd * gv[i].y + bv[i].y;
                    y[i].z = (x[i].z - mean) * rstd * gv[i].z + bv[i].z;
                    y[i].w = (x[i].w - mean) * rstd * gv[i].w + bv[i].w;
                }
            }
        }
        const long long row = (long long)b * p.L + j;
        if (!(indep && p.x_first_only && b > 0)) {
            if (p.Xb) {
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    typedef bf16_t bf16x4_t __attribute__((ext_vector_type(4)));
                    bf16x4_t o;
                    o[0] = (bf16_t)x[i].x;
                    o[1] = (bf16_t)x[i].y;
                    o[2] = (bf16_t)x[i].z;
                    o[3] = (bf16_t)x[i].w;
                    stream_store(o, (bf16x4_t*)(p.Xb + row * d + (i * 64 + lane) * 4));
                }
            } else {
#pragma unroll
                for (int i = 0; i < NV; ++i) *(float4*)(p.X + row * d + (i * 64 + lane) * 4) = x[i];
            }
        }
        if (!p.ln_g) continue;
        if (p.Hf) {
#pragma unroll
            for (int i = 0; i < NV; ++i) *(float4*)(p.Hf + row * d + (i * 64 + lane) * 4) = y[i];
        }
        if (p.Hb) {
            bf16_t* dst;
            if (p.n_sh == 0)
                dst = p.Hb + row * d;
            else if (j >= p.n_sh)
                dst = p.Hb + ((long long)b * n_own + (j - p.n_sh)) * d;
            else if (b == 0)
                dst = p.Hb_sh + (long long)j * d;
            else
                continue;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                typedef bf16_t bf16x4_t __attribute__((ext_vector_type(4)));
                bf16x4_t o;
                o[0] = (bf16_t)y[i].x;
                o[1] = (bf16_t)y[i].y;
                o[2] = (bf16_t)y[i].z;
                o[3] = (bf16_t)y[i].w;
                stream_store(o, (bf16x4_t*)(dst + (i * 64 + lane) * 4));
            }
        }
    }
}

void launch_embed(const EmbedP& p, hipStream_t st) {
    const long long rows = (long long)p.batch * p.L;
    if (rows <= 0) return;
    if (p.d % 256 == 0 && p.d <= 1024) {
        int CB = p.batch / 64;
        CB = CB < 1 ? 1 : (CB > 16 ? 16 : CB);
        // a long chunk pays where tokens are shared by the batch (computed once per wave); without shared tokens (zero-shot
        // windows, batched window passes) more, shorter waves keep more stores in flight: 155 -> see DESIGN.md section 10
        // (round 6: 1 / 2 / 8 elements per wave instead of 4 move config 5 by nothing: 4.00-4.12 ms every way)
        if (p.n_indep == 0 && CB > 4) CB = 4;
        const int nchunk = (p.batch + CB - 1) / CB;
        const int waves = p.L * nchunk;
        const dim3 grid((waves + 3) / 4), block(256);
        switch (p.d / 256) {
            case 1: hipLaunchKernelGGL(embed_rows_kernel<1>, grid, block, 0, st, p, CB, nchunk); break;
            case 2: hipLaunchKernelGGL(embed_rows_kernel<2>, grid, block, 0, st, p, CB, nchunk); break;
            case 3: hipLaunchKernelGGL(embed_rows_kernel<3>, grid, block, 0, st, p, CB, nchunk); break;
            default: hipLaunchKernelGGL(embed_rows_kernel<4>, grid, block, 0, st, p, CB, nchunk); break;
        }
        return;
    }
    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)rows), dim3(p.d < 256 ? p.d : 256), 0, st, p);
}

// ------------------------------------------------------------------------------------------ gather
__global__ __launch_bounds__(256) void gather_rows_kernel(GatherP p) {
    const long long row = blockIdx.x;
    const int b = (int)(row / p.rows_per_batch), i = (int)(row % p.rows_per_batch);
    const int s = p.rowsrc[i];
    const float* src = s >= 0 ? p.Xe + (long long)b * p.xe_bstride + (long long)s * p.d : p.table + (long long)(-s - 1) * p.d;
    for (int c = threadIdx.x * 4; c < p.d; c += blockDim.x * 4) {
        const float4 v = *(const float4*)(src + c);
        if (p.out) *(float4*)(p.out + row * p.d + c) = v;
        if (p.outb) {
            bf16_t* ob = p.outb + row * p.d + c;
            ob[0] = (bf16_t)v.x;
            ob[1] = (bf16_t)v.y;
            ob[2] = (bf16_t)v.z;
            ob[3] = (bf16_t)v.w;
        }
    }
}
void launch_gather_rows(const GatherP& p, hipStream_t st) {
    const long long rows = (long long)p.batch * p.rows_per_batch;
    if (rows <= 0) return;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)rows), dim3(p.d / 4 < 256 ? p.d / 4 : 256), 0, st, p);
}

// ------------------------------------------------------------------------------------------ layernorm
// one wave per row, d/64 (<= 16) values per lane kept in registers
__global__ __launch_bounds__(256) void layernorm_kernel(LnP p) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= p.rows) return;
    const float* x = p.X + (long long)map_row(p.xmap, r) * p.ldx;
    const int n = p.d >> 6;
    float v[16];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < n) {
            v[i] = x[i * 64 + lane];
            s += v[i];
        }
    const float inv_d = 1.0f / (float)p.d;
    float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < n) {
            const float c = v[i] - mean;
            q += c * c;
        }
    float rstd = rsqrtf(wave_sum(q) * inv_d + 1e-5f);
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < n) v[i] = (v[i] - mean) * rstd * p.g1[i * 64 + lane] + p.b1[i * 64 + lane];
    if (p.g2) {
        s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < n) s += v[i];
        mean = wave_sum(s) * inv_d;
        q = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < n) {
                const float c = v[i] - mean;
                q += c * c;
            }
        rstd = rsqrtf(wave_sum(q) * inv_d + 1e-5f);
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < n) v[i] = (v[i] - mean) * rstd * p.g2[i * 64 + lane] + p.b2[i * 64 + lane];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < n) {
            if (p.Yf) p.Yf[(long long)r * p.d + i * 64 + lane] = v[i];
            if (p.Yb) p.Yb[(long long)r * p.d + i * 64 + lane] = (bf16_t)v[i];
        }
}
// d % 256 == 0: 16-byte loads (lane owns 4 consecutive columns per 256-column slab), 8-byte bf16 stores
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void layernorm_vec_kernel(LnP p) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= p.rows) return;
    const float* x = p.X + (long long)map_row(p.xmap, r) * p.ldx;
    const int n = p.d >> 8;  // slabs of 256 columns (<= 4)
    f32x4v v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < n) {
            v[i] = *(const f32x4v*)(x + i * 256 + lane * 4);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    const float inv_d = 1.0f / (float)p.d;
    float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < n) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float c = v[i][e] - mean;
                q += c * c;
            }
        }
    float rstd = rsqrtf(wave_sum(q) * inv_d + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < n) {
            const f32x4v g = *(const f32x4v*)(p.g1 + i * 256 + lane * 4);
            const f32x4v b = *(const f32x4v*)(p.b1 + i * 256 + lane * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][e] = (v[i][e] - mean) * rstd * g[e] + b[e];
        }
    if (p.g2) {
        s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < n) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        mean = wave_sum(s) * inv_d;
        q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < n) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float c = v[i][e] - mean;
                    q += c * c;
                }
            }
        rstd = rsqrtf(wave_sum(q) * inv_d + 1e-5f);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < n) {
                const f32x4v g = *(const f32x4v*)(p.g2 + i * 256 + lane * 4);
                const f32x4v b = *(const f32x4v*)(p.b2 + i * 256 + lane * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i][e] = (v[i][e] - mean) * rstd * g[e] + b[e];
            }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < n) {
            const long long o = (long long)r * p.d + i * 256 + lane * 4;
            if (p.Yf) *(f32x4v*)(p.Yf + o) = v[i];
            if (p.Yb) {
                bf16x4v w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = (bf16_t)v[i][e];
                *(bf16x4v*)(p.Yb + o) = w;
            }
        }
}

// Many-row single LayerNorm (candidate pass): RPW rows per wave so that RPW times the loads are in flight per wave
// (the kernel is a pure HBM stream; per-row arithmetic and its order are those of layernorm_vec_kernel).
template <int NV, int RPW>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(LnP p) {
    const int lane = threadIdx.x & 63;
    const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (r0 >= p.rows) return;
    f32x4v v[RPW][NV];
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        const int r = r0 + k < p.rows ? r0 + k : p.rows - 1;
        const float* x = p.X + (long long)map_row(p.xmap, r) * p.ldx;
#pragma unroll
        for (int i = 0; i < NV; ++i) v[k][i] = *(const f32x4v*)(x + i * 256 + lane * 4);
    }
    f32x4v g[NV], b[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        g[i] = *(const f32x4v*)(p.g1 + i * 256 + lane * 4);
        b[i] = *(const f32x4v*)(p.b1 + i * 256 + lane * 4);
    }
    const float inv_d = 1.0f / (float)p.d;
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += (v[k][i][0] + v[k][i][1]) + (v[k][i][2] + v[k][i][3]);
        const float mean = wave_sum(s) * inv_d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float c = v[k][i][e] - mean;
                q += c * c;
            }
        const float rstd = rsqrtf(wave_sum(q) * inv_d + 1e-5f);
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[k][i][e] = (v[k][i][e] - mean) * rstd * g[i][e] + b[i][e];
        if (p.g2) {  // second LayerNorm on the result (decoder.norm followed by an output head's norm)
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) s2 += (v[k][i][0] + v[k][i][1]) + (v[k][i][2] + v[k][i][3]);
            const float mean2 = wave_sum(s2) * inv_d;
            float q2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float c = v[k][i][e] - mean2;
                    q2 += c * c;
                }
            const float rstd2 = rsqrtf(wave_sum(q2) * inv_d + 1e-5f);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const f32x4v gg = *(const f32x4v*)(p.g2 + i * 256 + lane * 4);
                const f32x4v bb = *(const f32x4v*)(p.b2 + i * 256 + lane * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[k][i][e] = (v[k][i][e] - mean2) * rstd2 * gg[e] + bb[e];
            }
        }
        if (r0 + k < p.rows) {
            const long long o = (long long)(r0 + k) * p.d;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const f32x4v y = v[k][i];
                if (p.Yf) *(f32x4v*)(p.Yf + o + i * 256 + lane * 4) = y;
                if (p.Yb) {
                    bf16x4v w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = (bf16_t)y[e];
                    *(bf16x4v*)(p.Yb + o + i * 256 + lane * 4) = w;
                }
            }
        }
    }
}
template <int RPW>
static void launch_ln_rows(const LnP& p, hipStream_t st) {
    const dim3 grid((p.rows + 4 * RPW - 1) / (4 * RPW)), block(256);
    switch (p.d / 256) {
        case 1: hipLaunchKernelGGL((layernorm_rows_kernel<1, RPW>), grid, block, 0, st, p); return;
        case 2: hipLaunchKernelGGL((layernorm_rows_kernel<2, RPW>), grid, block, 0, st, p); return;
        case 3: hipLaunchKernelGGL((layernorm_rows_kernel<3, RPW>), grid, block, 0, st, p); return;
        default: hipLaunchKernelGGL((layernorm_rows_kernel<4, RPW>), grid, block, 0, st, p); return;
    }
}

void launch_layernorm(const LnP& p, hipStream_t st) {
    if (p.rows <= 0) return;
    if (p.d % 256 == 0 && p.d <= 1024 && p.ldx % 4 == 0 && p.rows >= 8192) {
        static const int rpw = M3PC_ENV("M3PC_LN_RPW") ? atoi(M3PC_ENV("M3PC_LN_RPW")) : 2;  // 1 = the one-row kernel below
        if (rpw == 4) return launch_ln_rows<4>(p, st);
        if (rpw == 2) return launch_ln_rows<2>(p, st);
    }
    if (p.d % 256 == 0 && p.ldx % 4 == 0)
        hipLaunchKernelGGL(layernorm_vec_kernel, dim3((p.rows + 3) / 4), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL(layernorm_kernel, dim3((p.rows + 3) / 4), dim3(256), 0, st, p);
}

// ------------------------------------------------------------------------------------------ head out
__global__ __launch_bounds__(256) void head_out_kernel(HeadOutP p) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= p.rows) return;
    const float* x = p.X + (long long)r * p.ldx;
    const int n = p.d >> 6;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < n) v[i] = x[i * 64 + lane];
    float* y = p.Y + (long long)map_row(p.ymap, r) * p.ldy;
    for (int f = 0; f < p.D; ++f) {
        const float* w = p.W + (long long)f * p.d;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < n) s = fmaf(v[i], w[i * 64 + lane], s);
        s = wave_sum(s) + p.b[f];
        if (p.mean) s = __fadd_rn(__fmul_rn(s, p.stdv[f]), p.mean[f]);
        if (lane == 0) y[f] = s;
    }
}
// The same on the matrix cores, for the many-row bf16 passes (round 6; critic_lambda_guiding's 17-wide states head ran 65536 rows
// per plan step of config 3 through head_out_kernel: one wave per row, a 64-lane butterfly per output feature -- 4.5 % of the step
// for 2 GFLOP).  Y^T[feature][row] = W2[feature][k] X^T[k][row]: W2 = the MFMA A operand (D <= 32 features padded to 32, split
// into a bf16 head and a bf16 remainder so that the weights keep ~16 mantissa bits: two MFMAs per k-step, the FLOPs are nothing),
// the rows' hidden activations (bf16: the GEMM + gelu in front writes them so, half the bytes of the fp32 rows) = the B operand,
// read straight from global memory in operand order (16 bytes of a lane's own row per k-step).  An accumulator holds lane = row,
// registers = features: bias and de-tokeniser apply per register, a lane stores its own row's values.  The W2 fragments sit in LDS
// (2 x 32 KiB, built once per workgroup); workgroups are persistent over 128-row tiles.
namespace {
typedef float hf32x16 __attribute__((ext_vector_type(16)));
typedef unsigned hu32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 hbf16x8 __attribute__((ext_vector_type(8)));
constexpr int HO_D = 512, HO_KS = HO_D / 16;
}
__global__ __launch_bounds__(256) void head_out_mfma_kernel(HeadOutP p, int n_tiles) {
    __shared__ __attribute__((aligned(16))) hu32x4 wfrag[2][HO_KS][64];  // [hi | lo][k-step][lane]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    for (int i = tid; i < HO_KS * 64; i += 256) {
        const int s = i >> 6, ln = i & 63, f = ln & 31, h = ln >> 5;
        hbf16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float w = f < p.D ? p.W[(long long)f * HO_D + 16 * s + 8 * h + j] : 0.f;
            hi[j] = (bf16_t)w;
            lo[j] = (bf16_t)(w - (float)hi[j]);
        }
        wfrag[0][s][ln] = __builtin_bit_cast(hu32x4, hi);
        wfrag[1][s][ln] = __builtin_bit_cast(hu32x4, lo);
    }
    __syncthreads();
    // this lane's features: register e <-> feature (e & 3) + 8 (e >> 2) + 4 lh
    float bias[16], sc[16], sh[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int f = (e & 3) + 8 * (e >> 2) + 4 * lh;
        const bool on = f < p.D;
        bias[e] = on ? p.b[f] : 0.f;
        sc[e] = on && p.mean ? p.stdv[f] : 1.f;
        sh[e] = on && p.mean ? p.mean[f] : 0.f;
    }
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int r = t * 128 + 32 * wv + l31;
        const int rl = r < p.rows ? r : p.rows - 1;
        const bf16_t* x = p.Xb + (long long)rl * p.ldx + 8 * lh;
        hf32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int c = 0; c < HO_KS; c += 8) {
            hu32x4 b[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) b[s] = *(const hu32x4*)(x + 16 * (c + s));
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const hbf16x8 bv = __builtin_bit_cast(hbf16x8, b[s]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hbf16x8, wfrag[1][c + s][lane]), bv, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hbf16x8, wfrag[0][c + s][lane]), bv, acc, 0, 0, 0);
            }
        }
        if (r < p.rows) {
            float* y = p.Y + (long long)map_row(p.ymap, r) * p.ldy;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int f = (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (f < p.D) {
                    float v = acc[e] + bias[e];
                    if (p.mean) v = __fadd_rn(__fmul_rn(v, sc[e]), sh[e]);
                    y[f] = v;
                }
            }
        }
    }
}
bool head_out_mfma_covers(int rows, int d, int D) { return d == HO_D && D >= 1 && D <= 32 && rows >= 2048; }
void launch_head_out(const HeadOutP& p, hipStream_t st) {
    if (p.rows <= 0) return;
    if (p.Xb) {  // (the caller asked head_out_mfma_covers first)
        const int n_tiles = (p.rows + 127) / 128;
        hipLaunchKernelGGL(head_out_mfma_kernel, dim3(n_tiles < 512 ? n_tiles : 512), dim3(256), 0, st, p, n_tiles);
        return;
    }
    hipLaunchKernelGGL(head_out_kernel, dim3((p.rows + 3) / 4), dim3(256), 0, st, p);
}

// ------------------------------------------------------------------------------------------ actor head
// One wave per row.  Optional LayerNorm of the row first (ln_g / ln_b: decoder.norm, mtm_model.py:705 -- layernorm_vec_kernel's
// arithmetic, so the pair "LayerNorm launch + actor launch" of a batch-1 policy pass is one launch); the row and the weight rows
// are read 16 bytes per lane, and the 2 A dot products are independent chains reduced by one batch of butterflies (the kernel
// is pure latency: 128 rows, a few KB of weights).
template <int AMAX>
__global__ __launch_bounds__(256) void actor_head_kernel(ActorP p) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= p.rows) return;
    const float* x = p.X + (long long)map_row(p.xmap, r) * p.ldx;
    const int n = p.d >> 8;  // slabs of 256 columns (<= 4)
    f32x4v v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < n) {
            v[i] = *(const f32x4v*)(x + i * 256 + lane * 4);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    if (p.ln_g) {
        const float inv_d = 1.0f / (float)p.d;
        const float mean = wave_sum(s) * inv_d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < n)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float c = v[i][e] - mean;
                    q += c * c;
                }
        const float rstd = rsqrtf(wave_sum(q) * inv_d + 1e-5f);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < n) {
                const f32x4v g = *(const f32x4v*)(p.ln_g + i * 256 + lane * 4);
                const f32x4v b = *(const f32x4v*)(p.ln_b + i * 256 + lane * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i][e] = (v[i][e] - mean) * rstd * g[e] + b[e];
            }
    }
    float s1[AMAX], s2[AMAX];
#pragma unroll
    for (int f = 0; f < AMAX; ++f) {
        s1[f] = 0.f;
        s2[f] = 0.f;
        const int ff = f < p.A ? f : p.A - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < n) {
                const f32x4v wm = *(const f32x4v*)(p.Wmu + (long long)ff * p.d + i * 256 + lane * 4);
                const f32x4v wl = *(const f32x4v*)(p.Wls + (long long)ff * p.d + i * 256 + lane * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s1[f] = fmaf(v[i][e], wm[e], s1[f]);
                    s2[f] = fmaf(v[i][e], wl[e], s2[f]);
                }
            }
    }
#pragma unroll
    for (int f = 0; f < AMAX; ++f) {
        s1[f] = wave_sum(s1[f]);
        s2[f] = wave_sum(s2[f]);
    }
#pragma unroll
    for (int f = 0; f < AMAX; ++f)
        if (f < p.A && lane == f) {  // (one lane per feature: the tanh / exp of the features run side by side)
            p.mu[(long long)r * p.A + f] = s1[f] + p.bmu[f];
            float ls = tanhf(s2[f] + p.bls[f]);
            ls = -5.0f + 0.5f * (2.0f - (-5.0f)) * (ls + 1.0f);
            p.sd[(long long)r * p.A + f] = expf(ls);
        }
}
// d not a multiple of 256 (the tiny test configurations): the row 4 bytes per lane, no LayerNorm
__global__ __launch_bounds__(256) void actor_head_small_kernel(ActorP p) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= p.rows) return;
    const float* x = p.X + (long long)map_row(p.xmap, r) * p.ldx;
    const int n = p.d >> 6;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < n) v[i] = x[i * 64 + lane];
    for (int f = 0; f < p.A; ++f) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < n) {
                s1 = fmaf(v[i], p.Wmu[(long long)f * p.d + i * 64 + lane], s1);
                s2 = fmaf(v[i], p.Wls[(long long)f * p.d + i * 64 + lane], s2);
            }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) {
            p.mu[(long long)r * p.A + f] = s1 + p.bmu[f];
            float ls = tanhf(s2 + p.bls[f]);
            ls = -5.0f + 0.5f * (2.0f - (-5.0f)) * (ls + 1.0f);
            p.sd[(long long)r * p.A + f] = expf(ls);
        }
    }
}
bool actor_head_fuses_ln(int d, int A) { return d % 256 == 0 && d <= 1024 && A <= 8; }
void launch_actor_head(const ActorP& p, hipStream_t st) {
    if (p.rows <= 0) return;
    const dim3 grid((p.rows + 3) / 4), block(256);
    if (actor_head_fuses_ln(p.d, p.A)) {
        if (p.A <= 4) hipLaunchKernelGGL((actor_head_kernel<4>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((actor_head_kernel<8>), grid, block, 0, st, p);
    } else {
        hipLaunchKernelGGL(actor_head_small_kernel, grid, block, 0, st, p);  // (ln_g must be null: callers ask actor_head_fuses_ln)
    }
}

// ------------------------------------------------------------------------------------------ sampling
__global__ __launch_bounds__(256) void sample_kernel(SampleP p) {
    const long long tot = (long long)p.n_count * p.T * p.A;
    if (blockIdx.x == 0) {
        for (int x = threadIdx.x; x < p.T * p.A; x += blockDim.x) {
            if (p.loc_out && p.loc) p.loc_out[x] = p.loc[x];
            if (p.sd_out && p.sd) p.sd_out[x] = p.sd[x];
        }
    }
    for (long long x = blockIdx.x * (long long)blockDim.x + threadIdx.x; x < tot; x += (long long)gridDim.x * blockDim.x) {
        const int a = (int)(x % p.A);
        const int t = (int)((x / p.A) % p.T);
        const long long n = x / ((long long)p.A * p.T);
        float v;
        if (t < p.idx) {
            v = p.hist_actions[(p.widx ? (long long)p.widx[n] * p.T * p.A : 0) + t * p.A + a];
        } else {
            const int ta = t * p.A + a;
            const long long src = p.index ? (long long)p.index[n] : p.n_begin + n;
            if (p.mode == 2) {
                v = p.eps[(src * p.h + (t - p.idx)) * p.A + a];
            } else if (p.mode == 0) {
                const float e = p.eps[(src * p.T + t) * p.A + a];
                v = tanhf(__fadd_rn(__fmul_rn(e, p.sd[ta]), p.loc[ta]));
            } else {
                const float e = p.eps[(src * p.h + (t - p.idx)) * p.A + a];
                v = __fadd_rn(tanhf(p.loc[ta]), __fmul_rn(e, 0.09f));
                v = fminf(fmaxf(v, -0.99999f), 0.99999f);
            }
            if (p.sample_actions) p.sample_actions[(n * p.h + (t - p.idx)) * p.A + a] = v;
        }
        p.cand[x] = v;
    }
}
void launch_sample(const SampleP& p, hipStream_t st) {
    const long long tot = (long long)p.n_count * p.T * p.A;
    if (tot <= 0) return;
    const int grid = (int)((tot + 255) / 256 < 2048 ? (tot + 255) / 256 : 2048);
    hipLaunchKernelGGL(sample_kernel, dim3(grid), dim3(256), 0, st, p);
}

// ------------------------------------------------------------------------------------------ critic
// 16 rows per 256-thread block; thread c owns hidden unit c of both layers (hidden <= 256).  Inputs and
// the first hidden layer live in LDS and are read as broadcasts.
#define CR_ROWS 16
__global__ __launch_bounds__(256) void critic_kernel(CriticP p) {
    __shared__ float sa[CR_ROWS][32];
    __shared__ float h1[CR_ROWS][256];
    __shared__ float red[CR_ROWS][4];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * CR_ROWS;
    const int SA = p.S + p.A, Hd = p.hidden;
    const bool on = tid < Hd;
    for (int x = tid; x < CR_ROWS * SA; x += 256) {
        const int rr = x / SA, f = x % SA, r = r0 + rr;
        float v = 0.f;
        if (r < p.rows) v = f < p.S ? (p.states[(long long)r * p.S + f] - p.om[f]) / p.os[f] : p.actions[(long long)r * p.A + (f - p.S)];
        sa[rr][f] = v;
    }
    float qmin = 0.f;
    for (int net = 0; net < 2; ++net) {
        __syncthreads();
        {
            float acc[CR_ROWS];
#pragma unroll
            for (int rr = 0; rr < CR_ROWS; ++rr) acc[rr] = 0.f;
            if (on) {
                for (int f = 0; f < SA; ++f) {
                    const float w = p.W1T[net][f * Hd + tid];
#pragma unroll
                    for (int rr = 0; rr < CR_ROWS; ++rr) acc[rr] = fmaf(sa[rr][f], w, acc[rr]);
                }
                const float bb = p.b1[net][tid];
#pragma unroll
                for (int rr = 0; rr < CR_ROWS; ++rr) h1[rr][tid] = fmaxf(acc[rr] + bb, 0.f);
            }
        }
        __syncthreads();
        float acc[CR_ROWS];
#pragma unroll
        for (int rr = 0; rr < CR_ROWS; ++rr) acc[rr] = 0.f;
        float bb = 0.f, w3 = 0.f;
        if (on) {
            for (int k = 0; k < Hd; ++k) {
                const float w = p.W2T[net][k * Hd + tid];
#pragma unroll
                for (int rr = 0; rr < CR_ROWS; ++rr) acc[rr] = fmaf(h1[rr][k], w, acc[rr]);
            }
            bb = p.b2[net][tid];
            w3 = p.W3[net][tid];
        }
#pragma unroll
        for (int rr = 0; rr < CR_ROWS; ++rr) {
            float v = on ? fmaxf(acc[rr] + bb, 0.f) * w3 : 0.f;
            v = wave_sum(v);
            if ((tid & 63) == 0) red[rr][tid >> 6] = v;
        }
        __syncthreads();
        if (tid < CR_ROWS) {
            const float q = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3] + p.b3[net][0];
            qmin = net == 0 ? q : fminf(qmin, q);
            if (net == 1 && r0 + tid < p.rows) p.q[r0 + tid] = qmin;
        }
    }
}
// The same network on the matrix cores in fp32 (v_mfma_f32_32x32x2_f32: every product an exact fp32 fma), transposed like
// the fused layer tail: weights = A operand, data rows = B operand, so a lane owns ONE data row (lane & 31) and its
// registers run over hidden units -- acc[i] <-> unit 32 t + 8 (i / 4) + 4 (lane >> 5) + i % 4.  An accumulator after
// bias + ReLU therefore IS the B operand of the next layer's MFMAs: step (g, j) of layer 2 multiplies the k pair
// {8 g + j, 8 g + 4 + j} (one per lane half), which is register 4 (g % 4) + j of hidden tile g / 4 in both halves.  The 256
// hidden values of a row never leave the registers; no LDS, no barrier.  Weights are packed once (m3pc_set_critic) in
// operand order, four steps per lane and load: one 1-KiB coalesced load per four MFMAs and wave, read four groups ahead.
//   W1F: [tile t][group g < 4][lane][j]  = W1[32 t + (lane & 31)][2 (4 g + j) + (lane >> 5)]   (features padded to 32 with zeros)
//   W2F: [tile u][group g < hidden / 8][lane][j] = W2[32 u + (lane & 31)][8 g + 4 (lane >> 5) + j]   (+ 4 KiB of padding)
// A wave = 32 data rows, 2 x (16 + hidden / 2) x hidden / 32 MFMAs of 64 clocks: 65536 rows (walker2d critic N = 4096) take
// what 2048 waves on 1024 SIMDs take, ~2 x 143 k clocks, where the scalar kernel above (LDS-broadcast bound) took 680 us of
// chip time.  Row results do not depend on the row's position or on the number of rows.
template <int NT>
__global__ __launch_bounds__(256, 2) void critic_mfma_kernel(CriticP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float f32x16v __attribute__((ext_vector_type(16)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int r = (blockIdx.x * 4 + wave) * 32 + l31;
    if ((blockIdx.x * 4 + wave) * 32 >= p.rows) return;  // (wave-uniform; no barriers in this kernel)
    const long long rl = r < p.rows ? r : p.rows - 1;
    const int SA = p.S + p.A;
    float x[16];  // feature 2 s + lh of this lane's row
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int f = 2 * s + lh;
        float v = 0.f;
        if (f < p.S) v = (p.states[rl * p.S + f] - p.om[f]) / p.os[f];
        else if (f < SA) v = p.actions[rl * p.A + (f - p.S)];
        x[s] = v;
    }
    float qmin = 0.f;
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        f32x16v h1[NT];
        const f32x4v* w1p = (const f32x4v*)p.W1F[net] + lane;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4v b = *(const f32x4v*)(p.b1[net] + 32 * t + 8 * q + 4 * lh);
#pragma unroll
                for (int i = 0; i < 4; ++i) h1[t][4 * q + i] = b[i];
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4v w = w1p[(t * 4 + g) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j) h1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j], x[4 * g + j], h1[t], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) h1[t][i] = fmaxf(h1[t][i], 0.f);
        }
        const f32x4v* w2p = (const f32x4v*)p.W2F[net] + lane;
        f32x4v wf[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) wf[g] = w2p[g * 64];
        float qa = 0.f;
        for (int u = 0; u < NT; ++u) {
            f32x16v acc;
            f32x4v w3[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4v b = *(const f32x4v*)(p.b2[net] + 32 * u + 8 * q + 4 * lh);
                w3[q] = *(const f32x4v*)(p.W3[net] + 32 * u + 8 * q + 4 * lh);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[4 * q + i] = b[i];
            }
#pragma unroll
            for (int g = 0; g < 4 * NT; ++g) {
                const f32x4v w = wf[g % 4];
                wf[g % 4] = w2p[((u * 4 * NT) + g + 4) * 64];  // (runs 4 groups past the end: the buffer is padded)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j], h1[g / 4][4 * (g % 4) + j], acc, 0, 0, 0);
            }
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) sum = fmaf(fmaxf(acc[i], 0.f), w3[i / 4][i % 4], sum);
            sum += __shfl_xor(sum, 32);
            qa += sum;
        }
        const float q = qa + p.b3[net][0];
        qmin = net == 0 ? q : fminf(qmin, q);
    }
    if (lh == 0 && r < p.rows) p.q[r] = qmin;
#endif
}
bool critic_mfma_covers(int S, int A, int hidden) { return S + A <= 32 && (hidden == 64 || hidden == 128 || hidden == 256); }
size_t critic_w1f_floats(int hidden) { return (size_t)(hidden / 32) * 4 * 256; }
size_t critic_w2f_floats(int hidden) { return (size_t)(hidden / 32) * (hidden / 8) * 256 + 1024; }
// host-side packing of one network's two weight matrices (W1: hidden x SA, W2: hidden x hidden, row-major as in the state dict)
void critic_pack(const float* W1, const float* W2, int SA, int hidden, float* w1f, float* w2f) {
    const int NT = hidden / 32;
    for (int t = 0; t < NT; ++t)
        for (int g = 0; g < 4; ++g)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const int f = 2 * (4 * g + j) + (lane >> 5);
                    w1f[(((size_t)t * 4 + g) * 64 + lane) * 4 + j] = f < SA ? W1[(size_t)(32 * t + (lane & 31)) * SA + f] : 0.f;
                }
    const int NG = hidden / 8;
    for (int u = 0; u < NT; ++u)
        for (int g = 0; g < NG; ++g)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j)
                    w2f[(((size_t)u * NG + g) * 64 + lane) * 4 + j] = W2[(size_t)(32 * u + (lane & 31)) * hidden + 8 * g + 4 * (lane >> 5) + j];
    for (size_t i = (size_t)NT * NG * 256; i < critic_w2f_floats(hidden); ++i) w2f[i] = 0.f;
}
void launch_critic(const CriticP& p, hipStream_t st) {
    if (p.rows <= 0) return;
    if (p.W1F[0] && critic_mfma_covers(p.S, p.A, p.hidden)) {
        const dim3 grid((p.rows + 127) / 128), block(256);
        if (p.hidden == 256) hipLaunchKernelGGL(critic_mfma_kernel<8>, grid, block, 0, st, p);
        else if (p.hidden == 128) hipLaunchKernelGGL(critic_mfma_kernel<4>, grid, block, 0, st, p);
        else hipLaunchKernelGGL(critic_mfma_kernel<2>, grid, block, 0, st, p);
        return;
    }
    hipLaunchKernelGGL(critic_kernel, dim3((p.rows + CR_ROWS - 1) / CR_ROWS), dim3(256), 0, st, p);
}

// ------------------------------------------------------------------------------------------ scoring
// Same operation order as the reference loop (learner.py:300-316): per step t the discounted values are
// summed left to right, multiplied by (1-lambda) then lambda^t (fp32 each), and accumulated.
__global__ __launch_bounds__(256) void score_kernel(ScoreP p) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= p.n) return;
    const float* rw = p.rewards + (long long)n * p.h;
    const float* bt = p.boot + (long long)n * p.h;
    const float one_minus = (float)(1.0 - p.lmbda);
    float er = 0.f;
    double lam_t = 1.0;  // python float lmbda**t
    // The reference restarts the discounted reward sum at every t; summed left to right it is a prefix of the
    // next t's sum and disc is the same repeated product, so one running (pre, disc) pair reproduces every t bit
    // for bit in O(h) instead of O(h^2).
    float disc = p.gamma, pre = 0.f;
    for (int t = 0; t < p.h; ++t) {
        const float b = __fmul_rn(bt[t], p.boot_scale);
        if (p.boot_out) p.boot_out[(long long)n * p.h + t] = b;
        const float s = __fadd_rn(pre, __fmul_rn(b, disc));
        float w;
        if (t < p.h - 1)
            w = __fmul_rn(__fmul_rn(s, one_minus), (float)lam_t);
        else
            w = __fmul_rn(s, (float)lam_t);
        er = __fadd_rn(er, w);
        lam_t *= p.lmbda;
        pre = __fadd_rn(pre, __fmul_rn(rw[t], disc));
        disc = __fmul_rn(disc, p.gamma);
    }
    p.expect_return[n] = er;
    if (p.scatter_out) p.scatter_out[p.scatter_index[n]] = er;
}
void launch_score(const ScoreP& p, hipStream_t st) {
    if (p.n <= 0) return;
    hipLaunchKernelGGL(score_kernel, dim3((p.n + 255) / 256), dim3(256), 0, st, p);
}

// ------------------------------------------------------------------------------------------ tokenizer
__global__ __launch_bounds__(256) void tokenize_kernel(const void* in, int in_f64, float* out, long long n, int D,
                                                       const float* mean, const float* stdv, int normalize) {
    for (long long x = blockIdx.x * (long long)blockDim.x + threadIdx.x; x < n; x += (long long)gridDim.x * blockDim.x) {
        const int f = (int)(x % D);
        if (in_f64) {
            double v = ((const double*)in)[x];
            if (normalize) v = (v - (double)mean[f]) / (double)stdv[f];
            out[x] = (float)v;
        } else {
            float v = ((const float*)in)[x];
            if (normalize) v = (v - mean[f]) / stdv[f];
            out[x] = v;
        }
    }
}
__global__ __launch_bounds__(256) void detokenize_kernel(const float* in, float* out, long long n, int D, const float* mean,
                                                         const float* stdv, int normalize) {
    for (long long x = blockIdx.x * (long long)blockDim.x + threadIdx.x; x < n; x += (long long)gridDim.x * blockDim.x) {
        const int f = (int)(x % D);
        float v = in[x];
        if (normalize) v = __fadd_rn(__fmul_rn(v, stdv[f]), mean[f]);
        out[x] = v;
    }
}
static int grid_for(long long n) {
    const long long g = (n + 255) / 256;
    return (int)(g < 2048 ? (g > 0 ? g : 1) : 2048);
}
void launch_tokenize(const void* in, int in_f64, float* out, long long rows, int D, const float* mean, const float* stdv,
                     int normalize, hipStream_t st) {
    const long long n = rows * D;
    if (n <= 0) return;
    hipLaunchKernelGGL(tokenize_kernel, dim3(grid_for(n)), dim3(256), 0, st, in, in_f64, out, n, D, mean, stdv, normalize);
}
void launch_detokenize(const float* in, float* out, long long rows, int D, const float* mean, const float* stdv,
                       int normalize, hipStream_t st) {
    const long long n = rows * D;
    if (n <= 0) return;
    hipLaunchKernelGGL(detokenize_kernel, dim3(grid_for(n)), dim3(256), 0, st, in, out, n, D, mean, stdv, normalize);
}

// zero-shot path inference -> inverse dynamics hand-over (zeroshot_omtm/learner.py:240-246): the states head's raw output is
// de-tokenised in place (`pred`, (B,T,S)) and written over the window's observation rows [0, idx] and [idx+2, T-2]
__global__ __launch_bounds__(256) void goal_overlay_kernel(float* pred, const float* states_in, float* states_out, long long n, int T,
                                                           int D, int idx, const float* mean, const float* stdv, int normalize) {
    for (long long x = blockIdx.x * (long long)blockDim.x + threadIdx.x; x < n; x += (long long)gridDim.x * blockDim.x) {
        const int f = (int)(x % D), t = (int)((x / D) % T);
        float v = pred[x];
        if (normalize) v = __fadd_rn(__fmul_rn(v, stdv[f]), mean[f]);
        pred[x] = v;
        states_out[x] = (t <= idx || (t >= idx + 2 && t < T - 1)) ? v : states_in[x];
    }
}
void launch_goal_overlay(float* pred, const float* states_in, float* states_out, long long rows, int T, int D, int idx,
                         const float* mean, const float* stdv, int normalize, hipStream_t st) {
    const long long n = rows * D;
    if (n <= 0) return;
    hipLaunchKernelGGL(goal_overlay_kernel, dim3(grid_for(n)), dim3(256), 0, st, pred, states_in, states_out, n, T, D, idx, mean, stdv, normalize);
}

// The same hand-over for the pruned path inference (m3pc_goal_step_batch): the states head ran on the window rows the
// overlay reads only -- rows t <= idx, then idx + 2 <= t <= T - 2, in that order, nq per window -- and `pred` (E * nq, D)
// is already de-tokenised (head_out_kernel).  states_out = states_in with those rows replaced.
__global__ __launch_bounds__(256) void goal_overlay_rows_kernel(const float* pred, const float* states_in, float* states_out, long long n,
                                                                int T, int D, int idx, int nq) {
    for (long long x = blockIdx.x * (long long)blockDim.x + threadIdx.x; x < n; x += (long long)gridDim.x * blockDim.x) {
        const int f = (int)(x % D), t = (int)((x / D) % T);
        const long long e = x / ((long long)D * T);
        int q = -1;
        if (t <= idx) q = t;
        else if (t >= idx + 2 && t < T - 1) q = idx + 1 + (t - (idx + 2));
        states_out[x] = q >= 0 ? pred[(e * nq + q) * D + f] : states_in[x];
    }
}
void launch_goal_overlay_rows(const float* pred, const float* states_in, float* states_out, long long windows, int T, int D, int idx,
                              int nq, hipStream_t st) {
    const long long n = windows * T * D;
    if (n <= 0) return;
    hipLaunchKernelGGL(goal_overlay_rows_kernel, dim3(grid_for(n)), dim3(256), 0, st, pred, states_in, states_out, n, T, D, idx, nq);
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* in, bf16_t* out, long long n) {
    for (long long x = blockIdx.x * (long long)blockDim.x + threadIdx.x; x < n; x += (long long)gridDim.x * blockDim.x)
        out[x] = (bf16_t)in[x];
}
void launch_f32_to_bf16(const float* in, bf16_t* out, long long n, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, st, in, out, n);
}

__global__ __launch_bounds__(256) void fill_kernel(float* out, float value, long long n) {
    for (long long x = blockIdx.x * (long long)blockDim.x + threadIdx.x; x < n; x += (long long)gridDim.x * blockDim.x) out[x] = value;
}
void launch_fill(float* out, float value, long long n, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, st, out, value, n);
}

// Derived weight tables (m3pc_load_weights), computed on the device so that a weight update costs no host round trip:
//   transpose: out[j * rows + c] = in[c * cols + j]   (encoder_embed weight (d, D_k) -> (D_k, d))
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* in, float* out, int rows, int cols) {
    const long long n = (long long)rows * cols;
    for (long long x = blockIdx.x * (long long)blockDim.x + threadIdx.x; x < n; x += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(x / cols), j = (int)(x % cols);
        out[(long long)j * rows + c] = in[x];
    }
}
void launch_transpose_f32(const float* in, float* out, int rows, int cols, hipStream_t st) {
    if (rows <= 0 || cols <= 0) return;
    hipLaunchKernelGGL(transpose_f32_kernel, dim3(grid_for((long long)rows * cols)), dim3(256), 0, st, in, out, rows, cols);
}
//   embedding table: out[t, c] = (bias[c] + per_dim[c]) + pos[t, c]   (mtm_model.py:549-553 / 653-657; this association)
__global__ __launch_bounds__(256) void embed_table_kernel(const float* bias, const float* per_dim, const float* pos, float* out, int T,
                                                          int d) {
    const long long n = (long long)T * d;
    for (long long x = blockIdx.x * (long long)blockDim.x + threadIdx.x; x < n; x += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(x % d);
        out[x] = (bias[c] + per_dim[c]) + pos[x];
    }
}
void launch_embed_table(const float* bias, const float* per_dim, const float* pos, float* out, int T, int d, hipStream_t st) {
    hipLaunchKernelGGL(embed_table_kernel, dim3(grid_for((long long)T * d)), dim3(256), 0, st, bias, per_dim, pos, out, T, d);
}

}  // namespace m3pc

// The tail of one pre-LN transformer layer (mtm_model.py:379-409, the nn.TransformerEncoderLayer halves after the
// attention itself) as ONE launch on bf16 operands:
//
//     X' = res + bo + O Wo^T              out-proj + residual
//     a  = LayerNorm2(X')                 (bf16)
//     H  = gelu(a W1^T + b1)              (bf16, never leaves the registers)
//     X''= X' + b2 + H W2^T               FFN2 + residual
//     Hout = bf16(LN_A(X''))  [LN_B on top]  the LayerNorm(s) that consume the block output, or -- an encoder layer followed by
//     QKV  = bf16(LN_A(X'') Wqkv^T + bqkv)    another -- the next layer's whole Q|K|V projection (Hout then never exists)
//     or -- the decoder layer of a plan step whose two scored keys have scalar heads (rewards, returns: mtm_model.py:428-433) --
//     y = Linear(512,1)(gelu(Linear(512,512)(LN_head(LN_A(X''))))) de-tokenised: the whole output head (TAIL == 2; a
//     workgroup then owns 128 rows of ONE of the two keys, so that its four waves stream one head's weights)
//
// d = 512, ff = 2048.  A workgroup owns 128 token rows, each of its four waves (one per SIMD, the whole 512-entry
// register file) a strip of 32 of them -- through the whole chain, so nothing but O, the residual rows and the
// outputs touches HBM and no activation ever crosses a wave.  Every product is computed TRANSPOSED,
//     D^T[feature][token] += W[feature][k] * act^T[k][token]
// i.e. the weights are the MFMA A operand (32 output features x 16 k per v_mfma_f32_32x32x16_bf16) and the
// activations the B operand.  An accumulator then holds  lane = token, registers = features:
//   * a finished accumulator tile, rounded pairwise to bf16, IS the B operand of the next product (registers
//     8s..8s+7 are the fragment of k-step s, cdna_hip_programming.md section 3) -- the hidden chunk goes from FFN1 to
//     FFN2 and the LayerNorm-2 rows go into FFN1 without a lane ever exchanging data;
//   * LayerNorm statistics are sums over a lane's own registers plus one exchange with lane ^ 32;
//   * X' simply stays in the 256 accumulator registers while FFN2 adds to it.
// Register order of such a fragment permutes k inside each group of 16 (element j of lane half h is feature
// 16 s + 8 (j >> 2) + 4 h + (j & 3)); the weights are packed with the same permutation, so the dot products are complete.
//
// Weights: the A fragments of the whole chain, 1 KiB each (64 lanes x 16 B, exactly the register image), are packed
// ONCE per weight load in the order the kernel consumes them (pack_block_stream): 4608 fragments = 4.5 MiB per layer.
// The kernel is then a linear stream: LDS-DMA pieces of 1 KiB (whole cache lines, contiguous source, contiguous
// destination, one scalar offset), a ring of three 32-KiB stages (32 fragments = 32 MFMAs per wave), one
// ds_read_b128 per MFMA at lane * 16 + immediate -- conflict-free by construction, no swizzle, no address arithmetic.
// Per byte of L2->LDS traffic a 128-row tile does 128 FLOP (a 128x128 GEMM tile that also streams its A operand: 64).
//
// Phase order (144 phases of 32 MFMAs = one ring stage each per tile):
//   out-proj, phase jn = 0..15:  all 32 k-steps of feature tile jn (one accumulator chain of 32 MFMAs); tile jn's
//            accumulator starts at (residual + bias) of its 32 features, fetched three phases ahead: the residual rows
//            -- two thirds of a tile's input bytes -- stream in UNDER the out-proj instead of in front of it (round 4)
//   FFN, chunk c = 0..31 of 64 hidden units:  A0 A1 B1 B2, interleaved with the neighbouring chunks' (A0(c) B2(c-1) A1(c) B1(c))
//            A0 / A1: hidden tile 64c..+31 / 64c+32..+63 over the 32 k-steps of the model width (FFN1)
//            B1 / B2: hidden k-steps 0,1 / 2,3 of the chunk into all 16 feature tiles (FFN2)
//            gelu of hidden tile 0 runs on the VALU beside the MFMAs of B2(c-1) and A1, that of tile 1 beside B1 and A0(c+1).
// Registers: 256 accumulators + the 32 LayerNorm-2 fragments (128) + hidden tiles, their bf16 fragments and the
// fragment window do not fit 512 with room for the compiler, so the fragments of k-steps 24..31 live in LDS (8 KiB per
// wave, private to it) and are read like the weights, one per group of four MFMAs.
// Sync: ONE s_waitcnt vmcnt(7) lgkmcnt(0) + s_barrier per stage, placed eight MFMAs before the stage ends: every wave
// has read all of stage t (slot t % 3 is free: stage t+3's pieces go there) and stage t+1 has landed everywhere (its
// first fragments are read under the last MFMAs of stage t).  Stage t+2 stays in flight across the barrier.
// Fragments are read four at a time, two groups (eight MFMAs) ahead: one lgkmcnt wait per four MFMAs.
// With one wave per SIMD the kernel is bound by what that wave has to ISSUE besides its MFMAs (an MFMA leaves room for
// about five other instructions); the layout above is what keeps that count down.
#include <type_traits>

#include "gemm_epilogue.h"
#include "kernels.h"

namespace m3pc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef void __attribute__((address_space(3))) * lptr_t;

namespace {
constexpr int BD = 512, BFF = 2048;              // model width / FFN width this kernel is built for
constexpr int NT = BD / 32;                      // 16 feature tiles of 32
constexpr int KS = BD / 16;                      // 32 k-steps over the model width
constexpr int NCH = BFF / 64;                    // 32 hidden chunks of 64
constexpr int FR_OUT = NT * KS;                  // 512 out-proj fragments
constexpr int FR_TOTAL = FR_OUT + NCH * 128;     // 4608
constexpr int RS_FR = 32, RS_B = RS_FR * 1024, NRS = FR_TOTAL / RS_FR;  // 144 ring stages (= phases) of 32 KiB
constexpr int QKV_FR = 3 * FR_OUT;                // the next layer's Q|K|V projection behind the layer's own fragments: 1536
constexpr int NRS_QKV = NRS + QKV_FR / RS_FR;     // 192 stages
constexpr int HEAD_FR = FR_OUT;                   // one output head's Linear(512,512): 16 hidden tiles x 32 k-steps
constexpr int NRS_HEAD = NRS + HEAD_FR / RS_FR;   // 160 stages (the stream holds both heads: 144 .. 175)
constexpr int SPLIT_N = 4;                        // TAIL == 3: workgroups per 128-row tile
constexpr int NSLOT = 3;
constexpr int ACT_LDS = 8;                       // LayerNorm-2 fragments of k-steps 24..31 live in LDS: 8 KiB per wave
constexpr int ACT_OFF = NSLOT * RS_B;
// parameter tables (floats) behind them
constexpr int T_B1 = 0, T_B2 = T_B1 + BFF, T_G2 = T_B2 + BD, T_BE2 = T_G2 + BD, T_GA = T_BE2 + BD, T_BA = T_GA + BD,
              T_GB = T_BA + BD, T_BB = T_GB + 2 * BD, T_BO = T_BB + 2 * BD, T_END = T_BO + BD;
constexpr int TAB_OFF = ACT_OFF + 4 * ACT_LDS * 1024;
constexpr int LDS_BYTES = TAB_OFF + T_END * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
// k-step of the i-th MFMA of an FFN1 phase: per group of four, three k-steps from registers (0..23), one from LDS (24..31)
__host__ __device__ constexpr int a_kstep(int i) { return (i % 4 < 3) ? 3 * (i / 4) + i % 4 : 24 + i / 4; }
}  // namespace

// ------------------------------------------------------------------------------------------------ weight stream
// element j of lane (r = lane & 31, h = lane >> 5) of fragment f
__global__ __launch_bounds__(256) void pack_block_stream_kernel(const bf16_t* __restrict__ Wo, const bf16_t* __restrict__ W1,
                                                                const bf16_t* __restrict__ W2, bf16_t* __restrict__ out) {
    const int gid = blockIdx.x * 256 + threadIdx.x;  // one thread per 16-byte piece
    if (gid >= FR_TOTAL * 64) return;
    const int f = gid >> 6, lane = gid & 63, r = lane & 31, h = lane >> 5;
    bf16_t v[8];
    if (f < FR_OUT) {  // phase jn: the 32 k-steps of feature tile jn, natural k order (the operand is loaded from O)
        const int jn = f / 32, ks = f % 32;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = Wo[(size_t)(32 * jn + r) * BD + 16 * ks + 8 * h + j];
    } else {
        // FFN phases in EXECUTION order (ffn_phase_pos below): A0(0) A1(0) B1(0), then A0(c) B2(c-1) A1(c) B1(c) for c = 1..31, B2(31)
        const int g = f - FR_OUT, P = g / 32, i = g % 32;
        int kind, c;  // 0: A0, 1: A1, 2: B1, 3: B2 of chunk c
        if (P < 3) {
            c = 0, kind = P;
        } else if (P == 4 * NCH - 1) {
            c = NCH - 1, kind = 3;
        } else {
            const int q = P - 3, k = q % 4;
            c = 1 + q / 4;
            kind = k == 0 ? 0 : k == 1 ? 3 : k == 2 ? 1 : 2;
            if (k == 1) c -= 1;
        }
        if (kind < 2) {  // A0 | A1: hidden tile t, MFMA i (permuted k order: the operand is a LayerNorm-2 accumulator)
            const int t = kind, s = a_kstep(i);
            const size_t row = (size_t)(64 * c + 32 * t + r) * BD;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = W1[row + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)];
        } else {         // B1 | B2: hidden k-step s2 of the chunk, feature tile jn
            const int s2 = 2 * (kind - 2) + i / 16, jn = i % 16;
            const size_t row = (size_t)(32 * jn + r) * BFF + 64 * c + 16 * s2;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = W2[row + 8 * (j >> 2) + 4 * h + (j & 3)];
        }
    }
    bf16_t* o = out + (size_t)gid * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = v[j];
}

// fragments FR_TOTAL.. of a layer's stream: the NEXT layer's in_proj rows (Q | K | V: 48 feature tiles, one phase of 32 k-steps
// each, permuted k -- the other operand is the LayerNorm of an accumulator)
__global__ __launch_bounds__(256) void pack_block_qkv_kernel(const bf16_t* __restrict__ Wqkv, bf16_t* __restrict__ out) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= QKV_FR * 64) return;
    const int f = gid >> 6, lane = gid & 63, r = lane & 31, h = lane >> 5;
    const int jt = f / KS, ks = f % KS;  // phase jt = the 32 k-steps of feature tile jt (of the 48 of Q | K | V)
    const size_t row = (size_t)(32 * jt + r) * BD;
    bf16_t* o = out + (size_t)(FR_TOTAL * 64 + gid) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = Wqkv[row + 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3)];
}

// fragments FR_TOTAL.. of the decoder layer's stream: the first Linear of the two scalar output heads, head s at
// FR_TOTAL + 512 s: hidden tile t (32 units), MFMA i in the k order of an FFN1 phase (the other operand is a LayerNorm'd accumulator)
__global__ __launch_bounds__(256) void pack_block_heads_kernel(const bf16_t* __restrict__ Wh0, const bf16_t* __restrict__ Wh1,
                                                               bf16_t* __restrict__ out) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= 2 * HEAD_FR * 64) return;
    const int f = gid >> 6, lane = gid & 63, r = lane & 31, h = lane >> 5;
    const int hs = f / HEAD_FR, w = f % HEAD_FR, t = w / KS, ks = a_kstep(w % KS);
    const bf16_t* W = hs ? Wh1 : Wh0;
    const size_t row = (size_t)(32 * t + r) * BD;
    bf16_t* o = out + (size_t)(FR_TOTAL * 64 + gid) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = W[row + 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3)];
}

size_t block_stream_bytes() { return (size_t)(FR_TOTAL + QKV_FR) * 1024; }
bool block_fused_supported(int d, int ff) { return d == BD && ff == BFF; }

void launch_pack_block_stream(const bf16_t* Wo, const bf16_t* W1, const bf16_t* W2, bf16_t* out, hipStream_t st) {
    hipLaunchKernelGGL(pack_block_stream_kernel, dim3(FR_TOTAL * 64 / 256), dim3(256), 0, st, Wo, W1, W2, out);
}
void launch_pack_block_heads(const bf16_t* Wh0, const bf16_t* Wh1, bf16_t* out, hipStream_t st) {
    hipLaunchKernelGGL(pack_block_heads_kernel, dim3(2 * HEAD_FR * 64 / 256), dim3(256), 0, st, Wh0, Wh1, out);
}
void launch_pack_block_qkv(const bf16_t* Wqkv_next, bf16_t* out, hipStream_t st) {
    hipLaunchKernelGGL(pack_block_qkv_kernel, dim3(QKV_FR * 64 / 256), dim3(256), 0, st, Wqkv_next, out);
}

// ------------------------------------------------------------------------------------------------ the kernel
namespace {

__device__ __forceinline__ float half_swap_sum(float v) { return v + __shfl_xor(v, 32); }

// Every MFMA is an asm statement with fixed register classes: the 256 feature accumulators ARE the accumulator half
// of the register file ("a"), the hidden tiles (read by the gelu on the VALU) live in the architectural half ("v").
// Left to the builtin, hipcc picks one form for all MFMAs of the function and then rotates 16-register accumulator
// tiles through v_accvgpr copies and scratch around the loops -- differently after every edit.  What the compiler
// does not know about an asm MFMA (cdna_hip_programming.md 5.7): the wait states between its result and a VALU /
// accvgpr reader (mfma_done below), and between a VALU-written operand and the MFMA (the hidden fragments are
// packed at least one MFMA slot before their first use).
__device__ __forceinline__ void mfma_a(f32x16& c, u32x4 a, u32x4 b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_v(f32x16& c, u32x4 a, u32x4 b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_v_ab(f32x16& c, u32x4 a, u32x4 b) {  // B operand from the accumulator half of the file
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma_v0(f32x16& c, u32x4 a, u32x4 b) {  // c = a b (no accumulator input)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b));
}
// after the last MFMA into c, before anything but an MFMA reads it (8-pass MFMA: 12 wait states; 18 given)
__device__ __forceinline__ void mfma_done_v(f32x16& c) { asm volatile("s_nop 15\n\ts_nop 1" : "+v"(c)); }
__device__ __forceinline__ void mfma_done_a(f32x16 (&c)[16]) {
    asm volatile("s_nop 15\n\ts_nop 1"
                 : "+a"(c[0]), "+a"(c[1]), "+a"(c[2]), "+a"(c[3]), "+a"(c[4]), "+a"(c[5]), "+a"(c[6]), "+a"(c[7]), "+a"(c[8]), "+a"(c[9]),
                   "+a"(c[10]), "+a"(c[11]), "+a"(c[12]), "+a"(c[13]), "+a"(c[14]), "+a"(c[15]));
}

// makes the compiler forget the copies it holds of the accumulators: the next reader takes them from the accumulator
// registers again (one v_accvgpr_read each) instead of keeping 256 values alive through scratch between two passes
__device__ __forceinline__ void acc_touch(f32x16 (&c)[16]) {
    asm volatile(""
                 : "+a"(c[0]), "+a"(c[1]), "+a"(c[2]), "+a"(c[3]), "+a"(c[4]), "+a"(c[5]), "+a"(c[6]), "+a"(c[7]), "+a"(c[8]), "+a"(c[9]),
                   "+a"(c[10]), "+a"(c[11]), "+a"(c[12]), "+a"(c[13]), "+a"(c[14]), "+a"(c[15]));
}

// DBG (timing experiments): 1 = no DMA pieces, 2 = no gelu, 3 = clocks per FFN phase kind into p.stamps[8..11],
// 4 = VALU slice in a region of its own behind its MFMA, 5 = gelu of every second value only, 6 = no s_barrier in the
// per-stage sync, 7 = weight fragments read once (no LDS reads in the loops).  2, 5, 6, 7 compute wrong results.  p.stamps: phase stamps (shader clocks) of one workgroup
// TAIL: what follows the FFN in the stream -- 1: the next layer's in_proj rows (p.QKVout), 2: the two scalar output heads (p.head_out)
// XB (round 6): the residual rows p.res and the output rows p.Xout are bf16 (BlockP::x_bf16).  A 128-byte line then holds the 64
// features of TWO feature tiles, so the residual arrives and X'' leaves a tile PAIR at a time: half the LDS-DMA pieces, half the
// stores, half the bytes of both HBM bursts of a tile; everything between (X' in the accumulators, both LayerNorms) stays fp32.
template <int DBG, int TAIL, int XB = 0>
__global__ __launch_bounds__(256, 1) void block_fused_kernel(BlockP p) {
    constexpr bool QKV = TAIL == 1, HEADS = TAIL == 2, SPLIT = TAIL == 3;
    static_assert(!XB || TAIL == 0 || TAIL == 1, "bf16 residual rows: the plain and the next-Q|K|V forms only");
    // SPLIT (few tiles: a launch would leave most CUs idle for a whole tile): blockIdx.y = which quarter of the FFN's hidden
    // chunks this workgroup takes; every quarter repeats the out-proj and LayerNorm-2 (11 % of a tile), adds its share of
    // FFN2 to zero (quarter 0: to X' + b2) and stores the fp32 partial to its slab; block_split_reduce sums the slabs in order
    constexpr int NCHL = SPLIT ? NCH / SPLIT_N : NCH;  // hidden chunks of this workgroup
    static_assert((NCHL - 2) % 3 == 0 && (NCHL - 1) % 3 == 1, "the FFN loop runs three chunks at a time between the first and the last");
    constexpr int NST = QKV ? NRS_QKV : HEADS ? NRS_HEAD : SPLIT ? FR_OUT / RS_FR + 4 * NCHL : NRS;  // ring stages a workgroup consumes
    const int cbase = SPLIT ? (int)blockIdx.y * NCHL : 0;  // first hidden chunk
    __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int row0 = blockIdx.x * 128;
    // HEADS: workgroup b owns rows 128 (b >> 1) .. of row group hs = b & 1 -- the rows (r % out_mod) / out_grp == hs in their
    // order; gi = this lane's index among them (= its index in that head's output)
    const int hs = HEADS ? (int)(blockIdx.x & 1) : 0;
    const int gi = HEADS ? (int)(blockIdx.x >> 1) * 128 + 32 * wu + l31 : 0;
    const int gi_ld = HEADS ? (gi < p.M / 2 ? gi : p.M / 2 - 1) : 0;
    const int rtok = HEADS ? (gi_ld / p.out_grp) * p.out_mod + hs * p.out_grp + gi_ld % p.out_grp
                           : row0 + 32 * wu + l31;  // this lane's token row
    const bool valid = HEADS ? gi < p.M / 2 : rtok < p.M;
    const int rld = valid ? rtok : p.M - 1;         // (loads of the padding rows read the last row)
    const int lane16 = lane * 16;
    long long stamps[7];  // (scalar registers; stored at the very end, and only when p.stamps is set)
    long long psum[4] = {0, 0, 0, 0}, pt = 0;  // clocks spent in the four FFN phase kinds (A0, A1, B1, B2)
    (void)pt;
    stamps[0] = __builtin_readcyclecounter();

    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wstream, 0, (unsigned)((HEADS ? NRS + 2 * HEAD_FR / RS_FR : SPLIT ? NRS : NST) * RS_B), 0x00020000);
    (void)w_rs;
    // piece pc (0..7) of stage st: fragment wu + 4 pc of that stage -> the same position of ring slot `slot` = st % 3
    auto piece = [&](int st, int slot, int pc) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass of hipcc does not know this builtin)
        if (DBG == 1) return;
        int sw = st >= NST ? st - NST : st;  // past the end: the head of the stream again (never read)
        if (HEADS && sw >= NRS) sw += hs * (HEAD_FR / RS_FR);  // (this workgroup's head)
        if (SPLIT && sw >= FR_OUT / RS_FR) {  // (this workgroup's hidden chunks: its first A0 and last B2 sit one phase away from the run between them)
            const int n = sw - FR_OUT / RS_FR;
            sw += 4 * cbase - (n == 0 && cbase > 0) + (n == 4 * NCHL - 1 && cbase + NCHL < NCH);
        }
        const int fo = (wu + 4 * pc) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lptr_t)(smem + slot * RS_B + fo), 16, lane16, sw * RS_B + fo, 0, 0);
#endif
    };
    // slots 0, 1 and slot 2 are read through two base addresses so that every offset fits the 16-bit immediate of ds_read
    const char* const lbase0 = smem + lane16;
    const char* const lbase2 = smem + 2 * RS_B + lane16;
    auto frag = [&](int slot, int f) -> u32x4 { return *(const u32x4*)((slot == 2 ? lbase2 : lbase0 + slot * RS_B) + f * 1024); };
    // LayerNorm-2 fragment of k-step 24 + m of this wave
    char* const abase = smem + ACT_OFF + wu * ACT_LDS * 1024 + lane16;
    auto afrag = [&](int m) -> u32x4 { return *(const u32x4*)(abase + m * 1024); };
    float* const tab = (float*)(smem + TAB_OFF);
    // this lane's view of the tables (+ 4 lh), opaque to the compiler: every table read is then ONE base register + an
    // immediate offset (left to itself it builds a separate address register per read and spills them between the passes)
    typedef const float __attribute__((address_space(3))) * lds_cf32_t;  // (an LDS pointer: laundered as a generic one the reads become flat loads)
    lds_cf32_t tabl = (lds_cf32_t)(tab + 4 * lh);
    asm volatile("" : "+v"(tabl));

    // ---- prologue: first stages in flight, parameter tables, accumulators = residual + out-proj bias
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) piece(s, s, pc);
    piece(2, 2, 0);
    {
        for (int i = tid; i < BFF / 4; i += 256) *(f32x4*)(tab + T_B1 + 4 * i) = *(const f32x4*)(p.b1 + 4 * i);
        if (tid < BD / 4) {
            const int i = 4 * tid;
            *(f32x4*)(tab + T_B2 + i) = *(const f32x4*)(p.b2 + i);
            *(f32x4*)(tab + T_BO + i) = *(const f32x4*)(p.bo + i);
            *(f32x4*)(tab + T_G2 + i) = *(const f32x4*)(p.ln2_g + i);
            *(f32x4*)(tab + T_BE2 + i) = *(const f32x4*)(p.ln2_b + i);
            if (p.Hout || QKV || HEADS) {
                *(f32x4*)(tab + T_GA + i) = *(const f32x4*)(p.lnA_g + i);
                *(f32x4*)(tab + T_BA + i) = *(const f32x4*)(p.lnA_b + i);
            }
            if (!QKV && (p.Hout || HEADS) && p.lnB_g[0]) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    *(f32x4*)(tab + T_GB + k * BD + i) = *(const f32x4*)(p.lnB_g[k] + i);
                    *(f32x4*)(tab + T_BB + k * BD + i) = *(const f32x4*)(p.lnB_b[k] + i);
                }
            }
        }
    }
    // O fragments (B operand of the out-proj): all 32 k-steps, needed from the first phase on.  The residual rows come tile
    // by tile: the out-proj runs feature tile by feature tile, tile jn's accumulator starts at (res + bo) of its 32 features,
    // and only the first three tiles' residual values are fetched here -- every tile of a launch starts at the same time, so
    // what the prologue waits for is an HBM burst of all CUs at once (~11 B/clk/CU): 176 KB per workgroup instead of 384.
    u32x4 ofr[KS];  // dead after the out-proj, the LayerNorm-2 fragments take their place
    {
        const bf16_t* const orow = p.O + (size_t)rld * p.ldo + 8 * lh;
#pragma unroll
        for (int s = 0; s < KS; ++s) ofr[s] = *(const u32x4*)(orow + 16 * s);  // (default policy: as nt loads these fragments -- rows the attention kernel has just written -- took 10 % off the tile, round 6)
    }
    // Residual staging: this wave's 8 KiB of the LayerNorm-2 fragment region are idle until the out-proj is over -- two
    // buffers of one feature tile (32 rows x 128 B) each, filled by LDS-DMA in whole-line pieces (8 rows x 128 B: lane
    // 8 r + c of piece pp fetches chunk c ^ r of row 8 pp + r, a source-side swizzle that keeps the read-back conflict-free)
    // and read back by the lane that owns the values: nothing is held in registers while the loads are in flight, and the
    // compiler never sees a load it would have to wait for.
    unsigned rsrc[4];  // piece pp: this lane's byte offset (row 8 pp + (lane >> 3) of the wave's 32, swizzled chunk)
    {
        const int rr8 = lane >> 3, cc = lane & 7;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const int il = 8 * pp + rr8;  // row of the wave's 32
            int rt;
            if (HEADS) {
                int g2 = (int)(blockIdx.x >> 1) * 128 + 32 * wu + il;
                if (g2 >= p.M / 2) g2 = p.M / 2 - 1;
                rt = (g2 / p.out_grp) * p.out_mod + hs * p.out_grp + g2 % p.out_grp;
            } else {
                rt = row0 + 32 * wu + il;
                if (rt >= p.M) rt = p.M - 1;
            }
            int rs = rt;
            if (p.res_L > 0) {  // shared leading rows of a sequence: read from sequence 0
                const int jj = rt % p.res_L;
                if (jj < p.res_nshared) rs = jj;
            }
            size_t rowoff;
            if (p.rowtab) {
                const int w = rt % p.rt_mod;
                rowoff = (size_t)(w < p.res_nu ? p.rt_mod + (rt / p.rt_mod) * p.res_nu + w : w) * BD;
            } else {
                rowoff = (size_t)rs * p.ldr;
            }
            rsrc[pp] = XB ? (unsigned)(rowoff * 2 + 16 * (cc ^ rr8)) : (unsigned)((rowoff + 4 * (cc ^ rr8)) * 4);
        }
    }
    char* const rstage = smem + ACT_OFF + wu * ACT_LDS * 1024;
    // (the buffer form, as the weight pieces: hipcc orders LDS reads behind a global_load_lds it cannot tell apart with a
    // full vmcnt(0), which would drain the weight stream every phase)
    const __amdgpu_buffer_rsrc_t r_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.rowtab ? p.rowtab : p.res), 0, 0xfffffff0u, 0x00020000);
    (void)r_rs;
    auto rdma_piece = [&](int t, int pp) {  // piece pp (8 rows) of the residual values of feature tile t (XB: of the tile pair 2 t, 2 t + 1: 128 bytes of bf16) -> buffer t & 1
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rs, (lptr_t)(rstage + (t & 1) * 4096 + pp * 1024), 16, rsrc[pp], 128 * t, 0, M3PC_STREAM_AUX);
#endif
    };
    auto rdma = [&](int t) {
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) rdma_piece(t, pp);
    };
    rdma(0);
    rdma(1);
    // this lane's read-back position of chunk 2 q + lh of its row l31: piece l31 >> 3, lane 8 (l31 & 7) + (chunk ^ (l31 & 7))
    typedef const char __attribute__((address_space(3))) * lds_cc_t;  // (an LDS pointer: laundered as a generic one the reads become flat loads)
    lds_cc_t rback = (lds_cc_t)(rstage + (l31 >> 3) * 1024 + (l31 & 7) * 128);
    asm volatile("" : "+v"(rback));
    f32x16 acc[NT];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    stamps[1] = __builtin_readcyclecounter();

    // weight fragment groups, read TWO groups (eight MFMAs) ahead: group g of phase ph lives in R[(2 ph + g) % 3] -- 8 groups
    // per phase, so the index of a phase's first group is 2 ph mod 3 = (2 SL) % 3, static like the ring slot
    u32x4 R[3][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        R[0][k] = frag(0, k);
        R[1][k] = frag(0, 4 + k);
    }

    // one phase = one ring stage = 8 groups of {reads of the next group, one DMA piece, 4 x (MFMA, VALU slice)}
    //   ph: phase index (runtime); SL: its ring slot ph % 3 (static)
    //   extra(gn): further LDS reads for group gn of this phase (gn = 8: group 0 of the next phase)
    //   mma(i, a, g): MFMA i (0..31) with weight fragment a, group g;  valu(g, k): VALU work beside MFMA k of group g
    // Every MFMA slot is its own scheduling region, so the VALU slices stay between the MFMAs they are written beside.
    // NRES: vector-memory operations issued behind the previous phase's last piece that may stay in flight across the sync
    // besides the 7 pieces of the stage after next (the 4 loads of a residual tile issued at the end of the previous phase)
    auto phase_n = [&](int ph, auto sl_c, auto nres_c, auto&& extra, auto&& mma, auto&& valu) {
        constexpr int SL = decltype(sl_c)::value;
        constexpr int RB = (2 * SL) % 3;
        constexpr int NRES = decltype(nres_c)::value;
        static_assert(NRES == 0 || NRES == 4, "sync counts");
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g == 6) {  // every read of this stage is issued (two groups ahead): sync, then on into the next stage's slot
                if (DBG == 6) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
                else if (NRES == 4) asm volatile("s_waitcnt vmcnt(11) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            if (DBG != 7) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    R[(RB + g + 2) % 3][k] = g < 6 ? frag(SL, 4 * (g + 2) + k) : frag((SL + 1) % 3, 4 * (g - 6) + k);
            }
            extra(g + 1);
            // 8 DMA pieces per stage: pieces 1..7 of stage ph+2, piece 0 of stage ph+3 behind the sync
            if (g < 7) piece(ph + 2, (SL + 2) % 3, 1 + g);
            else piece(ph + 3, SL, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mma(4 * g + k, R[DBG == 7 ? 0 : (RB + g) % 3][k], g);
                if (DBG == 4) __builtin_amdgcn_sched_barrier(0);
                valu(g, k);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto phase = [&](int ph, auto sl_c, auto&& extra, auto&& mma, auto&& valu) {
        phase_n(ph, sl_c, std::integral_constant<int, 0>{}, extra, mma, valu);
    };
    auto no_valu = [](int, int) {};
    auto no_extra = [](int) {};
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;

    // ---- out-proj: phase jn = the 32 k-steps of feature tile jn.  Its accumulator starts at the residual + bias of its 32
    // features, read back from the staging buffer its pieces were sent to two phases ago.
    // Tile jn's start values in two parts -- the read-back of its residual values and bias, then a quarter at a time the sum into
    // the accumulator -- so that they ride in the last MFMA slots of phase jn - 1 (round 6; in front of the phase they cost ~0.4 k of
    // its 1.75 k clocks with no MFMA in flight).  The residual of tile jn is in LDS by then: the stage sync of phase jn - 1 lets only
    // the 11 (7) youngest vector-memory operations stay in flight, and tile jn's pieces are older than those.
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    f32x4 ai_x[4], ai_b[4];
    u32x2 ai_v[4];
    (void)ai_x;
    (void)ai_v;
    auto acc_init_reads = [&](int jn) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (XB)  // features 32 jn + 8 q + 4 lh .. of the pair's line: chunk 4 (jn & 1) + q, its half lh
                ai_v[q] = *(const u32x2 __attribute__((address_space(3)))*)(rback + ((jn >> 1) & 1) * 4096 + (((4 * (jn & 1) + q) ^ (l31 & 7)) << 4) + 8 * lh);
            else
                ai_x[q] = *(const f32x4 __attribute__((address_space(3)))*)(rback + (jn & 1) * 4096 + (((2 * q + lh) ^ (l31 & 7)) << 4));
            ai_b[q] = *(const f32x4 __attribute__((address_space(3)))*)(tabl + T_BO + 32 * jn + 8 * q);
        }
    };
    auto acc_init_quarter = [&](int jn, int q) {
        f32x4 x;
        if constexpr (XB) {
            x[0] = __builtin_bit_cast(float, ai_v[q][0] << 16);
            x[1] = __builtin_bit_cast(float, ai_v[q][0] & 0xffff0000u);
            x[2] = __builtin_bit_cast(float, ai_v[q][1] << 16);
            x[3] = __builtin_bit_cast(float, ai_v[q][1] & 0xffff0000u);
        } else {
            x = ai_x[q];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[jn][4 * q + i] = x[i] + ai_b[q][i];
    };
    // LayerNorm-2's row sums ride under the out-proj too (round 6): tile jn - 1 is final when phase jn runs -- one element per
    // second MFMA slot, in the order the separate pass summed them (tile by tile, register by register: the same bits); only the
    // last tile's 16 elements are left for behind the out-proj
    float s1o = 0.f, s2o = 0.f;
    auto ln2_stat = [&](float x) {
        s1o += x;
        s2o = fmaf(x, x, s2o);
    };
    // vmcnt bookkeeping (operations complete in issue order): the 4 pieces of residual tile jn + 2 are issued in the first four
    // MFMA groups of phase jn (their buffer was emptied by the read-back of tile jn, in phase jn - 1): they have two phases to land -- with one phase
    // (issued when phase jn + 1 starts) a tile alone on the chip measures the same, but beside the other streams' kernels every
    // phase waits for its residual tile and a pipelined step takes 2.2 ms instead of 1.25.
    // At the sync of phase jn the operations issued behind stage (jn + 1)'s last piece are one weight piece, these 4 and the
    // phase's first 6 pieces: vmcnt(11).  Wave-private data: no barrier.
    // XB: the prologue brought the pairs 0, 1 (tiles 0..3); pair P + 2 is sent for when phase 2 P + 1 starts -- its buffer's last
    // reader was the read-back of tile 2 P + 1, in phase 2 P -- and has three phases to land.
#define OUTPROJ(jn, SL, NR, RD)                                                                                              \
    phase_n(jn, SL{}, std::integral_constant<int, NR>{}, no_extra, [&](int i, u32x4 a, int) { mfma_a(acc[jn], a, ofr[i]); },         \
            [&](int g, int k) {                                                                                              \
                if ((RD) >= 0 && g < 4 && k == 3) rdma_piece((RD) >= 0 ? (RD) : 0, g);  /* (in front of the stage sync: counted there) */ \
                if ((jn) > 0 && (k & 1) == 0) ln2_stat(acc[(jn) > 0 ? (jn) - 1 : 0][2 * g + (k >> 1)]);                      \
                if ((jn) + 1 < NT) {                                                                                         \
                    constexpr int T1 = (jn) + 1 < NT ? (jn) + 1 : 0;                                                         \
                    if (g == 6 && k == 0) acc_init_reads(T1);                                                                \
                    if (g == 6 && k >= 2) acc_init_quarter(T1, k - 2);                                                       \
                    if (g == 7 && k < 2) acc_init_quarter(T1, 2 + k);                                                        \
                }                                                                                                            \
            });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (tile 0: the prologue's pieces -- waited for above already)
    acc_init_reads(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc_init_quarter(0, q);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (XB) {
        OUTPROJ(0, S0, 0, -1) OUTPROJ(1, S1, 4, 2) OUTPROJ(2, S2, 0, -1) OUTPROJ(3, S0, 4, 3)
        OUTPROJ(4, S1, 0, -1) OUTPROJ(5, S2, 4, 4) OUTPROJ(6, S0, 0, -1) OUTPROJ(7, S1, 4, 5)
        OUTPROJ(8, S2, 0, -1) OUTPROJ(9, S0, 4, 6) OUTPROJ(10, S1, 0, -1) OUTPROJ(11, S2, 4, 7)
        OUTPROJ(12, S0, 0, -1) OUTPROJ(13, S1, 0, -1) OUTPROJ(14, S2, 0, -1) OUTPROJ(15, S0, 0, -1)
    } else {
        OUTPROJ(0, S0, 4, 2) OUTPROJ(1, S1, 4, 3) OUTPROJ(2, S2, 4, 4) OUTPROJ(3, S0, 4, 5) OUTPROJ(4, S1, 4, 6)
        OUTPROJ(5, S2, 4, 7) OUTPROJ(6, S0, 4, 8) OUTPROJ(7, S1, 4, 9) OUTPROJ(8, S2, 4, 10) OUTPROJ(9, S0, 4, 11)
        OUTPROJ(10, S1, 4, 12) OUTPROJ(11, S2, 4, 13) OUTPROJ(12, S0, 4, 14) OUTPROJ(13, S1, 4, 15)
        OUTPROJ(14, S2, 0, -1) OUTPROJ(15, S0, 0, -1)
    }
#undef OUTPROJ
    mfma_done_a(acc);  // (the accumulators are next read by v_accvgpr_read)
    stamps[2] = __builtin_readcyclecounter();

    // ---- LayerNorm-2 of X' (in the accumulators) -> act (bf16 B-operand fragments of FFN1)
    u32x4 act[KS - ACT_LDS];  // k-steps 0..23; k-steps 24..31 go to this wave's LDS region
    {
        float s1 = s1o, s2 = s2o;  // (tiles 0..14: summed under the out-proj)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float x = acc[NT - 1][e];
            s1 += x;
            s2 = fmaf(x, x, s2);
        }
        __builtin_amdgcn_sched_barrier(0);
        s1 = half_swap_sum(s1);
        s2 = half_swap_sum(s2);
        const float mean = s1 * (1.0f / BD);
        float rstd = rsqrtf(fmaxf(s2 * (1.0f / BD) - mean * mean, 0.f) + 1e-5f);
        float nmr = -mean * rstd;
        acc_touch(acc);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            // (volatile asm statements keep their order: the table reads of fragment s stay behind this one and its
            // arithmetic in front of the one that closes the iteration; left alone, the compiler issues the table reads of
            // all fragments first and spills them)
            asm volatile("" : "+v"(rstd), "+v"(nmr) : : "memory");
            __builtin_amdgcn_sched_barrier(0);
            const int jn = s >> 1;
            float y[8];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int q = 2 * (s & 1) + k, n = 32 * jn + 8 * q;
                const f32x4 g = *(const f32x4 __attribute__((address_space(3)))*)(tabl + T_G2 + n);
                const f32x4 b = *(const f32x4 __attribute__((address_space(3)))*)(tabl + T_BE2 + n);
#pragma unroll
                for (int i = 0; i < 4; ++i) y[4 * k + i] = fmaf(fmaf(acc[jn][4 * q + i], rstd, nmr), g[i], b[i]);
            }
            bf16x8 w;
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = (bf16_t)y[j];
            if (s < KS - ACT_LDS) {
                act[s] = __builtin_bit_cast(u32x4, w);
                asm volatile("" : "+v"(act[s]));
            } else {
                *(u32x4*)(abase + (s - (KS - ACT_LDS)) * 1024) = __builtin_bit_cast(u32x4, w);
            }
        }
    }
    stamps[3] = __builtin_readcyclecounter();

    // ---- FFN
    f32x16 h0, h1;   // hidden accumulators of the chunk
    u32x4 hb[4];     // its four bf16 k-step fragments (B operand of FFN2)
    auto bias_quarter = [&](f32x16& hh, int c, int t, int q) {  // registers 4 q.. of a hidden accumulator := linear1 bias of their units
        const f32x4 b = *(const f32x4 __attribute__((address_space(3)))*)(tabl + T_B1 + 64 * c + 32 * t + 8 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) hh[4 * q + i] = b[i];
    };
    auto bias_init = [&](f32x16& hh, int c, int t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bias_quarter(hh, c, t, q);
    };
    // gelu of register e of hidden tile t.  The wave's VALU issue bounds these phases (transcendentals
    // cost 1.8 ordinary instructions, packed fp32 arithmetic does not overlap the MFMAs), so the cheapest form that
    // stays below the bf16 rounding of the hidden activations is used:
    //     gelu(x) = x / (1 + exp(-s(x))),  s(x) = x (c0 + c1 x^2 + c2 x^4),  x^2 clamped at 50
    // fitted (minimax over [-8, 8]) to x Phi(x): |d gelu| <= 2.6e-5 everywhere, where the exact-erf form of the GEMM
    // epilogues (gemm_epilogue.h, A&S 7.1.26) costs 13 instructions + 2 transcendentals per value against 7 + 2 here.
    // A value advances by a quarter per MFMA slot of its group (the transcendental of one slot is consumed in the next);
    // the last quarter of an odd value packs its pair into fragment 2 t + (e >> 3) of hb.
    float gx[2], gq[2], gs[2];
    auto gelu_val = [&](const f32x16& hh, int t, int e, int k) {
        constexpr float C0 = -2.3011212f, C1 = -0.10677572f, C2 = 0.001014263f;  // -log2(e) * (c0, c1, c2)
        const int j = e & 1;
        if (k == 0) {
            const float x = hh[e];
            gx[j] = x;
            gs[j] = fminf(x * x, 50.0f);
            gq[j] = fmaf(gs[j], C2, C1);
        } else if (k == 1) {
            gq[j] = fmaf(gs[j], gq[j], C0);
            gq[j] = __builtin_amdgcn_exp2f(gx[j] * gq[j]);
        } else if (k == 2) {
            gq[j] = __builtin_amdgcn_rcpf(gq[j] + 1.0f);
        } else {
            gx[j] = DBG == 2 ? gx[j] : gx[j] * gq[j];
            if (j == 1) {
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                bf16x2 w;
                w[0] = (bf16_t)gx[0];
                w[1] = (bf16_t)gx[1];
                hb[2 * t + (e >> 3)][(e & 7) >> 1] = __builtin_bit_cast(unsigned, w);
            }
        }
    };
    // the whole tile beside one phase (two values per group) | half hf of it (one value per group)
    auto gelu_full = [&](const f32x16& hh, int t, int g, int k) {
        if (DBG == 5 && g % 2) return;
        gelu_val(hh, t, 2 * g, k);
        gelu_val(hh, t, 2 * g + 1, k);
    };
    auto gelu_half = [&](const f32x16& hh, int t, int hf, int g, int k) { gelu_val(hh, t, 8 * hf + g, k); };
    // B operand of MFMA i of an FFN1 phase (group g = i / 4): registers for the first three of a group, LDS for the fourth
    u32x4 R2[2];  // the LDS-resident LayerNorm-2 fragment of an FFN1 group, by group parity
    auto a_operand = [&](int i, int g) -> u32x4 { return i % 4 < 3 ? act[a_kstep(i)] : R2[g & 1]; };
    auto act_reads = [&](int gn) { R2[gn & 1] = afrag(gn & 7); };  // (gn = 8: group 0 of the next FFN1 phase)
    if (SPLIT && blockIdx.y != 0) {  // the other quarters add their share of FFN2 to zero (X' stays with quarter 0)
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[jn][e] = 0.f;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    bias_init(h0, cbase, 0);
    bias_init(h1, cbase, 1);
    act_reads(0);
    // The four phases of a chunk are interleaved with its neighbours' so that every phase carries half a tile of gelu (the
    // wave's instruction issue bounds a phase that carries a whole one: 56 against 41 clocks per MFMA):
    //     A0(0) A1(0) B1(0) | A0(c) B2(c-1) A1(c) B1(c), c = 1.. | B2(last)
    // gelu of tile 0 of chunk c: first half beside B2(c-1), second beside A1(c); of tile 1: beside B1(c) and A0(c+1).  The
    // linear1 bias of the next use goes into a hidden accumulator in the phase after its gelu is over (h0: B1, h1: B2).
    // FFN2's k order (chunk by chunk, k-steps 0..3) is unchanged, and so is every bit of the result.
    auto stamp_kind = [&](int kind) {
        if (DBG == 3) { const long long t = __builtin_readcyclecounter(); psum[kind] += t - pt; pt = t; }
    };
    auto a_extra = [&](int gn) { if (gn < 8) act_reads(gn); };         // (a B phase follows)
    auto b_extra = [&](int gn) { if (gn == 8) act_reads(8); };         // (an A phase follows)
    if (DBG == 3) pt = __builtin_readcyclecounter();
    // chunk 0: A0 A1 B1 (ring slots 1 2 0)
    phase(16, S1{}, act_reads, [&](int i, u32x4 a, int g) { mfma_v(h0, a, a_operand(i, g)); }, no_valu);
    mfma_done_v(h0);
    stamp_kind(0);
    phase(17, S2{}, a_extra, [&](int i, u32x4 a, int g) { mfma_v(h1, a, a_operand(i, g)); },
          [&](int g, int k) { gelu_full(h0, 0, g, k); });
    mfma_done_v(h1);
    stamp_kind(1);
    // B1(c): hidden k-steps 0, 1; gelu of tile 1 (first half; all of it in the last chunk, whose B2 follows directly)
    auto phase_b1 = [&](int c, auto sl_c, auto last_c) {
        constexpr bool LAST = decltype(last_c)::value;
        phase(16 + 4 * c + 2, sl_c, b_extra, [&](int i, u32x4 a, int) { mfma_a(acc[i % 16], a, hb[i / 16]); },
              [&](int g, int k) {
                  if (LAST) {
                      gelu_full(h1, 1, g, k);
                  } else {
                      gelu_half(h1, 1, 0, g, k);
                      if (g % 2 == 1 && k == 0) bias_quarter(h0, c + 1 + cbase, 0, g >> 1);
                  }
              });
        stamp_kind(2);
    };
    phase_b1(0, S0{}, std::false_type{});
    auto rnd4 = [&](int c, auto s0_c, auto last_c) {  // A0(c) B2(c-1) A1(c) B1(c), starting in ring slot SL0
        constexpr int SL0 = decltype(s0_c)::value;
        using P0 = std::integral_constant<int, SL0 % 3>;
        using P1 = std::integral_constant<int, (SL0 + 1) % 3>;
        using P2 = std::integral_constant<int, (SL0 + 2) % 3>;
        const int ph = 16 + 4 * c - 1;
        phase(ph, P0{}, a_extra, [&](int i, u32x4 a, int g) { mfma_v(h0, a, a_operand(i, g)); },
              [&](int g, int k) { gelu_half(h1, 1, 1, g, k); });
        mfma_done_v(h0);
        stamp_kind(0);
        phase(ph + 1, P1{}, b_extra, [&](int i, u32x4 a, int) { mfma_a(acc[i % 16], a, hb[2 + i / 16]); },
              [&](int g, int k) {
                  gelu_half(h0, 0, 0, g, k);
                  if (g % 2 == 1 && k == 0) bias_quarter(h1, c + cbase, 1, g >> 1);
              });
        stamp_kind(3);
        phase(ph + 2, P2{}, a_extra, [&](int i, u32x4 a, int g) { mfma_v(h1, a, a_operand(i, g)); },
              [&](int g, int k) { gelu_half(h0, 0, 1, g, k); });
        mfma_done_v(h1);
        stamp_kind(1);
        phase_b1(c, P0{}, last_c);
    };
    // round c starts in slot (16 + 4 c - 1) % 3 = c % 3
    for (int c = 1; c < NCHL - 1; c += 3) {  // rounds 1..30 = ten times three
        rnd4(c, S1{}, std::false_type{});
        rnd4(c + 1, S2{}, std::false_type{});
        rnd4(c + 2, S0{}, std::false_type{});
    }
    rnd4(NCHL - 1, S1{}, std::true_type{});
    // B2 of the last chunk (slot 2)
    phase(16 + 4 * NCHL - 1, S2{}, no_extra, [&](int i, u32x4 a, int) { mfma_a(acc[i % 16], a, hb[2 + i / 16]); }, no_valu);
    stamp_kind(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the pieces issued past the end of the stream
    mfma_done_a(acc);
    stamps[4] = __builtin_readcyclecounter();

    // ---- epilogue: X'' = acc + b2 (fp32, optional) and the LayerNorm(s) of it (bf16)
    if (!SPLIT || blockIdx.y == 0) {
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = *(const f32x4 __attribute__((address_space(3)))*)(tabl + T_B2 + 32 * jn + 8 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[jn][4 * q + i] += b[i];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    acc_touch(acc);
    if constexpr (!HEADS) if (p.Xout) {  // (SPLIT: this quarter's slab of M rows; the head variant never stores X'')
        // Through the wave's staging buffers (idle since the FFN's last FFN1 phase) so that a store instruction writes 8 rows x
        // one whole 128-byte line: a lane owns 16 bytes of each of its row's lines, and stored from the registers every
        // instruction touched 32 lines for 1 KiB -- 8192 line touches per wave, which is what the X'' store took (23 k clocks
        // per tile, the texture path's one line per clock; staged: 512).  Same swizzle as the residual tiles' way in.
        const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)p.Xout, 0, (unsigned)((unsigned long long)(SPLIT ? SPLIT_N : 1) * p.M * p.ldx * (XB ? 2 : 4)), 0x00020000);
        (void)x_rs;
        const int r8 = lane >> 3, cc = lane & 7;
        unsigned xoff[4];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const int rr = row0 + 32 * wu + 8 * pp + r8;
            if constexpr (XB) xoff[pp] = rr < p.M ? (unsigned)rr * (unsigned)(p.ldx * 2) + (unsigned)((cc ^ r8) << 4) : 0x80000000u;
            else xoff[pp] = rr < p.M ? (unsigned)(((SPLIT ? (size_t)blockIdx.y * p.M : 0) + (size_t)rr) * p.ldx * 4 + ((cc ^ r8) << 4)) : 0x80000000u;
        }
        typedef char __attribute__((address_space(3))) * lds_c_t;
        lds_c_t xwr = (lds_c_t)(rstage + l31 * 128);
        lds_c_t xrd = (lds_c_t)(rstage + r8 * 128 + cc * 16);
        asm volatile("" : "+v"(xwr), "+v"(xrd));
        u32x4 xs[2][4];
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (XB) {
            // bf16 rows: the staging buffer holds a tile PAIR (32 rows x 128 B = 64 features); tile jn's quarter q is chunk
            // 4 (jn & 1) + q of its row's line, half lh of it; a finished pair leaves as four 1-KiB stores of 8 rows x one line
#pragma unroll
            for (int jn = 0; jn < NT; ++jn) {
                const int P = jn >> 1;
                // (the conversion reads VGPRs -- unlike the fp32 rows' ds_write_b128, which takes the accumulator registers as they
                // are -- and left alone hipcc copies all 256 accumulators out in front of the loop and spills around it: the copy
                // of a tile may not move above this statement)
                asm volatile("" : "+a"(acc[jn]));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    bf16x4 w;
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[i] = (bf16_t)acc[jn][4 * q + i];
                    *(u32x2 __attribute__((address_space(3)))*)(xwr + (P & 1) * 4096 + (((4 * (jn & 1) + q) ^ (l31 & 7)) << 4) + 8 * lh) = __builtin_bit_cast(u32x2, w);
                }
                if (jn & 1) {  // (read back and stored at once: one register set -- with two, hipcc spilled an accumulator tile around the stores)
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp) xs[0][pp] = *(const u32x4 __attribute__((address_space(3)))*)(xrd + (P & 1) * 4096 + pp * 1024);
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp) __builtin_amdgcn_raw_buffer_store_b128(xs[0][pp], x_rs, xoff[pp], P * 128, M3PC_STREAM_AUX);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 x;
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = acc[jn][4 * q + i];
                *(f32x4 __attribute__((address_space(3)))*)(xwr + (jn & 1) * 4096 + (((2 * q + lh) ^ (l31 & 7)) << 4)) = x;
            }
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) xs[jn & 1][pp] = *(const u32x4 __attribute__((address_space(3)))*)(xrd + (jn & 1) * 4096 + pp * 1024);
            if (jn > 0) {
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) __builtin_amdgcn_raw_buffer_store_b128(xs[(jn - 1) & 1][pp], x_rs, xoff[pp], (jn - 1) * 128, M3PC_STREAM_AUX);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) __builtin_amdgcn_raw_buffer_store_b128(xs[(NT - 1) & 1][pp], x_rs, xoff[pp], (NT - 1) * 128, M3PC_STREAM_AUX);
        }
#endif
    }
    stamps[5] = __builtin_readcyclecounter();
    acc_touch(acc);
    auto row_stats = [&](float& rstd, float& nmr) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int jn = 0; jn < NT; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float x = acc[jn][e];
                s1 += x;
                s2 = fmaf(x, x, s2);
                if (e == 15) __builtin_amdgcn_sched_barrier(0);
            }
        s1 = half_swap_sum(s1);
        s2 = half_swap_sum(s2);
        const float mean = s1 * (1.0f / BD);
        rstd = rsqrtf(fmaxf(s2 * (1.0f / BD) - mean * mean, 0.f) + 1e-5f);
        nmr = -mean * rstd;
    };
    if constexpr (QKV) {
        // ---- the next layer's Q|K|V projection: LN_A(X'') as bf16 fragments (the accumulators are free after that), then
        // three times 16 phases with the operands SWAPPED (activations = A operand, weights = B operand: the register images
        // are the same), so that an accumulator holds  lane & 31 = feature, registers = token rows  and one store instruction
        // writes 64 contiguous bytes of two rows (kv_fused_kernel below has the same scheme)
        float rstd, nmr;
        row_stats(rstd, nmr);
        acc_touch(acc);
        u32x4 qa[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            asm volatile("" : "+v"(rstd), "+v"(nmr) : : "memory");  // (see LayerNorm-2)
            __builtin_amdgcn_sched_barrier(0);
            const int jn = s >> 1;
            float y[8];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int q = 2 * (s & 1) + k, n = 32 * jn + 8 * q;
                const f32x4 g = *(const f32x4 __attribute__((address_space(3)))*)(tabl + T_GA + n);
                const f32x4 b = *(const f32x4 __attribute__((address_space(3)))*)(tabl + T_BA + n);
#pragma unroll
                for (int i = 0; i < 4; ++i) y[4 * k + i] = fmaf(fmaf(acc[jn][4 * q + i], rstd, nmr), g[i], b[i]);
            }
            bf16x8 w;
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = (bf16_t)y[j];
            qa[s] = __builtin_bit_cast(u32x4, w);
            asm volatile("" : "+a"(qa[s]));  // (the 32 fragments live where the feature accumulators were: B operands may)
        }
        const __amdgpu_buffer_rsrc_t q_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.QKVout, 0, p.qkv_bytes, 0x00020000);
        (void)q_rs;
        // Feature tile by feature tile (phase j = the 32 k-steps of tile j of the 48: one accumulator chain, as the out-proj),
        // two hidden-style accumulators in turn: while tile j accumulates, tile j - 1 is rounded to bf16 and written to this
        // wave's staging buffers (its 8 KiB of the LayerNorm-2 fragment region, idle since the FFN: 2 x {32 rows x 128 B},
        // 16-byte chunks swizzled by row & 7), the accumulator it leaves takes the bias of tile j + 1, and every second phase a
        // finished pair of tiles leaves as four 1-KiB stores -- 8 rows x one whole 128-byte line each -- issued right behind
        // the stage sync, so that they have two phases to complete before a sync has to wait for them (vector-memory
        // operations complete in issue order: the sync of the following phase counts them, vmcnt(11)).  Round 3 ran three
        // times 16 phases over all 16 tiles of Q, of K, of V with the operands swapped, stored each third with 256 two-byte
        // stores per lane and drained them (vmcnt(0)) before the next: 95-104 k clocks per tile for 1536 MFMAs.
        for (int i = tid; i < 3 * BD / 4; i += 256) *(f32x4*)(tab + T_B1 + 4 * i) = *(const f32x4*)(p.bqkv + 4 * i);  // (the linear1 bias table is dead)
        unsigned srow[4];  // store pp: byte offset of row 8 pp + (lane >> 3) of the wave's 32 + this lane's (un-swizzled) chunk
        {
            const int r8 = lane >> 3, cc = lane & 7;
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                const int rr = row0 + 32 * wu + 8 * pp + r8;
                srow[pp] = rr < p.M ? (unsigned)rr * (unsigned)(p.ldq * 2) + (unsigned)((cc ^ r8) * 16) : 0x80000000u;
            }
        }
        typedef char __attribute__((address_space(3))) * lds_c_t;
        lds_c_t const swr = (lds_c_t)(rstage + l31 * 128 + lh * 8);                     // + buffer * 4096 + swizzled chunk * 16
        lds_c_t srd = (lds_c_t)(rstage + (lane >> 3) * 128 + (lane & 7) * 16);          // + buffer * 4096 + pp * 1024
        asm volatile("" : "+v"(srd));
        f32x16 hA, hB;
        u32x4 sreg[4];
        auto bias_q = [&](f32x16& hh, int j, int q) {  // registers 4 q.. := in_proj bias of features 32 j + 8 q + 4 lh ..
            const f32x4 bb = *(const f32x4 __attribute__((address_space(3)))*)(tabl + T_B1 + 32 * j + 8 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) hh[4 * q + i] = bb[i];
        };
        auto stage_q = [&](const f32x16& hh, int t, int q) {  // quarter q of finished tile t -> bf16 -> staging
            typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            bf16x4v w;
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = (bf16_t)hh[4 * q + i];
            const int c = (t & 1) * 4 + q;
            *(u32x2 __attribute__((address_space(3)))*)(swr + ((t >> 1) & 1) * 4096 + ((c ^ (l31 & 7)) << 4)) = __builtin_bit_cast(u32x2, w);
        };
        // the X'' stores are out (they share vmcnt with the DMA pieces); the stages 144, 145 landed before the FFN ended.
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (+ the bias table)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            R[0][k] = frag(0, k);
            R[1][k] = frag(0, 4 + k);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bias_q(hA, 0, q);
            bias_q(hB, 0, q);  // (phase 0 stages "tile -1" from it: into the half of a buffer that tile 3 overwrites before it is read)
        }
        // phase j: HC accumulates tile j; HP (tile j - 1) is staged and re-initialised; STORE: the pair (j - 2, j - 1) leaves
        auto qkv_phase = [&](int j, auto sl_c, auto nres_c, f32x16& HC, f32x16& HP, auto store_c) {
            constexpr bool STORE = decltype(store_c)::value;
            phase_n(144 + j, sl_c, nres_c, no_extra, [&](int i, u32x4 a, int) { mfma_v_ab(HC, a, qa[i]); },
                    [&](int g, int k) {
                        if (g == 0) stage_q(HP, j - 1, k);
                        if (g == 1) bias_q(HP, j + 1, k);
                        if (STORE && g == 3) sreg[k] = *(const u32x4 __attribute__((address_space(3)))*)(srd + (((j - 2) >> 1) & 1) * 4096 + k * 1024);
#if defined(__HIP_DEVICE_COMPILE__)
                        // (behind the sync and stage ph + 2's last piece -- slot (6, 0) shares a scheduling region with that piece,
                        // so the stores start one slot later; j = 0: nothing is staged yet, the stores go out of the buffer's range)
                        if (STORE && ((g == 6 && k > 0) || (g == 7 && k == 0))) {
                            const int pp = g == 6 ? k - 1 : 3;
                            __builtin_amdgcn_raw_buffer_store_b128(sreg[pp], q_rs, j >= 2 ? srow[pp] : 0x80000000u, ((j - 2) >> 1) * 128, M3PC_STREAM_AUX);
                        }
#endif
                    });
            mfma_done_v(HC);
        };
        using NR0 = std::integral_constant<int, 0>;
        using NR4 = std::integral_constant<int, 4>;
        psum[0] = __builtin_readcyclecounter();
        for (int j = 0; j < 3 * NT; j += 6) {  // (even phases store; the phase behind a storing one counts its four stores)
            qkv_phase(j, S0{}, NR0{}, hA, hB, std::true_type{});
            qkv_phase(j + 1, S1{}, NR4{}, hB, hA, std::false_type{});
            qkv_phase(j + 2, S2{}, NR0{}, hA, hB, std::true_type{});
            qkv_phase(j + 3, S0{}, NR4{}, hB, hA, std::false_type{});
            qkv_phase(j + 4, S1{}, NR0{}, hA, hB, std::true_type{});
            qkv_phase(j + 5, S2{}, NR4{}, hB, hA, std::false_type{});
        }
        psum[1] = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the pieces issued past the end of the stream
#pragma unroll
        for (int q = 0; q < 4; ++q) stage_q(hB, 3 * NT - 1, q);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32x4 v = *(const u32x4 __attribute__((address_space(3)))*)(srd + (((3 * NT - 2) >> 1) & 1) * 4096 + k * 1024);
            __builtin_amdgcn_raw_buffer_store_b128(v, q_rs, srow[k], ((3 * NT - 2) >> 1) * 128, M3PC_STREAM_AUX);
        }
#endif
    } else if constexpr (HEADS) {
        // ---- the output head of this workgroup's key: y = w2 . gelu(W1 LN_head(LN_A(X'')) + b1) + b2, de-tokenised.
        // W1 runs like an FFN1: 16 phases of one hidden tile (32 units over the 32 k-steps), accumulators h0 / h1 in turn;
        // gelu of tile j-1 and its share of the dot product with w2 run on the VALU beside the MFMAs of tile j.
        constexpr int T_HB1 = T_B1, T_HW2 = T_B1 + BD;  // this head's b1 | w2 in the (now dead) linear1-bias table
        if (tid < BD / 4) *(f32x4*)(tab + T_HB1 + 4 * tid) = *(const f32x4*)(p.hb1[hs] + 4 * tid);
        else *(f32x4*)(tab + T_HW2 + 4 * (tid - BD / 4)) = *(const f32x4*)(p.hw2[hs] + 4 * (tid - BD / 4));
        float rstd, nmr;
        row_stats(rstd, nmr);
        acc_touch(acc);
#pragma unroll
        for (int jn = 0; jn < NT; ++jn)  // acc := LN_A(acc) (decoder.norm)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q == 0) asm volatile("" : "+v"(rstd), "+v"(nmr) : : "memory");  // (see LayerNorm-2)
                const int n = 32 * jn + 8 * q;
                const f32x4 g = *(const f32x4 __attribute__((address_space(3)))*)(tabl + T_GA + n);
                const f32x4 b = *(const f32x4 __attribute__((address_space(3)))*)(tabl + T_BA + n);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[jn][4 * q + i] = fmaf(fmaf(acc[jn][4 * q + i], rstd, nmr), g[i], b[i]);
                if (q == 3) __builtin_amdgcn_sched_barrier(0);
            }
        acc_touch(acc);
        row_stats(rstd, nmr);
        acc_touch(acc);
        {   // the head's own LayerNorm -> bf16 B-operand fragments: k-steps 0..23 in act[], 24..31 in this wave's LDS region
            lds_cf32_t const gtab = tabl + T_GB + hs * BD;
            lds_cf32_t const btab = tabl + T_BB + hs * BD;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                asm volatile("" : "+v"(rstd), "+v"(nmr) : : "memory");
                __builtin_amdgcn_sched_barrier(0);
                const int jn = s >> 1;
                float y[8];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int q = 2 * (s & 1) + k, n = 32 * jn + 8 * q;
                    const f32x4 g = *(const f32x4 __attribute__((address_space(3)))*)(gtab + n);
                    const f32x4 b = *(const f32x4 __attribute__((address_space(3)))*)(btab + n);
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[4 * k + i] = fmaf(fmaf(acc[jn][4 * q + i], rstd, nmr), g[i], b[i]);
                }
                bf16x8 w;
#pragma unroll
                for (int j = 0; j < 8; ++j) w[j] = (bf16_t)y[j];
                if (s < KS - ACT_LDS) {
                    act[s] = __builtin_bit_cast(u32x4, w);
                    asm volatile("" : "+v"(act[s]));
                } else {
                    *(u32x4*)(abase + (s - (KS - ACT_LDS)) * 1024) = __builtin_bit_cast(u32x4, w);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the head tables are in place for every wave
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // fragment groups 0, 1 of stage 144 (slot 0): it landed before the FFN ended
            R[0][k] = frag(0, k);
            R[1][k] = frag(0, 4 + k);
        }
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 hbb[2], hww[2];  // b1 / w2 of the two hidden units a group's gelu slice works on, by group parity
        float dot = 0.f;
        // units of tile tv, group g: registers e = 2 g, 2 g + 1 <-> features 32 tv + 8 (g >> 1) + 4 lh + 2 (g & 1) (+ 1)
        auto hload = [&](int tv, int g) {
            const int o = 32 * tv + 8 * (g >> 1) + 2 * (g & 1);
            hbb[g & 1] = *(const f32x2 __attribute__((address_space(3)))*)(tabl + T_HB1 + o);
            hww[g & 1] = *(const f32x2 __attribute__((address_space(3)))*)(tabl + T_HW2 + o);
        };
        auto gelu_dot = [&](const f32x16& hh, int g, int k) {
            constexpr float C0 = -2.3011212f, C1 = -0.10677572f, C2 = 0.001014263f;  // (gelu_slice)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (k == 0) {
                    const float x = hh[2 * g + j] + hbb[g & 1][j];
                    gx[j] = x;
                    gs[j] = fminf(x * x, 50.0f);
                    gq[j] = fmaf(gs[j], C2, C1);
                } else if (k == 1) {
                    gq[j] = fmaf(gs[j], gq[j], C0);
                    gq[j] = __builtin_amdgcn_exp2f(gx[j] * gq[j]);
                } else if (k == 2) {
                    gq[j] = __builtin_amdgcn_rcpf(gq[j] + 1.0f);
                } else {
                    dot = fmaf(gx[j] * gq[j], hww[g & 1][j], dot);
                }
            }
        };
        act_reads(0);
        // phase of hidden tile j (static): ring stage 144 + j, slot j % 3, accumulator h0 / h1 by parity
#define HEAD_PH(j, SL, HC, HP)                                                                                          \
        phase(144 + (j), SL{},                                                                                          \
              [&](int gn) { act_reads(gn); if (gn < 8) { if ((j) > 0) hload((j) - 1, gn); } else hload((j), 0); },       \
              [&](int i, u32x4 a, int g) { if (i == 0) mfma_v0(HC, a, a_operand(i, g)); else mfma_v(HC, a, a_operand(i, g)); }, \
              [&](int g, int k) { if ((j) > 0) gelu_dot(HP, g, k); });                                                  \
        mfma_done_v(HC);
        HEAD_PH(0, S0, h0, h1) HEAD_PH(1, S1, h1, h0) HEAD_PH(2, S2, h0, h1) HEAD_PH(3, S0, h1, h0)
        HEAD_PH(4, S1, h0, h1) HEAD_PH(5, S2, h1, h0) HEAD_PH(6, S0, h0, h1) HEAD_PH(7, S1, h1, h0)
        HEAD_PH(8, S2, h0, h1) HEAD_PH(9, S0, h1, h0) HEAD_PH(10, S1, h0, h1) HEAD_PH(11, S2, h1, h0)
        HEAD_PH(12, S0, h0, h1) HEAD_PH(13, S1, h1, h0) HEAD_PH(14, S2, h0, h1) HEAD_PH(15, S0, h1, h0)
#undef HEAD_PH
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the pieces issued past the end of the stream
        // gelu + dot of the last tile (15, in h1): group 0's operands were fetched by the last phase
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g < 7) hload(15, g + 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) gelu_dot(h1, g, k);
        }
        dot = half_swap_sum(dot);
        float yv = dot + p.hb2[hs][0];
        if (p.hmean[hs]) yv = __fadd_rn(__fmul_rn(yv, p.hstd[hs][0]), p.hmean[hs][0]);  // de-tokenise (continuous.py:86-94)
        if (valid && lh == 0) p.head_out[hs][gi] = yv;
    } else if (p.Hout) {
        int orow_h = rtok, sel = 0;
        if (p.out_mod > 0) {  // two row groups per out_mod rows: group s rows go to the s-th compact block, LN_B[s] applies
            const int w = rtok % p.out_mod;
            sel = w / p.out_grp;
            orow_h = sel * (p.M / p.out_mod) * p.out_grp + (rtok / p.out_mod) * p.out_grp + w % p.out_grp;
        }
        float rstd, nmr;
        row_stats(rstd, nmr);
        acc_touch(acc);
        const bool two = p.lnB_g[0] != nullptr;
        if (two) {  // acc := LN_A(acc), then statistics of that
#pragma unroll
            for (int jn = 0; jn < NT; ++jn)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q == 0) asm volatile("" : "+v"(rstd), "+v"(nmr) : : "memory");  // (see LayerNorm-2)
                    const int n = 32 * jn + 8 * q;
                    const f32x4 g = *(const f32x4 __attribute__((address_space(3)))*)(tabl + T_GA + n);
                    const f32x4 b = *(const f32x4 __attribute__((address_space(3)))*)(tabl + T_BA + n);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[jn][4 * q + i] = fmaf(fmaf(acc[jn][4 * q + i], rstd, nmr), g[i], b[i]);
                    if (q == 3) __builtin_amdgcn_sched_barrier(0);
                }
            acc_touch(acc);
            row_stats(rstd, nmr);
            acc_touch(acc);
        }
        lds_cf32_t const gtab = two ? tabl + T_GB + sel * BD : tabl + T_GA;
        lds_cf32_t const btab = two ? tabl + T_BB + sel * BD : tabl + T_BA;
        bf16_t* hrow = p.Hout + (size_t)(valid ? orow_h : 0) * p.ldh;
#pragma unroll
        for (int jn = 0; jn < NT; ++jn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q == 0) asm volatile("" : "+v"(rstd), "+v"(nmr) : : "memory");
                const int n = 32 * jn + 8 * q;
                const f32x4 g = *(const f32x4 __attribute__((address_space(3)))*)(gtab + n);
                const f32x4 b = *(const f32x4 __attribute__((address_space(3)))*)(btab + n);
                bf16x4 w;
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = (bf16_t)fmaf(fmaf(acc[jn][4 * q + i], rstd, nmr), g[i], b[i]);
                if (valid) *(bf16x4*)(hrow + n + 4 * lh) = w;
                if (q == 3) __builtin_amdgcn_sched_barrier(0);
            }
    }
    stamps[6] = __builtin_readcyclecounter();
    if (p.stamps && (int)blockIdx.x == p.stamp_block && lane == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) p.stamps[wu * 16 + k] = stamps[k];
        if (DBG == 3 || QKV)  // (QKV: psum[0], [1] = the clock at the first and behind the last Q|K|V phase)
#pragma unroll
            for (int k = 0; k < 4; ++k) p.stamps[wu * 16 + 8 + k] = psum[k];
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ decoder input
// kv_fused_kernel: decoder embedding of the un-masked tokens -> norm1 -> K|V projection, the same machinery (transposed
// products, one wave = 32 token rows with the 512 embedding features in its accumulators, packed fragment stream
// through the three-slot ring).  Stream of a key: 512 embedding fragments (phase sb = k-steps 2 sb, 2 sb + 1 of the 16
// feature tiles, natural k: the operand is loaded from Z), then for K and for V 512 fragments in the same phase order
// (permuted k: the operand is the LayerNorm of an accumulator).  48 phases.  The K half is stored (bf16) before the V
// half starts: stores and the DMA pieces share vmcnt and may complete out of order with respect to each other, so the
// stream is drained once there (s_waitcnt vmcnt(0)) instead of counting pieces across the stores.
namespace {
constexpr int KV_FR = FR_OUT + 2 * FR_OUT;   // 1536
constexpr int KV_NRS = KV_FR / RS_FR;        // 48
constexpr int KT_G = 0, KT_B = KT_G + BD, KT_BKV = KT_B + BD, KT_END = KT_BKV + 2 * BD;
constexpr int KV_STAGE_OFF = NSLOT * RS_B;           // 4 waves x two 4-KiB staging buffers (table rows in, K|V tiles out)
constexpr int KV_TAB_OFF = KV_STAGE_OFF + 4 * 8192;
constexpr int KV_LDS_BYTES = KV_TAB_OFF + KT_END * 4;
static_assert(KV_LDS_BYTES <= 160 * 1024, "LDS");
}  // namespace

__global__ __launch_bounds__(256) void pack_kv_stream_kernel(const bf16_t* __restrict__ We, const bf16_t* __restrict__ Wkv,
                                                             bf16_t* __restrict__ out) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= KV_FR * 64) return;
    const int f = gid >> 6, lane = gid & 63, r = lane & 31, h = lane >> 5;
    bf16_t v[8];
    const int jt = f / KS, s = f % KS;  // phase jt = the 32 k-steps of one feature tile: 16 of the embedding, then the 32 of K | V
    if (f < FR_OUT) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = We[(size_t)(32 * jt + r) * BD + 16 * s + 8 * h + j];
    } else {
        const size_t row = (size_t)(32 * (jt - NT) + r) * BD;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = Wkv[row + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)];
    }
    bf16_t* o = out + (size_t)gid * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = v[j];
}

size_t kv_stream_bytes() { return (size_t)KV_FR * 1024; }
void launch_pack_kv_stream(const bf16_t* Wemb, const bf16_t* Wkv, bf16_t* out, hipStream_t st) {
    hipLaunchKernelGGL(pack_kv_stream_kernel, dim3(KV_FR * 64 / 256), dim3(256), 0, st, Wemb, Wkv, out);
}

namespace {

__global__ __launch_bounds__(256, 1) void kv_fused_kernel(KvFusedP p) {
    __shared__ __attribute__((aligned(1024))) char smem[KV_LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int tiles0 = (p.M[0] + 127) / 128;
    const int grp = (int)blockIdx.x >= tiles0 ? 1 : 0;
    const int tile = (int)blockIdx.x - grp * tiles0;
    const int Mg = p.M[grp];
    const RowMap map = p.map[grp];
    const int rtok = tile * 128 + 32 * wu + l31;   // this lane's row of the group
    const bool valid = rtok < Mg;
    const int rld = valid ? rtok : Mg - 1;
    const long long mrow = map.rpg ? (long long)(rld / map.rpg) * map.gstride + rld % map.rpg + map.off : rld;
    const int lane16 = lane * 16;
    long long stamps[6];
    stamps[0] = __builtin_readcyclecounter();

    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wstream[grp], 0, (unsigned)(KV_FR * 1024), 0x00020000);
    (void)w_rs;
    auto piece = [&](int st, int slot, int pc) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int sw = st >= KV_NRS ? st - KV_NRS : st;
        const int fo = (wu + 4 * pc) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lptr_t)(smem + slot * RS_B + fo), 16, lane16, sw * RS_B + fo, 0, 0);
#endif
    };
    const char* const lbase0 = smem + lane16;
    const char* const lbase2 = smem + 2 * RS_B + lane16;
    auto frag = [&](int slot, int f) -> u32x4 { return *(const u32x4*)((slot == 2 ? lbase2 : lbase0 + slot * RS_B) + f * 1024); };
    float* const tab = (float*)(smem + KV_TAB_OFF);
    typedef const float __attribute__((address_space(3))) * lds_cf32_t;
    lds_cf32_t tabl = (lds_cf32_t)(tab + 4 * lh);
    asm volatile("" : "+v"(tabl));

    // ---- prologue
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) piece(s, s, pc);
    piece(2, 2, 0);
    if (tid < BD / 4) {
        const int i = 4 * tid;
        *(f32x4*)(tab + KT_G + i) = *(const f32x4*)(p.ln_g + i);
        *(f32x4*)(tab + KT_B + i) = *(const f32x4*)(p.ln_b + i);
        *(f32x4*)(tab + KT_BKV + i) = *(const f32x4*)(p.bkv + i);
        *(f32x4*)(tab + KT_BKV + BD + i) = *(const f32x4*)(p.bkv + BD + i);
    }
    // Z fragments (B operand of the embedding; dead after it)
    u32x4 ofr[KS];
    {
        const bf16_t* const zrow = p.Z + (size_t)mrow * p.ldz + 8 * lh;
#pragma unroll
        for (int s = 0; s < KS; ++s) ofr[s] = *(const u32x4*)(zrow + 16 * s);
    }
    // The position-table rows arrive like the residual of the fused layer tail: the embedding runs feature tile by feature
    // tile, tile jn's accumulator starts at the table values of its 32 features, and those come by LDS-DMA in whole-line
    // pieces (8 rows x 128 B, source-side swizzle) into this wave's two 4-KiB staging buffers, two phases ahead.  (Round 3
    // read them per lane from L2 in the prologue: 16 bytes of 32 different lines per instruction, 13 k of the tile's 115 k clocks.)
    char* const rstage = smem + KV_STAGE_OFF + wu * 8192;
    unsigned rsrc[4];
    {
        const int rr8 = lane >> 3, cc = lane & 7;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const int rt = tile * 128 + 32 * wu + 8 * pp + rr8;
            const int rs = rt < Mg ? rt : Mg - 1;
            rsrc[pp] = (unsigned)(((size_t)(rs % p.rt_mod[grp]) * BD + 4 * (cc ^ rr8)) * 4);
        }
    }
    const __amdgpu_buffer_rsrc_t r_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.rowtab[grp], 0, (unsigned)((size_t)p.rt_mod[grp] * BD * 4), 0x00020000);
    (void)r_rs;
    auto rdma = [&](int t) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rs, (lptr_t)(rstage + (t & 1) * 4096 + pp * 1024), 16, rsrc[pp], 128 * t, 0, 0);
#endif
    };
    rdma(0);
    rdma(1);
    typedef const char __attribute__((address_space(3))) * lds_cc_t;
    lds_cc_t rback = (lds_cc_t)(rstage + (l31 >> 3) * 1024 + (l31 & 7) * 128);
    asm volatile("" : "+v"(rback));
    f32x16 acc[NT];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    stamps[1] = __builtin_readcyclecounter();

    // weight fragment groups read two groups ahead, one phase = one ring stage: see block_fused_kernel
    u32x4 R[3][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        R[0][k] = frag(0, k);
        R[1][k] = frag(0, 4 + k);
    }
    auto phase_n = [&](int ph, auto sl_c, auto nres_c, auto&& mma, auto&& valu) {
        constexpr int SL = decltype(sl_c)::value;
        constexpr int RB = (2 * SL) % 3;
        constexpr int NRES = decltype(nres_c)::value;
        static_assert(NRES == 0 || NRES == 4, "sync counts");
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g == 6) {
                if (NRES == 4) asm volatile("s_waitcnt vmcnt(11) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                R[(RB + g + 2) % 3][k] = g < 6 ? frag(SL, 4 * (g + 2) + k) : frag((SL + 1) % 3, 4 * (g - 6) + k);
            if (g < 7) piece(ph + 2, (SL + 2) % 3, 1 + g);
            else piece(ph + 3, SL, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mma(4 * g + k, R[(RB + g) % 3][k], g);
                valu(g, k);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto no_valu = [](int, int) {};
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using NR0 = std::integral_constant<int, 0>;
    using NR4 = std::integral_constant<int, 4>;

    // ---- embedding: phase jn = the 32 k-steps of feature tile jn (vmcnt bookkeeping as the out-proj of block_fused_kernel, and --
    // round 6 -- the same bookkeeping in the MFMA slots: tile jn + 1's start values are read back behind phase jn's stage sync and
    // written into the accumulator a quarter per slot, the four table pieces of tile jn + 2 go out one per MFMA group in front of
    // the sync, norm1's row sums of tile jn - 1 take every second slot, in the order the separate pass summed them)
    f32x4 ai_x[4];
    auto acc_init_reads = [&](int jn) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            ai_x[q] = *(const f32x4 __attribute__((address_space(3)))*)(rback + (jn & 1) * 4096 + (((2 * q + lh) ^ (l31 & 7)) << 4));
    };
    auto acc_init_quarter = [&](int jn, int q) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[jn][4 * q + i] = ai_x[q][i];
    };
    auto rdma_piece = [&](int t, int pp) {
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rs, (lptr_t)(rstage + (t & 1) * 4096 + pp * 1024), 16, rsrc[pp], 128 * t, 0, 0);
#endif
    };
    float s1o = 0.f, s2o = 0.f;
    auto ln_stat = [&](float x) {
        s1o += x;
        s2o = fmaf(x, x, s2o);
    };
#define KV_EMB(jn, SL, NR)                                                                                                   \
    phase_n(jn, SL{}, std::integral_constant<int, NR>{}, [&](int i, u32x4 a, int) { mfma_a(acc[jn], a, ofr[i]); },            \
            [&](int g, int k) {                                                                                              \
                if ((jn) + 2 < NT && g < 4 && k == 3) rdma_piece((jn) + 2 < NT ? (jn) + 2 : 0, g);                            \
                if ((jn) > 0 && (k & 1) == 0) ln_stat(acc[(jn) > 0 ? (jn) - 1 : 0][2 * g + (k >> 1)]);                       \
                if ((jn) + 1 < NT) {                                                                                         \
                    constexpr int T1 = (jn) + 1 < NT ? (jn) + 1 : 0;                                                         \
                    if (g == 6 && k == 0) acc_init_reads(T1);                                                                \
                    if (g == 6 && k >= 2) acc_init_quarter(T1, k - 2);                                                       \
                    if (g == 7 && k < 2) acc_init_quarter(T1, 2 + k);                                                        \
                }                                                                                                            \
            });
    acc_init_reads(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc_init_quarter(0, q);
    __builtin_amdgcn_sched_barrier(0);
    KV_EMB(0, S0, 4) KV_EMB(1, S1, 4) KV_EMB(2, S2, 4) KV_EMB(3, S0, 4) KV_EMB(4, S1, 4) KV_EMB(5, S2, 4)
    KV_EMB(6, S0, 4) KV_EMB(7, S1, 4) KV_EMB(8, S2, 4) KV_EMB(9, S0, 4) KV_EMB(10, S1, 4) KV_EMB(11, S2, 4)
    KV_EMB(12, S0, 4) KV_EMB(13, S1, 4) KV_EMB(14, S2, 0) KV_EMB(15, S0, 0)
#undef KV_EMB
    mfma_done_a(acc);
    stamps[2] = __builtin_readcyclecounter();

    // ---- norm1 of the embedded rows -> act (bf16 B-operand fragments, kept where the accumulators were)
    u32x4 act[KS];
    {
        float s1 = s1o, s2 = s2o;  // (tiles 0..14: summed under the embedding)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float x = acc[NT - 1][e];
            s1 += x;
            s2 = fmaf(x, x, s2);
        }
        __builtin_amdgcn_sched_barrier(0);
        s1 = half_swap_sum(s1);
        s2 = half_swap_sum(s2);
        const float mean = s1 * (1.0f / BD);
        float rstd = rsqrtf(fmaxf(s2 * (1.0f / BD) - mean * mean, 0.f) + 1e-5f);
        float nmr = -mean * rstd;
        acc_touch(acc);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            // (volatile asm statements keep their order: the table reads of fragment s stay behind this one, its arithmetic
            // in front of the one that closes the iteration -- left alone, the compiler issues all 64 reads first and spills them)
            asm volatile("" : "+v"(rstd), "+v"(nmr) : : "memory");
            __builtin_amdgcn_sched_barrier(0);
            const int jn = s >> 1;
            float y[8];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int q = 2 * (s & 1) + k, n = 32 * jn + 8 * q;
                const f32x4 g = *(const f32x4 __attribute__((address_space(3)))*)(tabl + KT_G + n);
                const f32x4 b = *(const f32x4 __attribute__((address_space(3)))*)(tabl + KT_B + n);
#pragma unroll
                for (int i = 0; i < 4; ++i) y[4 * k + i] = fmaf(fmaf(acc[jn][4 * q + i], rstd, nmr), g[i], b[i]);
            }
            bf16x8 w;
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = (bf16_t)y[j];
            act[s] = __builtin_bit_cast(u32x4, w);
            asm volatile("" : "+a"(act[s]));
        }
    }
    stamps[3] = __builtin_readcyclecounter();

    // ---- K | V: 32 feature tiles, one phase each (the next layer's Q|K|V part of block_fused_kernel: two accumulators in turn,
    // the finished tile staged as bf16 through the wave's buffers, a pair of tiles stored as whole 128-byte lines behind the
    // stage sync of every second phase).  Round 3 ran K and V as 16 phases over 16 accumulators each with the operands
    // swapped, stored 256 two-byte values per lane and half, and drained the stream between the halves.
    const __amdgpu_buffer_rsrc_t kv_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.KV, 0, p.kv_bytes, 0x00020000);
    (void)kv_rs;
    unsigned srow[4];
    {
        const int r8 = lane >> 3, cc = lane & 7;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const int rr = tile * 128 + 32 * wu + 8 * pp + r8;
            const long long mr = map.rpg ? (long long)(rr / map.rpg) * map.gstride + rr % map.rpg + map.off : rr;
            srow[pp] = rr < Mg ? (unsigned)(mr * p.ldkv * 2 + ((cc ^ r8) << 4)) : 0x80000000u;
        }
    }
    typedef char __attribute__((address_space(3))) * lds_c_t;
    lds_c_t const swr = (lds_c_t)(rstage + l31 * 128 + lh * 8);
    lds_c_t srd = (lds_c_t)(rstage + (lane >> 3) * 128 + (lane & 7) * 16);
    asm volatile("" : "+v"(srd));
    f32x16 hA, hB;
    u32x4 sreg[4];
    auto bias_q = [&](f32x16& hh, int j, int q) {
        const f32x4 bb = *(const f32x4 __attribute__((address_space(3)))*)(tabl + KT_BKV + 32 * j + 8 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) hh[4 * q + i] = bb[i];
    };
    auto stage_q = [&](const f32x16& hh, int t, int q) {
        typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        bf16x4v w;
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = (bf16_t)hh[4 * q + i];
        const int c = (t & 1) * 4 + q;
        *(u32x2 __attribute__((address_space(3)))*)(swr + ((t >> 1) & 1) * 4096 + ((c ^ (l31 & 7)) << 4)) = __builtin_bit_cast(u32x2, w);
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bias_q(hA, 0, q);
        bias_q(hB, 0, q);  // (phase 0 stages "tile -1" from it: into the half of a buffer that tile 3 overwrites before it is read)
    }
    auto kv_phase = [&](int j, auto sl_c, auto nres_c, f32x16& HC, f32x16& HP, auto store_c) {
        constexpr bool STORE = decltype(store_c)::value;
        phase_n(16 + j, sl_c, nres_c, [&](int i, u32x4 a, int) { mfma_v_ab(HC, a, act[i]); },
                [&](int g, int k) {
                    if (g == 0) stage_q(HP, j - 1, k);
                    if (g == 1) bias_q(HP, j + 1 < 2 * NT ? j + 1 : j, k);
                    if (STORE && g == 3) sreg[k] = *(const u32x4 __attribute__((address_space(3)))*)(srd + (((j - 2) >> 1) & 1) * 4096 + k * 1024);
#if defined(__HIP_DEVICE_COMPILE__)
                    if (STORE && ((g == 6 && k > 0) || (g == 7 && k == 0))) {  // (behind the sync and stage ph + 2's last piece)
                        const int pp = g == 6 ? k - 1 : 3;
                        __builtin_amdgcn_raw_buffer_store_b128(sreg[pp], kv_rs, j >= 2 ? srow[pp] : 0x80000000u, ((j - 2) >> 1) * 128, M3PC_STREAM_AUX);
                    }
#endif
                });
        mfma_done_v(HC);
    };
    for (int j = 0; j < 2 * NT - 2; j += 6) {  // tiles 0..29 (phase 16 = slot 1); even phases store, the one behind counts the stores
        kv_phase(j, S1{}, NR0{}, hA, hB, std::true_type{});
        kv_phase(j + 1, S2{}, NR4{}, hB, hA, std::false_type{});
        kv_phase(j + 2, S0{}, NR0{}, hA, hB, std::true_type{});
        kv_phase(j + 3, S1{}, NR4{}, hB, hA, std::false_type{});
        kv_phase(j + 4, S2{}, NR0{}, hA, hB, std::true_type{});
        kv_phase(j + 5, S0{}, NR4{}, hB, hA, std::false_type{});
        if (j == 12) stamps[4] = __builtin_readcyclecounter();
    }
    kv_phase(2 * NT - 2, S1{}, NR0{}, hA, hB, std::true_type{});
    kv_phase(2 * NT - 1, S2{}, NR4{}, hB, hA, std::false_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the pieces issued past the end of the stream
#pragma unroll
    for (int q = 0; q < 4; ++q) stage_q(hB, 2 * NT - 1, q);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u32x4 v = *(const u32x4 __attribute__((address_space(3)))*)(srd + (((2 * NT - 2) >> 1) & 1) * 4096 + k * 1024);
        __builtin_amdgcn_raw_buffer_store_b128(v, kv_rs, srow[k], ((2 * NT - 2) >> 1) * 128, M3PC_STREAM_AUX);
    }
#endif
    stamps[5] = __builtin_readcyclecounter();
    if (p.stamps && (int)blockIdx.x == p.stamp_block && lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) p.stamps[wu * 16 + k] = stamps[k];
    }
}

}  // namespace

// Sum of the SPLIT_N partial slabs of block_fused_kernel<0, 3> in slab order, and the LayerNorm(s) that consume the block output:
//   x = slab_0 + slab_1 + slab_2 + slab_3 -> Xout (fp32, optional);  y = LN_B?(LN_A(x)) -> Hout (bf16, optional; row groups as BlockP)
// One wave per row (d = 512: 8 values per lane).
__global__ __launch_bounds__(256) void block_split_reduce_kernel(SplitReduceP p) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= p.M) return;
    f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
#pragma unroll
    for (int s = 0; s < SPLIT_N; ++s) {
        const float* row = p.slabs + ((size_t)s * p.M + r) * BD + 8 * lane;
        const f32x4 a = *(const f32x4*)row, b = *(const f32x4*)(row + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x0[i] += a[i];
            x1[i] += b[i];
        }
    }
    if (p.Xout) {
        float* xr = p.Xout + (size_t)r * p.ldx + 8 * lane;
        *(f32x4*)xr = x0;
        *(f32x4*)(xr + 4) = x1;
    }
    if (!p.Hout) return;
    float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    auto wave_sum = [](float t) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        return t;
    };
    auto ln = [&](const float* g, const float* b) {
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s1 += v[i];
        const float mean = wave_sum(s1) * (1.0f / BD);
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s2 = fmaf(v[i] - mean, v[i] - mean, s2);
        const float rstd = rsqrtf(wave_sum(s2) * (1.0f / BD) + 1e-5f);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaf((v[i] - mean) * rstd, g[8 * lane + i], b[8 * lane + i]);
    };
    ln(p.lnA_g, p.lnA_b);
    int orow = r;
    if (p.lnB_g[0]) {
        int sel = 0;
        if (p.out_mod > 0) {
            const int w = r % p.out_mod;
            sel = w / p.out_grp;
            orow = sel * (p.M / p.out_mod) * p.out_grp + (r / p.out_mod) * p.out_grp + w % p.out_grp;
        }
        ln(p.lnB_g[sel], p.lnB_b[sel]);
    }
    bf16x8 w;
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = (bf16_t)v[i];
    *(bf16x8*)(p.Hout + (size_t)orow * p.ldh + 8 * lane) = w;
}
void launch_block_split_reduce(const SplitReduceP& p, hipStream_t st) {
    if (p.M <= 0) return;
    hipLaunchKernelGGL(block_split_reduce_kernel, dim3((p.M + 3) / 4), dim3(256), 0, st, p);
}
int block_split_n() { return SPLIT_N; }

bool launch_kv_fused(const KvFusedP& p, hipStream_t st) {
    if (((uintptr_t)p.Z & 15) || (p.ldz % 8) || ((uintptr_t)p.KV & 15) || (p.ldkv % 8)) return false;  // (16-byte pieces of whole lines)
    if (p.kv_bytes == 0 || p.kv_bytes >= 0x80000000u) return false;
    int tiles = 0;
    for (int g = 0; g < 2; ++g) {
        if (p.M[g] < 0) return false;
        if (p.M[g] == 0) continue;
        if (!p.rowtab[g] || p.rt_mod[g] < 1 || ((uintptr_t)p.rowtab[g] & 15) || ((uintptr_t)p.wstream[g] & 1023) || !p.wstream[g]) return false;
        tiles += (p.M[g] + 127) / 128;
    }
    if (p.M[0] == 0 && p.M[1] > 0) return false;  // (group 1 alone: pass it as group 0)
    if (tiles == 0) return true;
    hipLaunchKernelGGL(kv_fused_kernel, dim3(tiles), dim3(256), 0, st, p);
    return true;
}

// Everything launch_block_fused checks before it launches: a caller that has to commit to the fused tail EARLIER than the launch
// (pruned_decoder's `mixp` path lays its inputs out for this kernel) asks here first.
bool block_fused_accepts(const BlockP& p) {
    if (p.M <= 0) return true;
    if (((uintptr_t)p.O & 15) || (p.ldo % 8) || ((uintptr_t)p.wstream & 1023)) return false;
    if (!p.rowtab && (((uintptr_t)p.res & 15) || (p.ldr % 4) || (unsigned long long)p.M * p.ldr * 4 >= 0xfffffff0ull)) return false;  // (32-bit buffer offsets)
    if (p.Xout && (((uintptr_t)p.Xout & 15) || (p.ldx % 4) || (unsigned long long)(p.split ? SPLIT_N : 1) * p.M * p.ldx * 4 >= 0x80000000ull)) return false;  // (32-bit buffer offsets; a row past M is addressed out of range)
    if (p.res_L > 0 && (p.rowtab || p.Xout == p.res || p.res_nshared > p.res_L)) return false;
    if (p.res_nu < 0 || (p.res_nu > 0 && (!p.rowtab || p.res_nu > p.rt_mod || p.split))) return false;
    if (p.rowtab && (unsigned long long)(p.rt_mod + (p.res_nu > 0 ? (unsigned long long)((p.M + p.rt_mod - 1) / p.rt_mod) * p.res_nu : 0)) * BD * 4 >= 0xfffffff0ull) return false;  // (in place, the shared rows would be overwritten while read)
    if (p.Hout && (((uintptr_t)p.Hout & 7) || (p.ldh % 4))) return false;
    if (p.out_mod > 0 && (p.M % p.out_mod != 0 || p.out_mod != 2 * p.out_grp)) return false;
    if (p.QKVout) {  // the next layer's Q|K|V rows instead of its norm1 rows
        if (p.Hout || p.lnB_g[0] || !p.lnA_g || !p.bqkv || ((uintptr_t)p.QKVout & 15) || p.ldq < 3 * BD || (p.ldq % 8)) return false;  // (16-byte pieces of whole lines)
        if (p.qkv_bytes == 0 || p.qkv_bytes >= 0x80000000u || (unsigned long long)p.M * p.ldq * 2 > p.qkv_bytes) return false;
    }
    if (p.head_out[0]) {  // the two scalar output heads instead of their LayerNorm rows
        if (p.Hout || p.QKVout || p.Xout || !p.head_out[1] || !p.lnA_g || !p.lnB_g[0] || !p.lnB_g[1] || p.out_mod <= 0 || (p.M & 1)) return false;
        for (int s = 0; s < 2; ++s)
            if (!p.hb1[s] || !p.hw2[s] || !p.hb2[s] || ((uintptr_t)p.hb1[s] & 15) || ((uintptr_t)p.hw2[s] & 15) || (p.hmean[s] && !p.hstd[s])) return false;
    }
    if (p.split) {  // four workgroups per tile, fp32 partials to four slabs of M rows behind Xout (block_split_reduce sums them)
        if (p.Hout || p.QKVout || p.head_out[0] || !p.Xout || p.Xout == p.res || p.res_L > 0) return false;
    }
    if (p.x_bf16) {  // bf16 residual rows in, bf16 X'' rows out: whole 128-byte lines of 64 features
        if (p.rowtab || p.split || p.head_out[0] || p.res_L > 0 || p.variant || !p.res) return false;
        if ((p.ldr % 8) || (unsigned long long)p.M * p.ldr * 2 >= 0xfffffff0ull) return false;
        if (p.Xout && ((p.ldx % 8) || (unsigned long long)p.M * p.ldx * 2 >= 0x80000000ull)) return false;
    }
    return true;
}

bool launch_block_fused(const BlockP& p, hipStream_t st) {
    if (p.M <= 0) return true;
    if (!block_fused_accepts(p)) return false;
    const dim3 grid((p.M + 127) / 128), block(256);
#ifdef M3PC_LAB  // timing experiments (tools/block_bench.py): the lab build only
    if (p.variant == 1) hipLaunchKernelGGL((block_fused_kernel<1, 0>), grid, block, 0, st, p);
    else if (p.variant == 2) hipLaunchKernelGGL((block_fused_kernel<2, 0>), grid, block, 0, st, p);
    else if (p.variant == 3) hipLaunchKernelGGL((block_fused_kernel<3, 0>), grid, block, 0, st, p);
    else if (p.variant == 4) hipLaunchKernelGGL((block_fused_kernel<4, 0>), grid, block, 0, st, p);
    else if (p.variant == 5) hipLaunchKernelGGL((block_fused_kernel<5, 0>), grid, block, 0, st, p);
    else if (p.variant == 6) hipLaunchKernelGGL((block_fused_kernel<6, 0>), grid, block, 0, st, p);
    else if (p.variant == 7) hipLaunchKernelGGL((block_fused_kernel<7, 0>), grid, block, 0, st, p);
    else
#endif
    if (p.x_bf16 && p.QKVout) hipLaunchKernelGGL((block_fused_kernel<0, 1, 1>), grid, block, 0, st, p);
    else if (p.x_bf16) hipLaunchKernelGGL((block_fused_kernel<0, 0, 1>), grid, block, 0, st, p);
    else if (p.split) hipLaunchKernelGGL((block_fused_kernel<0, 3>), dim3((p.M + 127) / 128, SPLIT_N), block, 0, st, p);
    else if (p.QKVout) hipLaunchKernelGGL((block_fused_kernel<0, 1>), grid, block, 0, st, p);
    else if (p.head_out[0]) hipLaunchKernelGGL((block_fused_kernel<0, 2>), dim3(2 * ((p.M / 2 + 127) / 128)), block, 0, st, p);
    else hipLaunchKernelGGL((block_fused_kernel<0, 0>), grid, block, 0, st, p);
    return true;
}

}  // namespace m3pc

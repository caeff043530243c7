/*
 * m3pc_hip_debug.h -- kernel-level test and bench hooks of the LAB build (libm3pc_hip_lab.so,
 * `python -m m3pc_amd.build --lab`, compiled with -DM3PC_LAB).  The product library libm3pc_hip.so
 * exports none of these and reads no environment variable; the lab build additionally honours the
 * A/B switches M3PC_NO_* / M3PC_GEMM_VARIANT / M3PC_TWO_STREAM / ... listed in DESIGN.md section 7.
 * Users: tests/test_gemm_kernels_gpu.py, tests/test_block_fused_gpu.py, tools/*.py.
 */
#ifndef M3PC_HIP_DEBUG_H
#define M3PC_HIP_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

/* one GEMM launch of the library's dispatch on caller tensors (dtype 0 fp32 / 1 bf16 operands):
 * C = epilogue(A (M,K) W (N,K)^T + bias [gelu] [+ res]); variant selects a kernel configuration (0: dispatch) */
int m3pc_debug_gemm(int dtype, const void* A, const void* Wt, const float* bias, const float* res, void* C, int M, int N,
                    int K, int gelu, int f32out, int variant, void* stream);
/* clock probes of the last probed GEMM workgroup: {shader clocks, 100-MHz ticks} / gemm_big phase timers */
int m3pc_debug_clock(long long* out2);
int m3pc_debug_clock_big(long long* out4);
/* the top-k kernels on their own: indices of the k largest of v (n), descending, ties to the lower index */
int m3pc_debug_topk(const float* v, int n, int k, int* idx_out, void* stream);
/* the fused layer tail (block_fused.hip) on caller tensors; see csrc/m3pc.hip for the argument layout */
long long m3pc_debug_block_stream_bytes(void);
int m3pc_debug_block_fused(const void* O, int M, const float* res, const float* rowtab, int rt_mod, const void* Wo, const void* W1,
                           const void* W2, void* stream_buf, int pack, const float* bo, const float* b1, const float* b2,
                           const float* ln2_g, const float* ln2_b, const float* lnA_g, const float* lnA_b, const float* lnB_g0,
                           const float* lnB_b0, const float* lnB_g1, const float* lnB_b1, int out_mod, int out_grp, float* Xout,
                           void* Hout, int variant, void* stream, long long* stamps);
/* the same with the next layer's Q|K|V projection behind the tail: QKV (M, 1536) bf16 = LN_A(X'') Wqkv^T + bqkv */
int m3pc_debug_block_fused_qkv(const void* O, int M, const float* res, const void* Wo, const void* W1, const void* W2, const void* Wqkv,
                               void* stream_buf, const float* bo, const float* b1, const float* b2, const float* ln2_g,
                               const float* ln2_b, const float* lnA_g, const float* lnA_b, const float* bqkv, float* Xout, void* QKV,
                               void* stream, long long* stamps, int x_bf16);  /* x_bf16: res / Xout are (M, 512) bf16 rows (round 6);
                                                                                 m3pc_debug_block_fused: bit 16 of `variant` */
/* the decoder form with the two scalar output heads inside the tail (see csrc/m3pc.hip for the argument layout) */
int m3pc_debug_block_fused_heads(const void* O, int M, const float* rowtab, int rt_mod, const void* Wo, const void* W1, const void* W2,
                                 const void* Wh, void* stream_buf, const float* bo, const float* b1, const float* b2, const float* ln2_g,
                                 const float* ln2_b, const float* lnA_g, const float* lnA_b, const float* lnB_g0, const float* lnB_b0,
                                 const float* lnB_g1, const float* lnB_b1, int out_mod, int out_grp, const float* hb1, const float* hw2,
                                 const float* hb2, const float* hmean, const float* hstd, float* out0, float* out1, void* stream,
                                 long long* stamps);
/* the fused decoder input (kv_fused_kernel) on caller tensors */
long long m3pc_debug_kv_stream_bytes(void);
int m3pc_debug_kv_fused(const void* Z, int n, int Le, int kept0, int off0, int kept1, int off1, const void* We0, const void* We1,
                        const void* Wkv, void* stream_buf, const float* rowtab0, const float* rowtab1, const float* ln_g,
                        const float* ln_b, const float* bkv, void* KV, void* stream, long long* stamps);
/* the bf16 attention of an encoder layer of the candidate pass on caller tensors: QKV (batch, n_own, 1536) per-candidate rows
 * [Q | K | V] and, when n_sh > 0, QKVs (n_sh, 1536) rows shared by the batch (first layer: history tokens); O (batch, n_own + n_sh, 512),
 * shared rows first.  4 heads of 128.  kernel: 0 = what the library picks, 1 = never the pipelined kernel, 2 / 3 = the pipelined kernel
 * without its arithmetic / without its loads (timing only).
 * stamps: optional 16 int64 (device): shader-clock stamps of one workgroup's third item in the pipelined kernel */
int m3pc_debug_attention_bf16(const void* QKV, const void* QKVs, void* O, int batch, int n_own, int n_sh, int kernel, void* stream,
                              long long* stamps);
/* the decoder's bf16 attention of an rtg_guiding candidate pass: Qtab (nq <= 32, 1536) the batch-shared query rows [Q | . | .], QKVm (Lm, 1536)
 * the masked tokens' rows [. | K | V] (their block of the softmax is pre-reduced into `pre`: 4 * nq * (2 + 128) floats of scratch), KV (n, 49, 1024)
 * the candidates' own [K | V] rows; O (n, nq, 512).  kernel as m3pc_debug_attention_bf16 */
int m3pc_debug_attention_dec_bf16(const void* Qtab, const void* QKVm, const void* KV, void* O, float* pre, int n, int nq, int Lm, int kernel,
                                  void* stream);
/* the same with Le own [K | V] rows per candidate (KV (n, Le, 1024)) and nq <= 64: the T = 64 decoder is Le = 97, nq = 64 */
int m3pc_debug_attention_dec_le_bf16(const void* Qtab, const void* QKVm, const void* KV, void* O, float* pre, int n, int nq, int Lm, int Le,
                                     int kernel, void* stream);
/* the decoder's bf16 attention of a critic_lambda_guiding candidate pass: Qown (n, Lq <= 4, 512) the candidates' own query rows, Qsh (Lq2, 1536)
 * the batch-shared query rows [Q | . | .], KV (n, 49, 1024) the candidates' own [K | V] rows, QKVm (79, 1536) the batch-shared rows [. | K | V];
 * O (n, Lq + Lq2, 512), own rows first.  kernel as m3pc_debug_attention_bf16 */
int m3pc_debug_attention_mix_bf16(const void* Qown, const void* Qsh, const void* KV, const void* QKVm, void* O, int n, int Lq, int Lq2, int kernel,
                                  void* stream);
/* in-kernel phase stamps of workgroup 37 of every fused-tail launch as the step runs: cap > 0 starts a ring of cap entries
 * (64 int64 each), cap == 0 copies it to `out` (host), reports the number of launches logged and stops */
int m3pc_debug_stamp_log(m3pc_handle* h, int cap, long long* out, int* n_logged);
/* XCD / CU of every workgroup of a launch on `stream`: out[2 i] = XCC_ID, out[2 i + 1] = HW_ID */
int m3pc_debug_xcc_probe(int* out, int n_blocks, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* M3PC_HIP_DEBUG_H */

/*
 * m3pc_hip.h -- C ABI of libm3pc_hip.so, the MI355X (gfx950) implementation of the m3pc
 * test-time MPC plan step.
 *
 * The reference (wkh923/m3pc) is pure Python/PyTorch and has no native interface; the entry
 * points below are what a ctypes binding for its hot path binds (INTEGRATION.md shows the
 * stub).  Each function names the reference code it replaces (paths relative to the
 * reference root).
 *
 * Conventions
 *   - every function returns 0 on success, a negative M3PC_E* code on failure;
 *     m3pc_last_error() returns a thread-local message for the last failure.
 *   - "device" pointers are HIP device pointers owned by the caller (e.g. tensor.data_ptr());
 *     they must stay alive until the stream has been synchronised.  "host" pointers are
 *     ordinary process memory and are consumed before the call returns.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream).  Calls enqueue work on
 *     it and return without synchronising unless documented otherwise.
 *   - one handle per (process, device); a handle is not re-entrant: calls are made from one host thread
 *     at a time.  Work enqueued by DIFFERENT calls may overlap on the device only as documented at
 *     "Pipelined plan steps" below.
 *   - all tensors are dense row-major fp32 unless stated otherwise.
 *   - the library reads no environment variable (A/B switches and kernel-level hooks live in the
 *     lab build libm3pc_hip_lab.so only, declared in m3pc_hip_debug.h).
 *
 * Pipelined plan steps.  A rollout over independent windows (the reference's loops plan one window per
 * call: replay_buffer.py:204-232, learner.py:645-741) may keep several plan steps in flight: each step
 * owns one of M3PC_SLOTS step slots (m3pc_plan_args::slot: the policy head and returns tokens of its
 * policy pass), and the handle holds three workspaces -- the candidate workspace (m3pc_candidate_pass,
 * m3pc_plan_step_batch's candidate pass, m3pc_forward, m3pc_score_actions of more than max_rescore
 * candidates or in bf16), the policy workspace (m3pc_policy_pass, m3pc_plan_step_batch's policy pass) and
 * the re-score workspace (m3pc_rescore* / fp32 m3pc_score_actions of <= max_rescore candidates).  Calls
 * that use the SAME workspace must be ordered on the device (same stream, or events); calls on different
 * workspaces may run on different streams at
 * the same time.  ABI v5: there are TWO policy workspaces and TWO re-score workspaces, picked by the parity of the step slot
 * (slot & 1): the policy passes / re-scores of two steps whose slots differ in parity may run on two streams at the same time.  m3pc_amd/planner.py (plan_async) is the reference user: candidate passes back to back
 * on the caller's stream, the policy passes and the re-scores of the neighbouring steps on two more.
 */
#ifndef M3PC_HIP_H
#define M3PC_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history.  v6 (round 6): no new entry point and no structure change; m3pc_rescore_merge now writes all 8 floats of its
 * host_stats block (slots 5..7 as zeros: a reader of the 8-float layout never sees an earlier race merge's values), so a caller
 * that passed the 5-float block of m3pc_topk_window must pass 8.  Inside the library: a bf16 candidate pass carries its residual
 * stream between the encoder layers in bf16 and runs the output heads' last Linear on bf16 hidden rows (scores move within the
 * bf16 deviation the certified re-score is calibrated on; fp32 passes are unchanged).  v5 (round 5): m3pc_topk_race_window, m3pc_rescore_merge_race, m3pc_merge_race_select (the certified multinomial
 * draw of the bf16 plan step); M3PC_PLAN_PRUNED_POLICY; m3pc_profile_enable(h, 3); the handle holds two chain workspaces per kind,
 * picked by the parity of m3pc_plan_args::slot.  v4: m3pc_goal_step_batch, m3pc_dims::max_goal_batch.  v3: M3PC_PLAN_DEFER_JOIN,
 * m3pc_candidate_join.  No structure changed layout in v5. */
#define M3PC_ABI_VERSION 6

#define M3PC_OK 0
#define M3PC_EINVAL (-1)   /* bad argument / shape mismatch            */
#define M3PC_ESTATE (-2)   /* weights / tokenizer / critic not loaded  */
#define M3PC_EHIP (-3)     /* HIP runtime error                        */
#define M3PC_ENOMEM (-4)   /* workspace too small for the request      */

/* modality order everywhere: the reference's dict insertion order
 * states, actions, rewards, returns (finetune_omtm/learner.py:348-366; mtm_model.py:625,676) */
#define M3PC_STATES 0
#define M3PC_ACTIONS 1
#define M3PC_REWARDS 2
#define M3PC_RETURNS 3

/* plan_guidance modes (finetune_omtm/learner.py:389-407) */
#define M3PC_MODE_RTG 0     /* rtg_guiding            learner.py:271-327 */
#define M3PC_MODE_CRITIC 1  /* critic_lambda_guiding  learner.py:211-268 */
#define M3PC_MODE_NOISE 2   /* noise_adding_lambda    learner.py:142-208 */

/* arithmetic of the batched candidate pass */
#define M3PC_PREC_FP32 0  /* fp32 operands, f32 MFMA (v_mfma_f32_32x32x2_f32), fp32 accumulate */
#define M3PC_PREC_BF16 1  /* bf16 operands, bf16 MFMA (v_mfma_f32_32x32x16_bf16), fp32 accumulate,
                             fp32 residual stream / LayerNorm / softmax */

typedef struct m3pc_handle m3pc_handle;

/* static sizes of one planner instance: omtmConfig + data_shapes + traj_length
 * (mtm_model.py:200-221, 324-344; finetune_omtm/config.yaml:5,29-34) */
typedef struct m3pc_dims {
    int state_dim;      /* S */
    int action_dim;     /* A */
    int traj_length;    /* T  (cfg.traj_length == model max_len) */
    int n_embd;         /* d, multiple of 64 */
    int n_head;
    int n_enc_layer;
    int n_dec_layer;
    int max_candidates; /* largest n_count a plan_step call will use on this device */
    int max_batch;      /* largest B for m3pc_forward */
    int critic_hidden;  /* TwinQ hidden width (256), 0 = no critic */
    int max_rescore;    /* largest n of m3pc_rescore* / fp32 m3pc_score_actions that runs in the re-score
                           workspace (beside a candidate pass); 0 = 64 */
    int max_goal_batch; /* largest batch of m3pc_goal_step_batch (zero-shot windows per call); 0 = none.  ABI v4 */
} m3pc_dims;

/* step slots of a handle (see "Pipelined plan steps") */
#define M3PC_SLOTS 4

/* one entry of a state_dict: fp32, contiguous, torch layout */
typedef struct m3pc_named_tensor {
    const char* name;   /* e.g. "encoder.layers.0.self_attn.in_proj_weight" */
    const float* data;
    long long numel;
    int on_device;      /* 0: host pointer, 1: device pointer */
} m3pc_named_tensor;

typedef struct m3pc_plan_args {
    int mode;          /* M3PC_MODE_*                                                        */
    int precision;     /* M3PC_PREC_* for the candidate pass (the policy pass is always fp32) */
    int horizon;       /* effective h of this step (learner.py:342-345)                      */
    int n_total;       /* cfg.action_samples: leading dimension of eps                       */
    int n_begin;       /* first candidate scored by this call (candidate sharding)           */
    int n_count;       /* number of candidates scored by this call                           */
    double lmbda;      /* TD(lambda) mixing, python float in the reference (learner.py:313-316) */
    double discount;   /* gamma (learner.py:307-309)                                         */
    double rtg;        /* return-to-go written into every returns slot (learner.py:368-385)  */
    int slot;          /* step slot in [0, M3PC_SLOTS) holding this step's policy pass           */
    int returns_f64;   /* dtype of `returns`: 0 float32, 1 float64                               */
    const void* returns; /* optional device (T,) raw returns row of the window (what
                          trajectory["returns"] holds, learner.py:272-293); NULL: `rtg` everywhere */
    int flags;         /* M3PC_PLAN_* bits (m3pc_candidate_pass); 0 = none                       */
    int window;        /* m3pc_candidate_pass behind m3pc_policy_pass_batch: which of the slot's windows
                          this step plans (0 after a single-window m3pc_policy_pass)                */
} m3pc_plan_args;

/* m3pc_plan_args::flags.  M3PC_PLAN_DEFER_JOIN (m3pc_candidate_pass only): a large bf16 pass runs its candidate
 * parts on the caller's stream and on streams of the handle; by default the caller's stream waits for all of them
 * before the call returns to it (outputs complete in stream order).  With this bit it does not: the outputs of the
 * parts that ran elsewhere are complete only behind m3pc_candidate_join(h, slot, stream), and the caller's stream is
 * free to start the next step's first part while the last part of this one still runs (no idle tail per step). */
#define M3PC_PLAN_DEFER_JOIN 1
/* M3PC_PLAN_PRUNED_POLICY (m3pc_policy_pass only, with loc == std == NULL): the policy head is computed at the h action tokens
 * t >= T - h alone -- the rows a plan step samples its candidates from (learner.py:285-287 reads sample((N,))[:, 0, T-h:]) -- through
 * the exactly pruned decoder (those tokens are masked under the rcbc mask: shared query rows and masked-token K|V from the plan
 * tables, the kept tokens alone through decoder-embed / K|V, out-proj / FFN / actor head on h rows instead of 4T).  Rows t < T - h of
 * the slot's loc / std (what m3pc_candidate_pass copies out) are zero.  Same arithmetic per computed row up to fp32 re-association. */
#define M3PC_PLAN_PRUNED_POLICY 2

const char* m3pc_last_error(void);
int m3pc_abi_version(void);

/* Learner.__init__'s model construction (learner.py:32-36): allocates handle, weight arena and
 * workspace on `device`.  Synchronous. */
int m3pc_create(const m3pc_dims* dims, int device, m3pc_handle** out);
int m3pc_destroy(m3pc_handle* h);

/* omtm.load_state_dict (learner.py:33-35): tensors by state_dict name.
 * Unknown names are ignored.  The FIRST call must bring every required name; later calls may bring any
 * subset -- fine-tuning updates the weights between rollouts (finetune.py:306) -- and re-derive only what
 * depends on the tensors that came: their bf16 copies, the packed fragment streams of the layers they
 * belong to, the embedding tables, and (decoder-side tensors only) the cached mask-pattern tables.
 * Synchronous (the tensors may be released when the call returns). */
int m3pc_load_weights(m3pc_handle* h, const m3pc_named_tensor* tensors, int n, void* stream);
/* what the last m3pc_load_weights did: out4 = {tensors copied, fused-layer-tail streams re-packed,
 * fused-decoder-input streams re-packed, 1 if the cached decoder tables were invalidated} */
int m3pc_load_stats(m3pc_handle* h, long long* out4);

/* ContinuousTokenizer(mean, std, stats, normalize) (tokenizers/continuous.py:31-62):
 * host arrays of `dim` floats. */
int m3pc_set_tokenizer(m3pc_handle* h, int key, const float* mean, const float* std, int dim, int normalize);

/* TwinQ weights (finetune_omtm/model.py:146-171): "q1.net.0.weight" ... "q2.net.4.bias",
 * plus the observation mean / std attributes (host, state_dim floats each). */
int m3pc_set_critic(m3pc_handle* h, const m3pc_named_tensor* tensors, int n, const float* obs_mean,
                    const float* obs_std, void* stream);

/* ContinuousTokenizer.encode / decode (tokenizers/continuous.py:68-94) on device rows:
 * out = (in - mean) / std   resp.  out = in * std + mean  (identity when normalize == 0).
 * `in_f64` != 0: the input rows are float64 and are normalised in float64 before the cast
 * (the reference's behaviour for `returns`, learner.py:371-374). */
int m3pc_tokenize(m3pc_handle* h, int key, const void* in, int in_f64, float* out, long long rows, void* stream);
int m3pc_detokenize(m3pc_handle* h, int key, const float* in, float* out, long long rows, void* stream);

/* omtm.forward (mtm_model.py:593-607) for one token per timestep and modality.
 *   tokens[k]  device (B,T,D_k) tokenised inputs
 *   masks[k]   host   (T,) 0/1 bytes, shared by the batch (mtm_model.py:572-575)
 *   out_*      device, may be NULL to skip a head:
 *              out_states (B,T,S), out_rewards (B,T,1), out_returns (B,T,1)   raw head outputs
 *              out_mu / out_std (B,T,A)     DiagGaussianActor loc / std (mtm_model.py:313-321)
 */
int m3pc_forward(m3pc_handle* h, int batch, const float* const tokens[4], const unsigned char* const masks[4],
                 float* out_states, float* out_rewards, float* out_returns, float* out_mu, float* out_std,
                 int precision, void* stream);

/* Zero-shot goal reaching: both forwards of action_piid_sample (zeroshot_omtm/learner.py:151-261) in one call, on RAW
 * (un-normalised) windows -- the tokenizer is applied inside, as in m3pc_plan_step:
 *   path inference under the pi mask (zeroshot_omtm/masks.py:72-91) -> the de-tokenised predicted observations over the
 *   window rows [0, idx] and [idx+2, T-2] (learner.py:240-246) -> inverse dynamics under the fid mask (masks.py:30-47).
 *   batch      E independent windows (<= max_batch); states (E,T,S), actions (E,T,A), rewards (E,T,1) device
 *   rtg        host (E,) return-to-go per window, written into every returns slot (learner.py:205-223)
 *   masks_*    host (T,) 0/1 bytes per modality, as m3pc_forward;  idx = T - h
 *   inferred       device out (E,T,S): the de-tokenised states head of the first forward
 *   window_states  device out (E,T,S): the observation rows the second forward saw
 *   out_mu/out_std device out (E,T,A): DiagGaussianActor loc / std of the second forward (mtm_model.py:313-321)
 * fp32; runs in the policy workspace and uses step slot 0. */
int m3pc_goal_step(m3pc_handle* h, int batch, const float* states, const float* actions, const float* rewards,
                   const double* rtg, const unsigned char* const masks_pi[4], const unsigned char* const masks_fid[4],
                   int idx, float* inferred, float* window_states, float* out_mu, float* out_std, void* stream);

/* m3pc_goal_step for MANY windows per call (BASELINE config 5: 64 environments x 1024 = 8192 windows per GPU), exactly
 * pruned to what the reference reads of the two forwards and run in the arithmetic of the candidate pass:
 *   path inference (pi mask, zeroshot_omtm/masks.py:72-91): the states head is read at the window rows t <= idx and
 *   idx+2 <= t <= T-2 only (zeroshot_omtm/learner.py:240-246) -- those decoder tokens are the queries of the decoder layer;
 *   inverse dynamics (fid mask, masks.py:30-47): the action distribution is read at token idx only (learner.py:250-256) --
 *   ONE query per window.  Un-read decoder rows are never computed; read rows are computed as in m3pc_goal_step.
 *   Neither mask keeps a rewards or returns token, so those rows of the window (and the return-to-go) never enter the
 *   arithmetic: the call takes states and actions only.
 *   batch      E independent windows (<= max_goal_batch): states (E,T,S), actions (E,T,A) device, RAW (un-normalised)
 *   idx        T - h; the masks are built inside (plan tables cached per idx)
 *   goal_mode  M3PC_GOAL_PIID: action_piid_sample (learner.py:151-261), both forwards;
 *              M3PC_GOAL_ID:   action_id_sample (learner.py:60-149), one forward under the gid mask (masks.py:50-69)
 *   precision  M3PC_PREC_BF16: bf16 MFMA kernels of the candidate pass (fused layer tails); M3PC_PREC_FP32: fp32 MFMA.
 *              Either way a window's result does not depend on which other windows share the call, as long as the
 *              batch sizes fall in the same kernel regime (environment sharding: no collective).  fp32: bit-identical at any
 *              batch size.  bf16: bit-identical between calls of the same regime; the regime boundaries are >= 2048 windows
 *              (the call runs as two parts on two streams), >= 12288 token rows per pass (fused layer tails; below: the
 *              split / GEMM forms), >= 64 windows (two short windows per attention tile) and >= 1024 (window, head) items
 *              (pipelined attention) -- a remainder shard that crosses one agrees with the unsharded call to the bf16
 *              tolerance (|d loc| <= 3e-2), not bit for bit; shard evenly, or use fp32, where bits must match.
 *   window_states  device out (E,T,S), optional: the observation rows the inverse-dynamics forward saw
 *   out_mu/out_std device out (E,A): DiagGaussianActor loc / std at token idx (mtm_model.py:313-321)
 * Runs in the candidate workspace. */
#define M3PC_GOAL_PIID 0
#define M3PC_GOAL_ID 1
int m3pc_goal_step_batch(m3pc_handle* h, int batch, const float* states, const float* actions, int idx, int goal_mode,
                         int precision, float* window_states, float* out_mu, float* out_std, void* stream);

/* The two halves of m3pc_plan_step as calls of their own, for pipelined callers:
 *   m3pc_policy_pass     learner.py:278-284: returns tokens + return-conditioned policy (batch 1, rcbc
 *                        mask, fp32) -> loc / std of the slot (and the caller's copies, optional).  Policy
 *                        workspace.
 *   m3pc_candidate_pass  learner.py:285-316: candidates from the slot's policy head and eps, batched
 *                        forward under the fd mask, TD(lambda) scores.  Candidate workspace.  Arguments as
 *                        m3pc_plan_step. */
int m3pc_policy_pass(m3pc_handle* h, const m3pc_plan_args* args, const float* states, const float* actions,
                     const float* rewards, float* loc, float* std, void* stream);
int m3pc_candidate_pass(m3pc_handle* h, const m3pc_plan_args* args, const float* states, const float* actions,
                        const float* rewards, const float* eps, float* loc, float* std, float* sample_actions,
                        float* expect_return, float* pred_rewards, float* pred_boot, void* stream);
/* The policy passes of E independent windows as ONE pass at batch E (learner.py:278-284 per window; several
 * environments step together: replay_buffer.py:204-232 per environment): states (E,T,S), actions (E,T,A), rewards (E,T,1)
 * device, rtg host (E,).  The slot then holds E policy heads; m3pc_candidate_pass plans window w of them with
 * m3pc_plan_args::window = w and that window's rows of states / actions / rewards.  loc / std: optional (E,T,A) out.
 * args: mode, horizon, slot are read.  E <= max_batch.  The few-row fp32 kernels choose their tiling by the row count, so
 * a window's policy head agrees with the single-window pass to fp32 rounding, not bit for bit. */
int m3pc_policy_pass_batch(m3pc_handle* h, const m3pc_plan_args* args, int n_windows, const float* states,
                           const float* actions, const float* rewards, const double* rtg, float* loc, float* std,
                           void* stream);
/* Orders `stream` behind every part of the last m3pc_candidate_pass(M3PC_PLAN_DEFER_JOIN) of step slot `slot`
 * (no-op when that pass joined by itself or ran in one part).  The consumer of a step's scores -- the re-score +
 * select of learner.py:318-325 -- calls it on its own stream. */
int m3pc_candidate_join(m3pc_handle* h, int slot, void* stream);

/* One MPC plan step up to (not including) the cross-candidate select:
 * rtg_guiding / critic_lambda_guiding / noise_adding_lambda (learner.py:142-316).
 *   states/actions/rewards   device (T,S) (T,A) (T,1): the raw (un-normalised) window assembled by
 *                            action_sample (learner.py:348-366), future rows zero
 *   eps       device standard normals: (n_total,T,A) for RTG/CRITIC -- the reference draws
 *             dist.sample((N,)) over every timestep (learner.py:285-287) -- or (n_total,h,A)
 *             for NOISE (learner.py:157-163)
 *   loc,std   device out (T,A), optional: the policy pass distribution
 *   sample_actions  device out (n_count,h,A): candidates [n_begin, n_begin+n_count)
 *   expect_return   device out (n_count,): TD(lambda) scores BEFORE the max shift
 *   pred_rewards, pred_boot  device out (n_count,h), optional: decoded predicted rewards and
 *             the bootstrap term (1000 x predicted return, or min(q1,q2))
 */
int m3pc_plan_step(m3pc_handle* h, const m3pc_plan_args* args, const float* states, const float* actions,
                   const float* rewards, const float* eps, float* loc, float* std, float* sample_actions,
                   float* expect_return, float* pred_rewards, float* pred_boot, void* stream);

/* Score caller-supplied candidate action sequences: learner.py:288-316 (from the overwrite of the window's future actions
 * on), without the policy pass and without sampling.  Serves the fp32 re-score of batched plans, CEM-style refinement
 * (sequence_dataset.py:919-1000) and any caller that brings its own candidates.
 *   n_windows E >= 1 history windows: states (E,T,S), actions (E,T,A), rewards (E,T,1) device, raw (un-normalised)
 *   cand            device (n_count,h,A): the candidates' actions of the last h steps (rows < T-h come from the window)
 *   window_index    device (n_count,) int32: the window each candidate continues; NULL = all window 0 (E == 1)
 *   args            mode (RTG or CRITIC scoring), precision, horizon, n_count, lmbda, discount are read
 *   expect_return   device out (n_count,); pred_rewards / pred_boot device out (n_count,h), optional */
int m3pc_score_actions(m3pc_handle* h, const m3pc_plan_args* args, int n_windows, const float* states,
                       const float* actions, const float* rewards, const float* cand, const int* window_index,
                       float* expect_return, float* pred_rewards, float* pred_boot, void* stream);

/* m3pc_plan_step for E independent windows in one pass of the kernels (the counterpart of a rollout loop that steps E
 * environments: replay_buffer.py:204-232 / learner.py:681-691 plan one window per call): policy pass at batch E,
 * args->n_total candidates per window, one candidate pass over all E * n_total rows.  All windows share the horizon.
 *   states (E,T,S), actions (E,T,A), rewards (E,T,1) device;  rtg host (E,) doubles
 *   eps device (E, n_total, T, A) [RTG/CRITIC] or (E, n_total, h, A) [NOISE];  window_index device (E*n_total,) int32 = c / n_total
 *   loc, std device out (E,T,A), optional;  sample_actions device out (E*n_total,h,A);  expect_return device out (E*n_total,) */
int m3pc_plan_step_batch(m3pc_handle* h, const m3pc_plan_args* args, int n_windows, const float* states,
                         const float* actions, const float* rewards, const double* rtg, const float* eps,
                         const int* window_index, float* loc, float* std, float* sample_actions, float* expect_return,
                         void* stream);

/* fp32 re-scoring of an arbitrary subset of the candidates of the plan step that owns args->slot
 * (its policy-pass loc/std are reused; same window, eps and args as that call, args->precision ignored).
 * Runs in the re-score workspace when n <= max_rescore, else in the candidate workspace.
 * Used after a bf16 candidate pass to make the reported arg-max independent of bf16 rounding: the same
 * learner.py:288-316 arithmetic, restricted to rows `index`.
 *   index           device (n,) int32 candidate ids in [0, n_total)
 *   sample_actions  device out (n,h,A), optional
 *   expect_return   device out (n,) */
int m3pc_rescore(m3pc_handle* h, const m3pc_plan_args* args, const float* states, const float* actions,
                 const float* rewards, const float* eps, const int* index, int n, float* sample_actions,
                 float* expect_return, void* stream);

/* m3pc_rescore of the k best entries of a full score vector, written back in place:
 *   expect_return  device in/out (n_total,): scores of ALL candidates (after the all-gather when
 *                  sharded); its k largest entries are replaced by their fp32 re-scores
 *   topk_index     device out (k,) int32, optional: the re-scored candidate ids, best first */
int m3pc_rescore_topk(m3pc_handle* h, const m3pc_plan_args* args, const float* states, const float* actions,
                      const float* rewards, const float* eps, float* expect_return, int k, int* topk_index,
                      void* stream);

/* Bound-driven re-score set (the arg-max of learner.py:318-325 must not depend on bf16 rounding): if every bf16 score is
 * within delta of its fp32 value up to a common shift, the fp32 arg-max lies among the candidates whose bf16 score is
 * within window = 2 delta of the bf16 maximum.  Finds the kmax + 1 best entries of expect_return (descending; ties to the
 * lower index) and n = clamp(#{E_i >= max E - window}, kmin, kmax).
 *   topk_index  device out (kmax + 1,) int32
 *   stats       device out float[4]: {n, max E - (best E not among the n) [inf if none], max E, the count over ALL n_total
 *               entries, unclamped: > kmax means the listed prefix does not cover the window}
 *   top_scores  device out (kmax + 1,), optional: expect_return[topk_index[i]] (kept for m3pc_rescore_merge)
 *   host_stats  optional: host-mapped (pinned) float[5] the kernel also writes -- the four stats, then `seq` into [4] with
 *               system scope; a caller that spins on host_stats[4] == seq reads n without a stream synchronisation */
int m3pc_topk_window(m3pc_handle* h, const float* expect_return, int n_total, int kmax, int kmin, float window,
                     int* topk_index, float* stats, float* top_scores, float* host_stats, float seq, void* stream);
/* The score vector the select runs on after a partial fp32 re-score, and the certificate that the partial re-score was
 * enough.  bf16 scores and their fp32 values differ by a common shift c plus a bounded deviation, so un-re-scored entries
 * are brought onto the fp32 scale before they meet the re-scored ones in one softmax / arg-max (learner.py:318-325):
 *   index / top_scores / top_rescored: the n best candidates by bf16 score (best first: m3pc_topk_window), their bf16
 *   scores and their fp32 re-scores (m3pc_rescore)
 *   c = lower median over the n listed entries of (top_scores[i] - top_rescored[i])
 *   merged[j] = scores[j] - c for all j < n_total;  merged[index[i]] = top_rescored[i]
 *   With |(bf16_j - fp32_j) - c| <= delta for every candidate, an un-listed j can hold the fp32 arg-max only if
 *   scores[j] > f* + c - delta (f* = the best re-scored fp32 score).  need = #{j : scores[j] > f* + c - delta}: need <= n
 *   certifies arg-max(merged) = the fp32 arg-max; otherwise re-score entries n .. need-1 of the order and merge again
 *   (need can only shrink).
 *   stats device out float[4] = {c, max_i |top_scores[i] - top_rescored[i] - c|, need, threshold - best un-listed score};
 *   host_stats / seq as for m3pc_topk_window, except that host_stats must hold 8 floats here: slots 5..7 are written as zeros,
 *   so that a reader of the 8-float layout of m3pc_rescore_merge_race never sees an earlier race merge's values on the same
 *   buffer.  `merged` may alias `scores`. */
int m3pc_rescore_merge(m3pc_handle* h, const float* scores, int n_total, const int* index, int n, const float* top_scores,
                       const float* top_rescored, float delta, float* merged, float* stats, float* host_stats, float seq,
                       void* stream);
/* The SAMPLED action of a plan step (learner.py:324-325: sample_action = a0[torch.multinomial(p, 1)], the action every online
 * rollout step executes, replay_buffer.py:206-216) must not depend on bf16 rounding either.  torch.multinomial(p, 1) is
 * arg-max_j p_j / q_j with q ~ Exp(1) (m3pc_select), and with p = softmax(temperature (E - max E)) that is the race
 * arg-max_j (temperature E_j - log q_j).  An un-re-scored candidate's key is off by at most temperature * delta, so only
 * candidates whose bf16 key comes within 2 temperature delta of the best can win: ABI v5 lists them, re-scores them beside the
 * best-by-score candidates and certifies the draw as m3pc_rescore_merge certifies the arg-max.
 *
 * m3pc_topk_race_window: m3pc_topk_window (window = 0) plus the rmax best candidates by race key, both in ONE list of
 * rmax + kmax + 1 entries laid out for contiguous slices:
 *   list[rmax + i]      the i-th best candidate by score (i <= kmax), best first
 *   list[rmax - 1 - i]  the i-th best candidate by race key temperature * E_j - log expo_j (i < rmax)
 * so "the r best racers and the n best scorers" is list[rmax - r .. rmax + n): one m3pc_rescore call, one merge.
 *   expo         device (n_total,) the Exp(1) variates m3pc_select will draw with
 *   list_scores  device out (rmax + kmax + 1,), optional: expect_return[list[i]]
 *   stats / host_stats / seq as m3pc_topk_window (of the score part); stats may be NULL (one launch fewer up to 2048 candidates:
 *   the ranking kernel writes list_scores itself).  rmax <= min(64, n_total). */
int m3pc_topk_race_window(m3pc_handle* h, const float* expect_return, const float* expo, float temperature, int n_total, int kmax,
                          int kmin, int rmax, int* list, float* stats, float* list_scores, float* host_stats, float seq, void* stream);
/* m3pc_rescore_merge over such a list: `list` = r race entries followed by n score entries (a slice of the list above),
 * list_scores / list_rescored their bf16 scores and fp32 re-scores.  The shift c, the deviation, merged[] and the arg-max
 * certificate (need) are m3pc_rescore_merge's over all r + n entries.  Race certificate: K* = max over the listed entries of
 * (temperature * rescored_i - log expo_i); need_race = #{j : temperature * scores[j] - log expo[j] >= K* + temperature (c - delta)}
 * over the whole vector: need_race <= r certifies that the draw arg-max_j merged-p_j / expo_j is the fp32 draw; otherwise re-score
 * the race entries r .. need_race-1 and merge again.
 *   stats device out float[8] = {c, deviation, need, margin, -, need_race, K*, K* + temperature (c - delta)}; the host copy
 *   carries `seq` in slot 4 (host_stats must hold 8 floats).  r + n <= 1024. */
int m3pc_rescore_merge_race(m3pc_handle* h, const float* scores, const float* expo, float temperature, int n_total, const int* list,
                            int r, int n, const float* list_scores, const float* list_rescored, float delta, float* merged,
                            float* stats, float* host_stats, float seq, void* stream);
/* m3pc_rescore_merge_race followed by m3pc_select on the merged vector (with the same expo / temperature), in ONE launch: both are
 * one workgroup over the whole vector, and the re-score's tail is a chain of dependent launches.  The certificate's statistics
 * reach host_stats before the select part runs.  Arguments as the two calls. */
int m3pc_merge_race_select(m3pc_handle* h, const float* scores, const float* expo, float temperature, int n_total, const int* list,
                           int r, int n, const float* list_scores, const float* list_rescored, float delta, float* merged,
                           float* stats, float* host_stats, float seq, const float* a0, long long a0_stride, float* p,
                           float* eval_action, int* argmax, int* sample_idx, float* sample_action, void* stream);
/* m3pc_rescore of the n listed candidates (device int32 ids), written back into the full score vector in place. */
int m3pc_rescore_listed(m3pc_handle* h, const m3pc_plan_args* args, const float* states, const float* actions,
                        const float* rewards, const float* eps, const int* index, int n, float* expect_return,
                        void* stream);

/* The cross-candidate tail (learner.py:318-325) over all N candidates (after an all-gather when
 * sharded): p = softmax(temperature * (E - max E)), eval_action = sum p*a0 / sum p, argmax E, and the
 * multinomial draw sample_action = a0[multinomial(p, 1)].
 *   a0 device (n, A) rows with stride a0_stride floats (sample_actions[:,0,:] => stride h*A)
 *   expo      device (n,) Exp(1) variates, optional.  torch.multinomial(p, 1) is argmax(p / q) with
 *             q = empty_like(p).exponential_(1) drawn from the caller's generator; passing that q
 *             here reproduces torch's draw without its dozen tiny launches.
 *   p device out (n,), eval_action device out (A,), argmax device out (1,),
 *   sample_idx device out (1,), sample_action device out (A,) -- each optional */
int m3pc_select(m3pc_handle* h, const float* expect_return, const float* a0, long long a0_stride, int n,
                float temperature, const float* expo, float* p, float* eval_action, int* argmax,
                int* sample_idx, float* sample_action, void* stream);

/* kernel-level timing of the last plan_step for bench.py / profiling: when enabled the library
 * brackets the MFMA launches with hipEvents on the stream they run on.  enable = 2: in addition the two candidate
 * halves of a plan step (normally overlapped on two streams) run one after the other on `stream`, so that a
 * bracket holds that launch alone.  enable = 3: that ordering WITHOUT the event brackets (for an external kernel trace
 * of every launch alone on the chip: rocprofv3 -- python3 bench.py --depth 0 --serial-halves). */
int m3pc_profile_enable(m3pc_handle* h, int enable);
/* sums since the last reset over the MFMA launches of one arithmetic (precision = M3PC_PREC_*, or -1 for
 * all) or over the fused layer-tail launches alone (M3PC_PROF_LAYER_TAIL: the dominant kernel, block_fused.hip):
 * launches, milliseconds between the bracketing events, flops (2*M*N*K per product) */
#define M3PC_PROF_LAYER_TAIL 16
int m3pc_profile_read(m3pc_handle* h, int precision, long long* launches, double* gemm_ms, double* gemm_flops,
                      int reset);

#ifdef __cplusplus
}
#endif
#endif /* M3PC_HIP_H */

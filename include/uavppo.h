/*
 * uavppo.h -- C ABI of libuavppo.so: the MI355X (gfx950) hot path of the PPOV2.0/2.1 trainer.
 *
 * The reference (su1phurd/UAV-WRF-LES-PPO-LSTM) has no FFI layer: its boundary is the Python
 * module surface used by PPOV2.0/train_ppo2.0.py:8-13 (config / environment / model).  The
 * Python modules under uav-wrf-les-ppo-lstm_amd/ keep that surface and bind these entry points
 * with ctypes (INTEGRATION.md shows the stub).  Each entry point names the reference code it
 * replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (owned by the caller, e.g. a torch tensor) unless a
 *     parameter is documented "host";
 *   - `stream` is a hipStream_t passed as void*; no entry point synchronises with the host;
 *   - return 0 on success, non-zero on error (uav_last_error() gives the text); the Python
 *     side raises RuntimeError, the reference's error convention (model.py:47-49);
 *   - rollout buffers are (env, T, feat) row-major: x[n][t][f];
 *   - thread-compatible: one caller thread per handle; a handle owns ONE scratch workspace, so the calls made on it
 *     must be ordered on the device (one stream, or events between streams) -- use one handle per concurrent stream.
 */
#ifndef UAVPPO_H
#define UAVPPO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct uav_ctx uav_ctx;   /* opaque: device id, CU count, one scratch workspace */
typedef void* uav_stream;         /* hipStream_t */

#define UAV_ABI_VERSION 9   /* 9: uav_lstm_stepper_step_pair; uav_lstm_dgates_bytes / uav_lstm_dgates_f32: at h = 256 on the fp16-split arithmetic `dgates` is an opaque buffer (the BPTT's fp16 piece chunks, stored once) -- size it with uav_lstm_dgates_bytes; UAV_DEBUG_DG_F32; 8: uav_lstm_wgrad db_hh; uav_comm_* / uav_allreduce / uav_allreduce_f64 / uav_allgather_bytes / uav_rccl_version (RCCL behind the ABI); the h = 256 cluster kernels, uav_lstm_cluster_errors and UAV_DEBUG_CLUSTER* left the library (tools/experiments/lstm_cluster); 7: uav_env_cfg.curriculum, uav_curriculum_* (device-side curriculum), uav_episode_rows; 6: uav_lstm_cluster_errors, UAV_DEBUG_CLUSTER (h = 256 persistent cluster kernels); 5: uav_rollout_tail; 4: uav_set_debug_flags; the fused MLP kernels follow uav_set_lstm_arith (fp16 split by default); the UAV_LSTM_* environment variables are read once, by uav_create; procedural field / step noise in f64 (pinned by oracle/procedural_oracle.py); 3: uav_gemm_f16x3, uav_lstm_stepper_*, uav_policy_sample_at, uav_store_transition, uav_lstm_bwd (w_ih, I, dx), uav_lstm_bwd_caps, uav_lstm_bwd_stack; 2: uav_policy_sample index_offset, uav_clip_adam pmax_out, uav_set_lstm_arith, uav_absmax, uav_mlp_ppo_grad, uav_rollout policy_kind 0 */

/* GAE modes (train_ppo2.0.py:18-32 vs PPOV1.0/ppo0.0.py:337-350) */
#define UAV_GAE_REFERENCE_EXACT 0  /* mask from done[t+1], last step bootstraps from itself */
#define UAV_GAE_STANDARD        1  /* mask from done[t], bootstrap from last_val            */

/* environment variants */
#define UAV_ENV_V20 0   /* sigma = 500/16, clip 499      PPOV2.0/environment.py:54,105 */
#define UAV_ENV_V21 1   /* sigma = 15                    PPOV2.1/environment.py:56     */
#define UAV_ENV_V11 2   /* clip 500-1e-6, MAX_STEPS 5000 PPOV1.1/environment.py:105    */

/* field sources for the plume sampler */
#define UAV_FIELD_PROCEDURAL   0  /* counter-RNG field, O(1) memory per env           */
#define UAV_FIELD_MATERIALISED 1  /* bank of [F][500][500][2] f64 tables in HBM       */

int         uav_abi_version(void);
const char* uav_last_error(void);

/* ws_bytes: scratch for deterministic two-stage reductions and split-K slabs (>= 1 MiB).
 * A handle belongs to ONE device; a process may hold handles for several devices (per-kernel launch attributes are kept
 * per device inside the library).  Calls through a handle must be made with that device current (torch does this) and
 * from one thread at a time.  uav_create reads the process-level switches UAV_LSTM_F32_MFMA / UAV_LSTM_BF16X6 (initial
 * uav_set_lstm_arith mode) and UAV_LSTM_STEP_F32 / UAV_LSTM_X_F32 (initial debug flags) from the environment, once;
 * no entry point reads the environment afterwards. */
int  uav_create(uav_ctx** out, int device, size_t ws_bytes);
void uav_destroy(uav_ctx* ctx);

/* How uav_lstm_fwd / _bwd / _wgrad evaluate their f32 matrix products (per handle; default FP16X3).  All three give f32
 * results; they differ in the operand RANGE they accept (see the range note at uav_lstm_wgrad) and in speed:
 *   FP16X3   two fp16 pieces per operand, three MFMA products   |w| < 65504, |x| < 4096, |h0| < 64   fastest
 *   BF16X6   three bf16 pieces, six products                    f32's whole exponent range           ~1.2x slower
 *   F32_MFMA exact-f32 MFMA                                     f32's whole exponent range           ~1.7x slower
 * A caller that cannot bound its operands measures them (uav_absmax; uav_clip_adam's pmax_out) and switches: the Python
 * trainer does exactly that (uavppo/trainer.py: check_ranges).  uav_rollout exists in the FP16X3 form only. */
#define UAV_ARITH_FP16X3   0
#define UAV_ARITH_BF16X6   1
#define UAV_ARITH_F32_MFMA 2
int uav_set_lstm_arith(uav_ctx* ctx, int mode);
int uav_get_lstm_arith(const uav_ctx* ctx);
/* A/B switches of the h = 256 step path (tests / measurements only; results are bit-identical or within f32 tolerance):
 *   STEP_F32  uav_lstm_fwd / _bwd at h = 256 take the generic exact-f32 step path instead of the fp16-split step kernels
 *   X_F32     a wide layer input is read as f32 rows by every workgroup instead of pre-split piece planes
 * Every flag leaves results bit-identical or within f32 tolerance; nothing here can make an entry point return garbage. */
#define UAV_DEBUG_STEP_F32 1u
#define UAV_DEBUG_X_F32    2u
#define UAV_DEBUG_DG_F32   4u          /* h = 256 step path: gate gradients ALSO written as f32 rows and the weight gradients taken from those (the round-4 form; an A/B switch, results within f32 tolerance) */
#define UAV_DEBUG_GEMM_TN_OFF 0x10u   /* the dW-shaped split-fp16 products on the older one-slab-in-flight kernel (same results; an A/B switch) */
int uav_set_debug_flags(uav_ctx* ctx, unsigned flags);
/* out[0] (f32, device) = max |x[i]| over n floats, a NaN counting as +inf: the range probe for the modes above. */
int uav_absmax(uav_ctx* ctx, const float* x, int64_t n, float* out, uav_stream stream);

/* ---- G1: GAE scan.  Replaces train_ppo2.0.py:18-32 (one wavefront per env row, affine
 * suffix scan over T with wave shuffles).  rew,val,done,adv: f32 [n_env][horizon].
 * last_val: f32 [n_env] (STANDARD mode) or NULL. */
int uav_gae(uav_ctx* ctx, const float* rew, const float* val, const float* done,
            const float* last_val, int n_env, int horizon, float gamma, float lam, int mode,
            float* adv, uav_stream stream);

/* ---- G2: whole-buffer advantage statistics and normalisation (train_ppo2.0.py:35-40).
 * uav_adv_stats writes stats3 = {sum, sum of squares, count} (f64, device); ranks all-reduce
 * stats3 between the two calls.  uav_adv_normalise: adv_out = (adv-mean)/(std+1e-6) with the
 * unbiased std and the reference's guard (std<1e-6 or NaN -> 1), ret_out = adv_out + val. */
int uav_adv_stats(uav_ctx* ctx, const float* adv, int64_t n, double* stats3, uav_stream stream);
int uav_adv_normalise(uav_ctx* ctx, const float* adv, const float* val, int64_t n,
                      const double* stats3, float* adv_out, float* ret_out, uav_stream stream);

/* ---- T1 input (the curriculum of model.py:131-164 consumes one success flag per finished episode): the success bits of
 * the episodes that ended in a rollout, compacted in (env, time) order without a host round trip.  flags u8 [n] as written
 * by uav_rollout / uav_env_step (bit0 done, bit1 reached) -> msg u8 [4 + cap + 1]: count (4 bytes, little endian) | bit1 of
 * the k-th ended episode at msg[4 + k] for k < cap | one spare byte.  One fixed-size message per rank: a single all-gather
 * and a single device-to-host copy per iteration feed every rank's replicated curriculum. */
int uav_pack_success_bits(uav_ctx* ctx, const uint8_t* flags, int64_t n, int cap, uint8_t* msg, uav_stream stream);

/* ---- U2: clipped-PPO loss forward + backward in one pass (train_ppo2.0.py:55-83).
 * logits f32 [n][n_act] (pre-softmax), value f32 [n], act i32 [n]; inv_n = 1/(global sample
 * count).  loss_sums (f64 [4], device): {sum -min(s1,s2), sum 0.5*max(.), sum entropy,
 * count of NaN probabilities}.  dlogits [n][n_act], dvalue [n] are d(total)/d(.) where
 * total = policy + value - ent_beta*entropy, each a mean over 1/inv_n samples.  dhead_bias
 * (f32 [n_act+1], or NULL): column sums of (dlogits | dvalue) = gradient of the head biases.
 * Packed form: value == NULL and dvalue == NULL -> `logits` is the heads buffer [n][n_act+1]
 * (logits | value) and `dlogits` receives d(total)/d(heads) in the same [n][n_act+1] layout. */
int uav_ppo_loss(uav_ctx* ctx, const float* logits, const float* value, const int32_t* act,
                 const float* logp_old, const float* adv, const float* ret, const float* val_old,
                 int64_t n, int n_act, float inv_n, float clip, float ent_beta,
                 double* loss_sums, float* dlogits, float* dvalue, float* dhead_bias,
                 uav_stream stream);

/* U2 with the actor/critic heads fused in: heads = y W_head^T + b_head is formed by MFMA straight
 * from the policy trunk's output y [n][hidden] (the LSTM's y), never written to HBM; outputs as the
 * packed form above (dheads [n][n_act+1]).  w_head [n_act+1][hidden], b_head [n_act+1]. */
int uav_ppo_loss_from_y(uav_ctx* ctx, const float* y, const float* w_head, const float* b_head,
                        const int32_t* act, const float* logp_old, const float* adv, const float* ret,
                        const float* val_old, int64_t n, int hidden, int n_act, float inv_n, float clip,
                        float ent_beta, double* loss_sums, float* dheads, float* dhead_bias,
                        uav_stream stream);

/* ---- K3 (sampling part): softmax + Categorical sample + log_prob + NaN check
 * (train_ppo2.0.py:161-163,189; torch Categorical(probs) semantics).  u: uniforms in [0,1)
 * [n] or NULL to use the counter RNG: row i draws from Philox(seed; counter & 0xffffffff, index_offset + i,
 * counter >> 32) -- with counter = (iteration << 32) | step and index_offset = the global index of this rank's
 * env 0 that is uav_rollout's own key, so a job samples the same actions however it is sharded.
 * forced_act: i32 [n] or NULL; when given, act_out = forced_act (parity tests / greedy callers).
 * nan_count: i32[1] device. */
int uav_policy_sample(uav_ctx* ctx, const float* logits, int64_t n, int n_act, const float* u,
                      uint64_t seed, uint64_t counter, int64_t index_offset, const int32_t* forced_act,
                      int32_t* act_out, float* logp_out, float* probs_out, int32_t* nan_count,
                      uav_stream stream);

/* The same draw for time step t of a step-wise rollout, read from and stored into the (env, T) arrays directly: heads =
 * row t of a [n][T][n_act+1] (logits | value) array, i.e. rows heads_stride floats apart; act_out [n] (contiguous: the
 * environment step's input); act_buf / val_buf / logp_buf [n][T] receive column t (PPOBuffer.store's action, value,
 * log_prob: model.py:86-93 at train_ppo2.0.py:192).  Counter RNG only (or forced_act). */
int uav_policy_sample_at(uav_ctx* ctx, const float* heads, int64_t heads_stride, int64_t n, int n_act, uint64_t seed,
                         uint64_t counter, int64_t index_offset, const int32_t* forced_act, int32_t* act_out, int T, int t,
                         int32_t* act_buf, float* val_buf, float* logp_buf, int32_t* nan_count, uav_stream stream);
/* PPOBuffer.store's remaining columns for step t of n envs: keep (the restart mask step t ran with), reward, done, flags
 * -> column t of the [n][T] buffers; keep [n] is then overwritten with the next step's mask 1 - done. */
int uav_store_transition(uav_ctx* ctx, int n, int T, int t, float* keep, const float* rew, const float* done,
                         const uint8_t* flags, float* keep_buf, float* rew_buf, float* done_buf, uint8_t* flags_buf,
                         uav_stream stream);

/* ---- U3: global-norm clip + Adam on one flat f32 buffer (train_ppo2.0.py:87-88,114;
 * torch clip_grad_norm_ / optim.Adam formulas).  `step` is the 1-based optimiser step.
 * gnorm_out: f32[1] device (pre-clip global L2 norm) or NULL.  pmax_out: f32[1] device or NULL: max |param| AFTER the
 * step (the kernel touches every parameter anyway) -- the weight-range probe of uav_set_lstm_arith. */
int uav_clip_adam(uav_ctx* ctx, float* param, const float* grad, float* exp_avg,
                  float* exp_avg_sq, int64_t n, int64_t step, float lr, float beta1, float beta2,
                  float eps, float max_norm, float* gnorm_out, float* pmax_out, uav_stream stream);
/* AdamW (torch.optim.AdamW, PPOV2.0/train_lstm.py:67): as uav_clip_adam with the decoupled decay param *= 1 - lr*weight_decay. */
int uav_clip_adamw(uav_ctx* ctx, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm,
                   float* gnorm_out, uav_stream stream);
/* nn.SmoothL1Loss(beta), reduction mean (train_lstm.py:66), forward + backward: loss_mean f64[1] device,
 * dpred [n] = d(loss)/d(pred). */
int uav_smooth_l1(uav_ctx* ctx, const float* pred, const float* target, int64_t n, float beta, double* loss_mean,
                  float* dpred, uav_stream stream);
/* nn.MSELoss(peak, y_peak) + nn.BCELoss(sigmoid(stop_logit), y_stop), both means (PPOV2.1/train_lstm.py:110-113), forward +
 * backward: out, target, dout f32 [n][2] = (peak, stop_logit) / (y_peak, y_stop) / d(loss)/d(out); loss_mean f64[1]. */
int uav_mse_bce(uav_ctx* ctx, const float* out, const float* target, int64_t n, double* loss_mean, float* dout,
                uav_stream stream);

/* ---- uav_lstm_fwd one time step per call (H = 256, fp16-split arithmetic only): the rollout of a stacked / wide LSTM
 * policy, where an environment step sits between two time steps.  Same kernels and bit-identical results to
 * uav_lstm_fwd over the same inputs; the weights are split once per rollout, the recurrent state stays on the device in
 * the caller-owned `state` buffer (uav_lstm_stepper_bytes(N, I, H) bytes, 256-byte aligned; 0 = shape not supported),
 * and y / stash are the [N][T][H] / [N][T][6H] arrays uav_lstm_bwd reads, filled at time index t.
 *   begin  splits the weights, sets the state to (h0, c0) [N][H]
 *   step   x [N][T][I] (row t read), writes y[:, t], stash[:, t]; at t == T-1 also hn, cn [N][H] (required pointers).
 *          keep_t: NULL, or [N] restart mask of THIS step (uav_lstm_fwd's keep[:, t]: 0 where the env's episode ended in
 *          the previous step -- the state entering the step is zeroed; nn.LSTM has no such mask: it is how the vectorised
 *          rollout restarts the recurrent state, train_ppo2.0.py:157-198 runs one episode at a time).
 *          below: NULL, or the stepper state of the layer below (same N, hidden 256 = this I) already stepped to t: its
 *          h_t is then read from that state's piece planes instead of x (same values: x must still be its y array) */
size_t uav_lstm_stepper_bytes(int N, int I, int H);
int uav_lstm_stepper_begin(uav_ctx* ctx, void* state, const float* w_ih, const float* w_hh, const float* b_ih,
                           const float* b_hh, const float* h0, const float* c0, int N, int I, int H, uav_stream stream);
int uav_lstm_stepper_step(uav_ctx* ctx, void* state, const float* x, const void* below, const float* keep_t, int N, int T,
                          int t, int I, int H, float* y, float* stash, float* hn, float* cn, uav_stream stream);
/* Two INDEPENDENT stepper steps as one launch (their workgroups share the CUs: a step launch is mostly latency, two side by side
 * take ~1.8 x one).  Meant for the forward pass over a stored sequence of a two-layer stack: a = layer 1's step t + 1, b = layer 2's
 * step t with below = layer 1's state -- b reads the piece planes layer 1 wrote at step t, which step t + 1 only reads.  Neither
 * call may read what the other writes.  Same kernels and bit-identical results to the two uav_lstm_stepper_step calls. */
typedef struct uav_stepper_call {
    void* state; const float* x; const void* below; const float* keep_t;
    int t, I;
    float *y, *stash, *hn, *cn;
} uav_stepper_call;
int uav_lstm_stepper_step_pair(uav_ctx* ctx, const uav_stepper_call* a, const uav_stepper_call* b, int N, int T, int H,
                               uav_stream stream);

/* ---- dense f32 building block (exact-f32 MFMA): C[M][N] (+)= op(A)[M][K] * op(B)[K][N] + bias[N].
 * Element (i,k) of op(A) is A[i*sa_m + k*sa_k]; element (k,j) of op(B) is B[k*sb_k + j*sb_n]
 * (strides in elements, so NN / NT / TN are all the same call).  accumulate!=0 adds into C. */
int uav_gemm_f32(uav_ctx* ctx, int64_t M, int64_t N, int64_t K, const float* A, int64_t sa_m,
                 int64_t sa_k, const float* B, int64_t sb_k, int64_t sb_n, float* C, int64_t ldc,
                 const float* bias, int accumulate, uav_stream stream);

/* The same product on the 16-bit matrix pipe at f32 accuracy: every f32 product as three fp16 piece products with f32
 * accumulation (csrc/gemm_h3.hip), for the large shapes: M a multiple of 128, N of 128, each operand with unit stride
 * along k or along its row index (rows 16-byte aligned where k is contiguous); other shapes are refused (use
 * uav_gemm_f32).  Operands must lie inside fp16's range (|a| < 65504); a_absmax, when not NULL, points to a device
 * float holding max |A| (uav_absmax): A is then scaled by one power of two into that range -- what a gradient operand
 * (~1e-6) needs -- and the result scaled back, exactly.  No reference counterpart (it is how uav_lstm_wgrad evaluates
 * dW = dG^T [h_prev | x] and dx = dG W_ih when H is not 64/128). */
int uav_gemm_f16x3(uav_ctx* ctx, int64_t M, int64_t N, int64_t K, const float* A, int64_t sa_m,
                   int64_t sa_k, const float* B, int64_t sb_k, int64_t sb_n, float* C, int64_t ldc,
                   const float* bias, int accumulate, const float* a_absmax, uav_stream stream);

/* out[c] = sum over rows of x[r][c]  (bias gradients; deterministic two-stage reduction).  cols <= 1024. */
int uav_colsum(uav_ctx* ctx, const float* x, int64_t rows, int cols, float* out, uav_stream stream);
/* LayerNorm(cols, eps 1e-5) + ReLU over the rows of z [rows][cols] (cols in 64/128/256/512): z is overwritten with the
 * normalised values, a = relu(z * gamma + beta), rstd [rows].  The nn.LayerNorm -> nn.ReLU pairs of model.py:22-27 and of
 * the stop predictor's head, PPOV2.0/model.py:213-218. */
int uav_ln_relu(uav_ctx* ctx, float* z, float* a, float* rstd, const float* gamma, const float* beta, int64_t rows,
                int cols, uav_stream stream);
/* backward of uav_ln_relu: d [rows][cols] holds dL/da on entry and dL/dz on return; xhat = the z the forward left behind;
 * dgamma, dbeta [cols] (overwritten). */
int uav_ln_relu_bwd(uav_ctx* ctx, float* d, const float* xhat, const float* rstd, const float* gamma, const float* beta,
                    int64_t rows, int cols, float* dgamma, float* dbeta, uav_stream stream);

/* ---- M2: the reference's MLP policy (model.py:17-53), forward and backward.
 * params: flat f32[36230-like] in the order W1[h1][in] b1 g1 be1 W2[h2][h1] b2 g2 be2
 * Whead[n_act+1][h2] bhead[n_act+1] (actor rows then the critic row).
 * stash: f32 [B][h1 + h2 + h1 + h2 + 2] (LN inputs normalised, post-ReLU activations, rstd) or
 * NULL for inference.  heads: f32 [B][n_act+1] = logits | value.
 * uav_mlp_bwd: dheads [B][n_act+1] -> grad (flat, same layout as params; overwritten). */
int64_t uav_mlp_param_count(int in_dim, int h1, int h2, int n_act);
int64_t uav_mlp_stash_floats(int h1, int h2);
int uav_mlp_fwd(uav_ctx* ctx, const float* params, const float* x, int64_t B, int in_dim, int h1,
                int h2, int n_act, float* heads, float* stash, uav_stream stream);
int uav_mlp_bwd(uav_ctx* ctx, const float* params, const float* x, float* stash,
                const float* dheads, int64_t B, int in_dim, int h1, int h2, int n_act,
                float* grad, uav_stream stream);

/* ---- M2 + U2 fused: gradient of the clipped-PPO loss through the reference's MLP policy in ONE pass over the samples
 * (train_ppo2.0.py:55-86: model forward, Categorical log_prob, the three loss terms, backward).  Forward, loss and backward
 * of a 16-sample tile stay on the CU (LayerNorm statistics and activations are never written to HBM); what is read is the
 * 44 algorithmic bytes per sample.  obs [n][6], act i32 [n], logp_old / adv / ret / val_old f32 [n]; inv_n = 1 / (global
 * sample count); loss_sums f64[4] as uav_ppo_loss; grad: flat f32 gradient in the layout of `params` (overwritten;
 * includes the head biases).  The fused kernel is specialised for the reference's sizes (in 6, 256, 128, 5 actions); other
 * sizes take uav_mlp_fwd + uav_ppo_loss + uav_mlp_bwd.
 * Arithmetic: the handle's mode (uav_set_lstm_arith) also governs this kernel and uav_rollout's policy_kind 0.  FP16X3
 * (default): the 256 x 128 layer and its transpose product da1 = W2^T dz2 as three fp16 piece products per f32 product
 * (f32 results; dz2 block-scaled per sample by a power of two); needs max |param| < 2048, so that W2 and the layer's input
 * a1 <= sqrt(255) |g1| + |be1| stay inside fp16's range.  Any other mode: every product on exact-f32 MFMA, no range
 * limit.  The Python trainer measures max |param| (uav_clip_adam's pmax_out) and switches by itself.  Keep ONE mode between
 * a uav_rollout(policy_kind 0) and the uav_mlp_ppo_grad calls that consume its log-probabilities: both run the same forward
 * code in the same arithmetic, which is what makes the first epoch's probability ratio exactly 1. */
int uav_mlp_ppo_grad(uav_ctx* ctx, const float* params, const float* obs, const int32_t* act, const float* logp_old,
                     const float* adv, const float* ret, const float* val_old, int64_t n, int in_dim, int h1, int h2,
                     int n_act, float inv_n, float clip, float ent_beta, double* loss_sums, float* grad,
                     uav_stream stream);

/* ---- L1: nn.LSTM-semantics sequence kernels (gate order i,f,g,o; bias b_ih+b_hh;
 * PPOV2.0/model.py:206-212, PPOV2.1/model.py:263).  One layer per call.
 * x [N][T][I], keep [N][T] (1 = carry the recurrent state into step t, 0 = restart from zero;
 * NULL = all ones), h0,c0 [N][H].  Outputs y [N][T][H], hn,cn [N][H] and, when stash != NULL,
 * the BPTT stash f32 [N][T][6H] = gates(i,f,g,o after activation) | c_prev | h_prev.  The h_prev slot [.., 5H:6H] is opaque scratch:
 * it is written only where uav_lstm_wgrad reads h_prev from it (H other than 64 / 128 / 256, the wide-range arithmetic modes at
 * H = 256, UAV_DEBUG_DG_F32); with I <= 6 at H = 64 / 128 and at H = 256 on the fp16-split arithmetic the weight gradients take
 * h_prev[n][t] = y[n][t-1] keep[n][t] (h0 at t = 0) from y and the slot stays untouched.
 * w_ih [4H][I], w_hh [4H][H], b_ih,b_hh [4H].
 * heads != NULL: also heads [N][T][n_heads] = y W_head^T + b_head (the actor | critic Linear layers of
 * model.py:44,52 applied to the top layer; w_head [n_heads][H], b_head [n_heads], n_heads <= 8), computed inside
 * the sequence kernel where it exists, so the loss (uav_ppo_loss, packed form) reads n_heads floats per sample. */
int uav_lstm_fwd(uav_ctx* ctx, const float* x, const float* keep, const float* h0, const float* c0,
                 const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                 int N, int T, int I, int H, float* y, float* hn, float* cn, float* stash,
                 const float* w_head, const float* b_head, int n_heads, float* heads,
                 uav_stream stream);
/* BPTT sequence kernel.  Gradient of y comes either as dy [N][T][H], or -- for the actor-critic
 * heads, fused -- as dheads [N][T][n_heads] with w_head [n_heads][H] (dy = dheads . w_head is
 * formed in registers, never written to HBM); exactly one of dy / dheads is non-NULL.  dhn,dcn
 * [N][H] or NULL.  Writes dgates (per-step gate gradients, the input of uav_lstm_wgrad: uav_lstm_dgates_bytes(ctx, N, T, H)
 * bytes -- f32 [N][T][4H], except at H = 256 on the fp16-split arithmetic, where the buffer is OPAQUE: the two fp16 pieces per
 * row the recurrent product consumes, scaled per (env, step) by a power of two, in MFMA fragment order + two f32 scales per
 * (env, step) (plain and restart-masked), i.e. 8 bytes per (env, step) more than the f32 rows, N rounded up to 64; uav_lstm_wgrad reads that form
 * directly and uav_lstm_dgates_f32 converts it to f32 rows) and dh0, dc0 [N][H] (or NULL).  The arithmetic mode and debug
 * flags must not change between uav_lstm_bwd and the uav_lstm_wgrad that consumes its dgates. */
int uav_lstm_bwd(uav_ctx* ctx, const float* keep, const float* stash, const float* w_hh,
                 const float* dy, const float* dheads, const float* w_head, int n_heads,
                 const float* dhn, const float* dcn, int N, int T, int H, float* dgates, float* dh0,
                 float* dc0, const float* w_ih, int I, float* dx, uav_stream stream);
/* Bytes of the `dgates` buffer of uav_lstm_bwd / _bwd_stack / _wgrad for this shape under the handle's arithmetic (0 on a bad
 * argument), and its contents as f32 rows out [N][T][4H] (a copy where the buffer already is f32 rows). */
size_t uav_lstm_dgates_bytes(uav_ctx* ctx, int N, int T, int H);
int uav_lstm_dgates_f32(uav_ctx* ctx, const float* dgates, int N, int T, int H, float* out, uav_stream stream);
/* What uav_lstm_bwd can do for this layer shape under the handle's arithmetic (a bit mask):
 *   UAV_BWD_FUSES_DX      dx [N][T][I] = dG W_ih (the gradient of the layer's input, for the layer below) is formed by
 *                         uav_lstm_bwd itself -- its per-step recurrent product multiplies the same dG fragments (H = 256 = I
 *                         on the fp16-split arithmetic): pass w_ih [4H][I] and dx there and dx = NULL to uav_lstm_wgrad.
 *                         Without the bit pass NULL, NULL to uav_lstm_bwd and ask uav_lstm_wgrad for dx.
 *   UAV_BWD_TAKES_DHEADS  the gradient of y may come as dheads + w_head (dy = dheads . w_head formed on chip); without
 *                         the bit form dy with uav_gemm_f32 and pass that. */
#define UAV_BWD_FUSES_DX 1
#define UAV_BWD_TAKES_DHEADS 2
#define UAV_BWD_STACKS 4        /* uav_lstm_bwd_stack is available for layers of this shape (H = 256 = I, fp16-split arithmetic) */
int uav_lstm_bwd_caps(uav_ctx* ctx, int I, int H);
/* The BPTTs of a STACK of LSTM layers (nn.LSTM(num_layers > 1), model.py:206-212) as one pipelined call: layers[0] is the top
 * layer, driven by dy [N][T][H] or dheads + w_head exactly like uav_lstm_bwd; every layer but the last forms the input
 * gradient dx [N][T][H] that drives the layer below (its w_ih [4H][H] and dx are required there; the last layer's may be
 * NULL), and the layer below starts step t as soon as dx[:, t] exists -- each layer runs on its own internal stream one step
 * behind the one above, joined to `stream` before the call returns control of it.  Same kernels and results as one
 * uav_lstm_bwd per layer, top down.  Needs UAV_BWD_STACKS; 1..4 layers; the layers' dgates must be distinct arrays. */
typedef struct uav_lstm_bwd_layer {
    const float* keep;     /* [N][T] or NULL */
    const float* stash;    /* [N][T][6H] */
    const float* w_hh;     /* [4H][H] */
    const float* w_ih;     /* [4H][H]; NULL for the last layer */
    float* dgates;         /* out: uav_lstm_dgates_bytes(ctx, N, T, H) bytes */
    float* dx;             /* [N][T][H] out; NULL for the last layer */
    const float* dhn;      /* [N][H] or NULL */
    const float* dcn;      /* [N][H] or NULL */
    float* dh0;            /* [N][H] out or NULL */
    float* dc0;            /* [N][H] out or NULL */
} uav_lstm_bwd_layer;
int uav_lstm_bwd_stack(uav_ctx* ctx, int n_layers, const uav_lstm_bwd_layer* layers, const float* dy, const float* dheads,
                       const float* w_head, int n_heads, int N, int T, int H, uav_stream stream);
/* Time-batched weight gradients from dgates in ONE fused pass (csrc/wgrad.hip):
 * dw_ih [4H][I] = dG^T X, dw_hh [4H][H] = dG^T Hprev with Hprev[n][t] = y[n][t-1]*keep[n][t]
 * (h0[n]*keep[n][0] at t = 0), db [4H] (= db_ih = db_hh; db_hh, when non-NULL, receives the same values: nn.LSTM's second
 * bias gradient, so the caller needs no copy launch) and -- when dheads != NULL (top layer) --
 * dw_head [n_heads][H] = dheads^T y.  dx [N][T][I] = dG W_ih when non-NULL.  y [N][T][H] is this
 * layer's forward output.  I > 6 (stacked layers) takes generic split-K GEMMs and needs `stash`.
 * Range note: the default kernels of uav_lstm_fwd / _bwd / _wgrad / uav_rollout evaluate their matrix products as fp16
 * piece products at f32 accuracy (csrc/common.h, split2h); they assume |w| < 65504 for the recurrent and head weights and,
 * in the I <= 6 weight-gradient kernel, |x| < 4096 and |h0| < 64 (observations and hidden states are O(1)).  UAV_ARITH_BF16X6 (uav_set_lstm_arith) selects bf16 piece products
 * (f32's exponent range, twice the matrix work), UAV_ARITH_F32_MFMA the exact-f32 MFMA kernels; at H = 256 both take the
 * generic exact-f32 step path (the fp16-split step kernels have no wide-range twin). */
int uav_lstm_wgrad(uav_ctx* ctx, const float* x, const float* keep, const float* h0, const float* y,
                   const float* stash, const float* dgates, const float* w_ih, const float* dheads,
                   int n_heads, int N, int T, int I, int H, float* dw_ih, float* dw_hh, float* db, float* db_hh,
                   float* dw_head, float* dx, uav_stream stream);

/* ---- E1-E5: vectorised plume environment (environment.py:19-169).  State lives in one
 * caller-owned device blob of uav_env_state_bytes(n_env) bytes. */
typedef struct uav_env_cfg {
    int32_t variant;        /* UAV_ENV_*                                                    */
    int32_t field_mode;     /* UAV_FIELD_*                                                  */
    int32_t n_fields;       /* F (materialised mode)                                        */
    int32_t bonus_is_f64;   /* explore_bonus became np.float64 (model.py:142): f64 arithmetic */
    int32_t env_offset;     /* global index of this rank's env 0 (RNG key / field choice)   */
    int32_t n_env_total;    /* envs over all ranks; episode k of env e uses field (e+k*total)%F */
    int32_t trend_k;        /* extra observation channels obs[6+i] = obs[2](t) - obs[2](t-1-i), i < trend_k <= 2
                               (BASELINE C5 'trend obs'); every obs buffer then has 6 + trend_k features */
    int32_t pad_;
    double  radius;         /* current_radius  (model.py:132)                               */
    double  bonus;          /* explore_bonus   (model.py:133)                               */
    uint64_t seed;          /* counter-RNG key (procedural fields, sources, step noise)     */
    const double* bank;     /* [F][500][500][2] (conc, tke) f64, materialised mode          */
    const double* bank_src; /* [F][2] source positions, materialised mode                   */
    const void*   curriculum; /* device pointer to a uav_curriculum state, or NULL: when set, every env kernel reads
                                 radius / bonus / bonus_is_f64 from it at launch and ignores the three fields above -- the
                                 curriculum then never crosses to the host (uav_curriculum_update) */
} uav_env_cfg;

/* ---- T1 on the device: PPOTrainer.update (model.py:131-164) for all episodes that ended in a rollout, fed by the messages of
 * uav_pack_success_bits.  `state` is an opaque device block of uav_curriculum_state_bytes() bytes; its first four doubles are
 * { current_radius, explore_bonus, bonus_is_f64 (0 / 1), overflow flag }, then int64 { episodes seen, successes seen }, then
 * int32 { window length, successes in the window }.  uav_curriculum_update walks msgs[world][4 + cap + 1] rank by rank, episode
 * by episode, in ONE thread (the window logic is sequential; a few thousand episodes cost tens of microseconds on a side stream). */
/* ---- N1 on the device: the per-episode sums behind the reference's 11 CSV columns (train_ppo2.0.py:129-135, filled at
 * :177-183, :200-206, :230-242) for every episode that ENDED in a rollout.  rew f32 [n_env][T], info f32 [n_env][T][10] (5 reward
 * parts | obs[2] | ..), flags u8 [n_env][T] (bit0 done, bit1 reached).  carry f64 [n_env][8]: running sums of { total reward, the 5
 * parts } and the step count of the episode in progress, updated in place (episodes span rollouts; zero it once).  Every ended
 * episode appends one row of 12 doubles { global env index, t, total, conc, explore, move, tke, boundary, steps, success,
 * final_conc (obs[2] x 100 when reached, else 0), 0 } to rows [cap][12] in ARBITRARY order (count[0] = number appended, int32, zero
 * it first; rows beyond cap are counted but dropped): a caller sorts by (env, t).  Sums are f64 and sequential in t, like a host loop. */
int uav_episode_rows(uav_ctx* ctx, const float* rew, const float* info, const uint8_t* flags, int n_env, int T, int env_offset,
                     double* carry, double* rows, int cap, int32_t* count, uav_stream stream);
size_t uav_curriculum_state_bytes(void);
int uav_curriculum_init(uav_ctx* ctx, void* state, double radius, double bonus, int bonus_is_f64, uav_stream stream);
int uav_curriculum_update(uav_ctx* ctx, void* state, const uint8_t* msgs, int world, int cap, uav_stream stream);

size_t uav_env_state_bytes(int n_env);
/* reset every env (environment.py:41-49); obs_out f32 [n_env][6 + trend_k] */
int uav_env_reset(uav_ctx* ctx, void* state, int n_env, const uav_env_cfg* cfg /*host*/,
                  float* obs_out, uav_stream stream);
/* one step of every env with auto-reset (environment.py:82-169 + the reset of
 * train_ppo2.0.py:139).  act i32 [n]; noise f64 [n][2] standard normals or NULL (counter RNG).
 * obs_out [n][6+trend_k] = next state to act on; rew f32 [n]; done f32 [n]; flags u8 [n] (bit0 done,
 * bit1 reached); info f32 [n][5] or NULL; term_obs [n][6+trend_k] or NULL (obs of the ended step). */
int uav_env_step(uav_ctx* ctx, void* state, int n_env, const uav_env_cfg* cfg /*host*/,
                 const int32_t* act, const double* noise, float* obs_out, float* rew, float* done,
                 uint8_t* flags, float* info, float* term_obs, double* rew64, uav_stream stream);
/* copy out per-env state for tests / the drop-in attributes: pos f32 [n][2], source f64 [n][2],
 * steps i32 [n], episode i32 [n] (any may be NULL) */
int uav_env_peek(uav_ctx* ctx, const void* state, int n_env, float* pos, double* source,
                 int32_t* steps, int32_t* episode, uav_stream stream);

/* the 500x500 (conc, tke) f64 tables of env `env_index`'s CURRENT episode, i.e. what the reference
 * holds in env.conc_field / env.tke_field (environment.py:61-62); field_out f64 [500][500][2] */
int uav_env_materialise(uav_ctx* ctx, const void* state, int n_env, const uav_env_cfg* cfg /*host*/,
                        int env_index, double* field_out, uav_stream stream);

/* ---- R1: fused persistent rollout (train_ppo2.0.py:157-198 for n_env environments):
 * policy step + sample + env step + store, T steps in one launch; one workgroup owns a tile
 * of envs and keeps h/c in LDS/registers across the time loop.
 * policy_kind 0 = the reference's MLP 6-256-128 (params as uav_mlp_fwd; h, c, keep, stash, y_out unused / NULL; `hidden`
 * ignored), 1 = single-layer LSTM (params: w_ih w_hh b_ih b_hh Whead bhead).  Buffers (env,T,.) : obs [N][T][6], act i32, rew, val, logp, done f32
 * [N][T], flags u8 [N][T].  cur_obs [N][6] in/out (state to act on), h,c [N][H] in/out (LSTM),
 * keep [N][T] out (LSTM: 0 where the state restarted), last_val [N] out or NULL (V of the state
 * after the last step, for UAV_GAE_STANDARD).  forced_act i32 [N][T] / noise f64 [N][T][2] are
 * NULL outside parity tests.  stash [N][T][6H] + y_out [N][T][H] (both or neither): the BPTT stash of
 * uav_lstm_fwd for exactly this rollout, so the first PPO epoch (same parameters) skips its forward.
 * info (or NULL) f32 [N][T][10]: the five reward parts of environment.py:161-167 (concentration, explore,
 * move, tke, boundary), obs[2] of the step, agent_pos (x, y) after the move -- what train_ppo2.0.py:166-183,203
 * accumulates / logs per episode -- and source_pos (x, y) of the episode the step belongs to (gaussian_params mu_x / mu_y,
 * which PPOV2.1/train_ppo2.0.py:222-232 logs for every episode).
 * heads (or NULL) f32 [N][T][n_act+1]: logits | value of every step (with y_out: epoch 0 of the update needs no
 * forward pass and no head product). */
int uav_rollout(uav_ctx* ctx, void* env_state, int n_env, const uav_env_cfg* cfg /*host*/,
                int policy_kind, const float* params, int hidden, int horizon, uint64_t iter,
                float* cur_obs, float* h, float* c, float* obs, int32_t* act, float* rew,
                float* val, float* logp, float* done, uint8_t* flags, float* keep, float* last_val,
                const int32_t* forced_act, const double* noise, int32_t* nan_count,
                float* stash, float* y_out, float* info, float* heads, uav_stream stream);

/* The tail of step t of a step-wise rollout as ONE launch (train_ppo2.0.py:165-198 after the recurrent layers): policy heads of
 * the top layer (heads[:, t] = y_t W_head^T + b_head, the sums of uav_gemm_f32's few-column kernel bit for bit; y = row t of a
 * [n][T][hidden] array given as the pointer to y[0][t] and its row stride y_stride floats, likewise heads / heads_stride),
 * uav_policy_sample_at's draw (same key: seed, counter = iteration << 32 | t, index_offset + env), uav_env_step's step with
 * auto-reset (cfg, noise [n][2] or NULL, as there), uav_store_transition's columns (keep [n] in / out), and the observation
 * step t + 1 starts from: cur_obs [n][obs_dim] and, while t + 1 < T, obs_seq[:, t + 1] of the [n][T][obs_dim] array.
 * Replaces five launches that all sit on the step's dependency chain; results identical to calling them one by one. */
int uav_rollout_tail(uav_ctx* ctx, void* env_state, int n_env, const uav_env_cfg* cfg /*host*/, const float* y, int64_t y_stride,
                     int hidden, const float* w_head, const float* b_head, int n_act, float* heads, int64_t heads_stride, int T,
                     int t, uint64_t seed, uint64_t counter, int64_t index_offset, const int32_t* forced_act, const double* noise,
                     int32_t* act_out, float* cur_obs, float* obs_seq, float* keep, int32_t* act_buf, float* val_buf,
                     float* logp_buf, float* keep_buf, float* rew_buf, float* done_buf, uint8_t* flags_buf, int32_t* nan_count,
                     uav_stream stream);

/* ---- K9: the iteration's exchanges over RCCL / xGMI (SURVEY 8b `uav_allreduce`, 8e).  One process per GPU, one communicator per
 * handle; RCCL is bound at run time (dlopen librccl.so.1 -- inside a PyTorch process the copy torch already loaded; $UAV_RCCL_LIB
 * overrides), so the library itself has no link-time dependency on it.  A host that is not PyTorch uses these instead of
 * torch.distributed (INTEGRATION.md "Collectives"); the Python trainer takes them with UAVPPO_COLLECTIVES=abi.
 *   uav_comm_unique_id  rank 0 draws the 128-byte id (host memory) and hands it to the other ranks by any host channel
 *   uav_comm_init       every rank, with the handle's device current: blocks until all `world` ranks have joined
 *   uav_allreduce       in-place SUM of `count` f32 over the ranks on `stream`: the flat gradient before uav_clip_adam.  The loss
 *                       kernels take inv_n = 1 / GLOBAL sample count, so the sum is the mean of train_ppo2.0.py:85 and the
 *                       global-norm clip of :87 that follows it is identical on every rank
 *   uav_allreduce_f64   in-place SUM of f64: uav_adv_stats' (sum, sumsq, count) before uav_adv_normalise (train_ppo2.0.py:35-39)
 *   uav_allgather_bytes recv [world][bytes_per_rank] <- every rank's send [bytes_per_rank]: uav_pack_success_bits' message,
 *                       consumed in rank order by uav_curriculum_update (model.py:131-164 replicated)
 * All are asynchronous on `stream` and ordered with the kernels around them; errors (no RCCL, no communicator, RCCL's own)
 * return non-zero with uav_last_error().  Issue a handle's collectives on ONE stream (or order them with events identically on
 * every rank): RCCL executes them in the order each device reaches them, and two ranks that reach two collectives of one
 * communicator in opposite orders wait for each other (uavppo/trainer.py moves its side-stream all-gather to the main stream
 * under this carrier for that reason). */
#define UAV_COMM_ID_BYTES 128
int uav_rccl_version(int* out /*host*/);
int uav_comm_unique_id(void* id_out /*host, UAV_COMM_ID_BYTES*/);
int uav_comm_init(uav_ctx* ctx, const void* id /*host, UAV_COMM_ID_BYTES*/, int rank, int world);
int uav_comm_world(const uav_ctx* ctx);   /* 0: no communicator */
int uav_comm_rank(const uav_ctx* ctx);    /* -1: no communicator */
int uav_comm_destroy(uav_ctx* ctx);       /* also done by uav_destroy */
int uav_allreduce(uav_ctx* ctx, float* flat_grad, int64_t count, uav_stream stream);
int uav_allreduce_f64(uav_ctx* ctx, double* buf, int64_t count, uav_stream stream);
int uav_allgather_bytes(uav_ctx* ctx, const void* send, void* recv, int64_t bytes_per_rank, uav_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* UAVPPO_H */

// loss.hip -- U2 (clipped-PPO loss fwd+bwd) and the sampling half of K3.
//
// Reference: PPOV2.0/train_ppo2.0.py:55-83 (loss), :161-163,189 (sample + log_prob); the
// Categorical(probs) behaviour restated here (renormalise, clamp to [eps, 1-eps], log) is
// torch.distributions' -- see oracle/ppo_oracle.py:categorical_logp.
// HBM-bound elementwise kernels: loss reads 4*(A+6) B and writes 4*(A+1) B per sample.
#include "common.h"
#include "philox.h"

#include "loss_core.h"

template <int A>
__device__ __forceinline__ void loss_block_epilogue(const LossAcc& a, const float* s_db, double* partial) {
    __shared__ double sm[4];
    const double s_pl = block256_sum(a.pl, sm), s_vl = block256_sum(a.vl, sm);
    const double s_en = block256_sum(a.en, sm), s_nan = block256_sum(a.nan, sm);
    double dbs[A + 1];
#pragma unroll
    for (int k = 0; k <= A; ++k) dbs[k] = block256_sum((double)s_db[k], sm);
    if (threadIdx.x == 0) {
        double* pp = partial + (size_t)LOSS_PSTRIDE * blockIdx.x;
        pp[0] = s_pl; pp[1] = s_vl; pp[2] = s_en; pp[3] = s_nan;
#pragma unroll
        for (int k = 0; k <= A; ++k) pp[4 + k] = dbs[k];
    }
}

template <int A>
__global__ __launch_bounds__(256) void ppo_loss_kernel(
    const float* __restrict__ logits, const float* __restrict__ value, const int32_t* __restrict__ act,
    const float* __restrict__ logp_old, const float* __restrict__ adv, const float* __restrict__ ret,
    const float* __restrict__ val_old, int64_t n, float inv_n, float clip, float beta,
    double* __restrict__ partial, float* __restrict__ dlogits, float* __restrict__ dvalue, int ldl, int ldv) {
    LossAcc acc{0.0, 0.0, 0.0, 0.0};
    float s_db[A + 1];                       // column sums of (dlogits | dvalue) = gradient of the head biases
#pragma unroll
    for (int k = 0; k <= A; ++k) s_db[k] = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float z[A], dz[A];
#pragma unroll
        for (int k = 0; k < A; ++k) z[k] = logits[i * ldl + k];
        const float dV = ppo_sample<A>(z, value[i * ldv], act[i], logp_old[i], adv[i], ret[i], val_old[i], inv_n, clip,
                                       beta, acc, dz);
        dvalue[i * ldv] = dV;
        s_db[A] += dV;
#pragma unroll
        for (int k = 0; k < A; ++k) {
            dlogits[i * ldl + k] = dz[k];
            s_db[k] += dz[k];
        }
    }
    loss_block_epilogue<A>(acc, s_db, partial);
}

// ---- heads fused into the loss: heads = y W_head^T + b_head by MFMA straight from the LSTM output ----
// One wave = 64 samples (4 m-tiles of 16).  Each 16-row tile of y is loaded with fully coalesced
// float4 reads into a padded per-wave LDS tile (fragment-shaped loads straight from global touch 64
// cache lines per instruction and thrash L1: 241 us vs the HBM-bound ~60 us), A-fragments are then
// conflict-free ds_read_b128 (k permuted as in lstm.hip), B-fragments of W_head come from LDS too; the
// 16x16 result tiles are transposed through LDS so that lane l owns sample l, then the same per-sample
// loss math runs.  `heads` is never written to HBM and the separate heads GEMM disappears.
typedef float f32x4_l __attribute__((ext_vector_type(4)));
template <int A, int H>
__global__ __launch_bounds__(256) void ppo_loss_from_y_kernel(
    const float* __restrict__ y, const float* __restrict__ w_head, const float* __restrict__ b_head,
    const int32_t* __restrict__ act, const float* __restrict__ logp_old, const float* __restrict__ adv,
    const float* __restrict__ ret, const float* __restrict__ val_old, int64_t n, float inv_n, float clip, float beta,
    double* __restrict__ partial, float* __restrict__ dheads) {
    constexpr int KS = H / 4, SEG = KS + 4, S = 4 * SEG + 8, NH = A + 1;
    __shared__ __attribute__((aligned(16))) float wbuf[16 * S];
    __shared__ __attribute__((aligned(16))) float ybuf[4][16 * S];   // per wave: one 16-row tile of y
    __shared__ float hd[4][64 * 9];                 // per wave: 64 samples x 8 heads (row padded to 9)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 15, kq = lane >> 4;
    for (int idx = threadIdx.x; idx < 16 * H; idx += 256) {
        const int hdx = idx / H, uu = idx % H;
        wbuf[hdx * S + (uu / KS) * SEG + (uu % KS)] = (hdx < NH) ? w_head[(size_t)hdx * H + uu] : 0.f;
    }
    __syncthreads();
    LossAcc acc{0.0, 0.0, 0.0, 0.0};
    float s_db[A + 1];
#pragma unroll
    for (int k = 0; k <= A; ++k) s_db[k] = 0.f;
    float bh[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) bh[k] = b_head[k];
    const int64_t ntile = (n + 255) / 256;
    for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int64_t row0 = tile * 256 + w * 64;
        f32x4_l c4[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            // coalesced tile load: 16 rows x H floats = 16*H/4 float4, 64 lanes -> H/16 float4 per lane
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < H / 16; ++i) {
                const int idx = i * 64 + lane, row = idx / (H / 4), c = idx % (H / 4);
                const int64_t r = row0 + 16 * mi + row;
                const float4 v = *reinterpret_cast<const float4*>(y + (r < n ? r : n - 1) * H + 4 * c);
                *reinterpret_cast<float4*>(&ybuf[w][row * S + ((4 * c) / KS) * SEG + (4 * c) % KS]) = v;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): tile written (single wave, LDS in order)
            __builtin_amdgcn_wave_barrier();
            const float* yr = &ybuf[w][j * S + kq * SEG];
            f32x4_l a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
            const float* wr = wbuf + j * S + kq * SEG;
#pragma unroll
            for (int s = 0; s < KS; s += 4) {
                const float4 av = *reinterpret_cast<const float4*>(yr + s);
                const float4 bv = *reinterpret_cast<const float4*>(wr + s);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, a1, 0, 0, 0);
            }
            c4[mi] = a0 + a1;
        }
        // transpose: D[row = 4*kq + r][col = j]  ->  hd[sample][head]
        if (j < 8) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) hd[w][(16 * mi + 4 * kq + r) * 9 + j] = c4[mi][r];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0); single-wave exchange, LDS is in order
        __builtin_amdgcn_wave_barrier();
        const int64_t i = row0 + lane;
        if (i < n) {
            float z[A], dz[A];
#pragma unroll
            for (int k = 0; k < A; ++k) z[k] = hd[w][lane * 9 + k] + bh[k];
            const float V = hd[w][lane * 9 + A] + bh[A];
            const float dV = ppo_sample<A>(z, V, act[i], logp_old[i], adv[i], ret[i], val_old[i], inv_n, clip, beta, acc, dz);
            s_db[A] += dV;
#pragma unroll
            for (int k = 0; k < A; ++k) {
                dheads[i * NH + k] = dz[k];
                s_db[k] += dz[k];
            }
            dheads[i * NH + A] = dV;
        }
        __builtin_amdgcn_wave_barrier();
    }
    loss_block_epilogue<A>(acc, s_db, partial);
}

// one pass over the block partials: thread (column c = tid % 16, row group g = tid / 16) sums rows g, g+64, ... of
// column c (coalesced 128-B rows, independent loads), then the 64 group sums of a column are added in a fixed order
__global__ __launch_bounds__(1024) void loss_final_kernel(const double* __restrict__ partial, int nb, int n_heads,
                                                          double* __restrict__ out4, float* __restrict__ dbias) {
    __shared__ double sm[64][17];
    const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int i = g;
    for (; i + 192 < nb; i += 256) {
        s0 += partial[(size_t)LOSS_PSTRIDE * i + c];
        s1 += partial[(size_t)LOSS_PSTRIDE * (i + 64) + c];
        s2 += partial[(size_t)LOSS_PSTRIDE * (i + 128) + c];
        s3 += partial[(size_t)LOSS_PSTRIDE * (i + 192) + c];
    }
    for (; i < nb; i += 64) s0 += partial[(size_t)LOSS_PSTRIDE * i + c];
    sm[g][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (threadIdx.x < 4 + n_heads) {
        double r = 0.0;
#pragma unroll
        for (int k = 0; k < 64; ++k) r += sm[k][threadIdx.x];
        if (threadIdx.x < 4) out4[threadIdx.x] = r;
        else if (dbias) dbias[threadIdx.x - 4] = (float)r;
    }
}

// ---- sampling ------------------------------------------------------------------------------------
template <int A>
__global__ __launch_bounds__(256) void policy_sample_kernel(
    const float* __restrict__ logits, int64_t n, const float* __restrict__ u, uint64_t seed,
    uint64_t counter, int64_t index_offset, const int32_t* __restrict__ forced, int32_t* __restrict__ act_out,
    float* __restrict__ logp_out, float* __restrict__ probs_out, int32_t* __restrict__ nan_count,
    int64_t stride = A, int T = 0, int t = 0, int32_t* __restrict__ act_buf = nullptr, float* __restrict__ val_buf = nullptr,
    float* __restrict__ logp_buf = nullptr) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float z[A], p[A];
#pragma unroll
    for (int k = 0; k < A; ++k) z[k] = logits[i * stride + k];
    bool bad;
    float qa;
    const int a = sample_categorical<A>(z, p, u ? &u[i] : nullptr, seed, counter, index_offset + i, forced ? &forced[i] : nullptr, qa, bad);
    if (bad) atomicAdd(nan_count, 1);
    if (probs_out) {
#pragma unroll
        for (int k = 0; k < A; ++k) probs_out[i * A + k] = p[k];
    }
    act_out[i] = a;
    const float lp = logf(fminf(fmaxf(qa, F32_EPS), 1.0f - F32_EPS));
    if (logp_out) logp_out[i] = lp;
    if (act_buf) {             // uav_policy_sample_at: straight into the (env, T) rollout buffers; value = the row's next element
        act_buf[i * T + t] = a;
        logp_buf[i * T + t] = lp;
        val_buf[i * T + t] = logits[i * stride + A];
    }
}

// PPOBuffer.store for one time step of N envs (model.py:86-93 as called at train_ppo2.0.py:192): the environment's
// outputs of step t and the restart mask the step ran with go to column t of the (env, T) buffers; keep becomes the
// next step's mask 1 - done.
__global__ __launch_bounds__(256) void store_transition_kernel(int n, int T, int t, float* __restrict__ keep,
                                                               const float* __restrict__ rew, const float* __restrict__ done,
                                                               const uint8_t* __restrict__ flags, float* __restrict__ keep_buf,
                                                               float* __restrict__ rew_buf, float* __restrict__ done_buf,
                                                               uint8_t* __restrict__ flags_buf) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t o = (int64_t)i * T + t;
    const float d = done[i];
    keep_buf[o] = keep[i];
    rew_buf[o] = rew[i];
    done_buf[o] = d;
    flags_buf[o] = flags[i];
    keep[i] = 1.0f - d;
}

// block partials [nb][LOSS_PSTRIDE] -> loss_sums[4] (+ head-bias gradient), for the fused kernels of other files
int launch_loss_final(double* partial, int nb, int n_heads, double* loss_sums, float* dhead_bias, hipStream_t st) {
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(1024), 0, st, partial, nb, n_heads, loss_sums, dhead_bias);
    UAV_LAUNCH_CHECK();
    return 0;
}

extern "C" {

int uav_ppo_loss(uav_ctx* ctx, const float* logits, const float* value, const int32_t* act,
                 const float* logp_old, const float* adv, const float* ret, const float* val_old,
                 int64_t n, int n_act, float inv_n, float clip, float ent_beta, double* loss_sums,
                 float* dlogits, float* dvalue, float* dhead_bias, uav_stream stream) {
    UAV_REQUIRE(ctx && logits && act && logp_old && adv && ret && val_old && loss_sums && dlogits,
                "uav_ppo_loss: NULL argument");
    // value == NULL: packed heads -- logits is [n][n_act+1] (logits | value) and dlogits the same shape
    UAV_REQUIRE((value == nullptr) == (dvalue == nullptr), "uav_ppo_loss: value and dvalue must both be NULL (packed) or both given");
    const int ldl = value ? n_act : n_act + 1, ldv = value ? 1 : n_act + 1;
    if (!value) { value = logits + n_act; dvalue = dlogits + n_act; }
    UAV_REQUIRE(n > 0, "uav_ppo_loss: n=%lld", (long long)n);
    int nb = (int)((n + 255) / 256);
    if (nb > LOSS_BLOCKS) nb = LOSS_BLOCKS;
    double* partial = (double*)ctx->ws;
#define LAUNCH_LOSS(A_)                                                                                  \
    hipLaunchKernelGGL(ppo_loss_kernel<A_>, dim3(nb), dim3(256), 0, as_stream(stream), logits, value,   \
                       act, logp_old, adv, ret, val_old, n, inv_n, clip, ent_beta, partial, dlogits, dvalue, ldl, ldv)
    switch (n_act) {
        case 2: LAUNCH_LOSS(2); break;
        case 3: LAUNCH_LOSS(3); break;
        case 4: LAUNCH_LOSS(4); break;
        case 5: LAUNCH_LOSS(5); break;
        case 6: LAUNCH_LOSS(6); break;
        case 8: LAUNCH_LOSS(8); break;
        default: UAV_REQUIRE(false, "uav_ppo_loss: n_act=%d unsupported", n_act);
    }
#undef LAUNCH_LOSS
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(1024), 0, as_stream(stream), partial, nb, n_act + 1, loss_sums,
                       dhead_bias);
    UAV_LAUNCH_CHECK();
    return 0;
}

int uav_ppo_loss_from_y(uav_ctx* ctx, const float* y, const float* w_head, const float* b_head, const int32_t* act,
                         const float* logp_old, const float* adv, const float* ret, const float* val_old, int64_t n,
                         int hidden, int n_act, float inv_n, float clip, float ent_beta, double* loss_sums, float* dheads,
                         float* dhead_bias, uav_stream stream) {
    UAV_REQUIRE(ctx && y && w_head && b_head && act && logp_old && adv && ret && val_old && loss_sums && dheads,
                "uav_ppo_loss_from_y: NULL argument");
    UAV_REQUIRE(n > 0 && n_act == 5, "uav_ppo_loss_from_y: n=%lld n_act=%d (5 actions supported)", (long long)n, n_act);
    int nb = (int)((n + 255) / 256);
    if (nb > LOSS_BLOCKS) nb = LOSS_BLOCKS;
    double* partial = (double*)ctx->ws;
#define LAUNCH_LY(H_)                                                                                             \
    hipLaunchKernelGGL((ppo_loss_from_y_kernel<5, H_>), dim3(nb), dim3(256), 0, as_stream(stream), y, w_head, b_head, \
                       act, logp_old, adv, ret, val_old, n, inv_n, clip, ent_beta, partial, dheads)
    switch (hidden) {
        case 64: LAUNCH_LY(64); break;
        case 128: LAUNCH_LY(128); break;
        case 256: LAUNCH_LY(256); break;
        default: UAV_REQUIRE(false, "uav_ppo_loss_from_y: hidden=%d unsupported (64/128/256)", hidden);
    }
#undef LAUNCH_LY
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(1024), 0, as_stream(stream), partial, nb, n_act + 1, loss_sums,
                       dhead_bias);
    UAV_LAUNCH_CHECK();
    return 0;
}

int uav_policy_sample(uav_ctx* ctx, const float* logits, int64_t n, int n_act, const float* u,
                      uint64_t seed, uint64_t counter, int64_t index_offset, const int32_t* forced_act, int32_t* act_out,
                      float* logp_out, float* probs_out, int32_t* nan_count, uav_stream stream) {
    UAV_REQUIRE(ctx && logits && act_out && logp_out && nan_count && n > 0, "uav_policy_sample: bad argument");
    const int nb = (int)((n + 255) / 256);
#define LAUNCH_S(A_)                                                                                      \
    hipLaunchKernelGGL(policy_sample_kernel<A_>, dim3(nb), dim3(256), 0, as_stream(stream), logits, n, u, \
                       seed, counter, index_offset, forced_act, act_out, logp_out, probs_out, nan_count)
    switch (n_act) {
        case 2: LAUNCH_S(2); break;
        case 3: LAUNCH_S(3); break;
        case 4: LAUNCH_S(4); break;
        case 5: LAUNCH_S(5); break;
        case 6: LAUNCH_S(6); break;
        case 8: LAUNCH_S(8); break;
        default: UAV_REQUIRE(false, "uav_policy_sample: n_act=%d unsupported", n_act);
    }
#undef LAUNCH_S
    UAV_LAUNCH_CHECK();
    return 0;
}

int uav_policy_sample_at(uav_ctx* ctx, const float* heads, int64_t heads_stride, int64_t n, int n_act, uint64_t seed,
                         uint64_t counter, int64_t index_offset, const int32_t* forced_act, int32_t* act_out, int T, int t,
                         int32_t* act_buf, float* val_buf, float* logp_buf, int32_t* nan_count, uav_stream stream) {
    UAV_REQUIRE(ctx && heads && act_out && act_buf && val_buf && logp_buf && nan_count && n > 0, "uav_policy_sample_at: bad argument");
    UAV_REQUIRE(heads_stride > n_act && T > 0 && t >= 0 && t < T, "uav_policy_sample_at: heads_stride=%lld n_act=%d T=%d t=%d",
                (long long)heads_stride, n_act, T, t);
    const int nb = (int)((n + 255) / 256);
#define LAUNCH_S(A_)                                                                                                         \
    hipLaunchKernelGGL(policy_sample_kernel<A_>, dim3(nb), dim3(256), 0, as_stream(stream), heads, n, (const float*)nullptr, \
                       seed, counter, index_offset, forced_act, act_out, (float*)nullptr, (float*)nullptr, nan_count,       \
                       heads_stride, T, t, act_buf, val_buf, logp_buf)
    switch (n_act) {
        case 2: LAUNCH_S(2); break;
        case 3: LAUNCH_S(3); break;
        case 4: LAUNCH_S(4); break;
        case 5: LAUNCH_S(5); break;
        case 6: LAUNCH_S(6); break;
        case 8: LAUNCH_S(8); break;
        default: UAV_REQUIRE(false, "uav_policy_sample_at: n_act=%d unsupported", n_act);
    }
#undef LAUNCH_S
    UAV_LAUNCH_CHECK();
    return 0;
}

int uav_store_transition(uav_ctx* ctx, int n, int T, int t, float* keep, const float* rew, const float* done,
                         const uint8_t* flags, float* keep_buf, float* rew_buf, float* done_buf, uint8_t* flags_buf,
                         uav_stream stream) {
    UAV_REQUIRE(ctx && keep && rew && done && flags && keep_buf && rew_buf && done_buf && flags_buf, "uav_store_transition: NULL argument");
    UAV_REQUIRE(n > 0 && T > 0 && t >= 0 && t < T, "uav_store_transition: n=%d T=%d t=%d", n, T, t);
    hipLaunchKernelGGL(store_transition_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), n, T, t, keep, rew, done,
                       flags, keep_buf, rew_buf, done_buf, flags_buf);
    UAV_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

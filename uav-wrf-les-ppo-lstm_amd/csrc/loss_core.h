// loss_core.h -- the per-sample arithmetic of the clipped-PPO loss (train_ppo2.0.py:55-83), shared by the elementwise
// loss kernels (loss.hip) and the fused MLP update (mlp_fused.hip), so every path rounds alike.
#pragma once
#include "common.h"
#include "philox.h"

constexpr float F32_EPS = 1.1920928955078125e-07f;

template <int A>
__device__ __forceinline__ void softmax_row(const float* z, float* p) {
    float m = z[0];
#pragma unroll
    for (int k = 1; k < A; ++k) m = fmaxf(m, z[k]);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < A; ++k) {
        p[k] = expf(z[k] - m);
        s += p[k];
    }
#pragma unroll
    for (int k = 0; k < A; ++k) p[k] = p[k] / s;
}

// One Categorical(probs) draw from logits z[A] (train_ppo2.0.py:171-173; torch.multinomial's inverse-CDF semantics over the
// normalised probabilities): p[] = softmax(z), the action (forced, from a given uniform, or from the counter RNG keyed
// (seed; step-in-rollout, GLOBAL env index, iteration) like the fused rollout kernels, so a job draws the same actions however
// its envs are sharded over ranks and whichever rollout path runs it), qa = its normalised probability, bad = a NaN probability.
template <int A>
__device__ __forceinline__ int sample_categorical(const float* z, float* p, const float* u, uint64_t seed, uint64_t counter,
                                                  int64_t env_global, const int32_t* forced, float& qa, bool& bad) {
    softmax_row<A>(z, p);
    float psum = 0.f;
    bad = false;
#pragma unroll
    for (int k = 0; k < A; ++k) {
        psum += p[k];
        bad |= (p[k] != p[k]);
    }
    int a;
    if (forced) {
        a = *forced;
    } else {
        float uu;
        if (u) uu = *u;
        else {
            const Philox4 r = philox4x32_10(seed, (uint32_t)counter, (uint32_t)env_global, (uint32_t)(counter >> 32), RNG_ACTION);
            uu = u01_f32(r.x);
        }
        float cdf = 0.f;
        a = A - 1;
        const float target = uu * psum;
#pragma unroll
        for (int k = 0; k < A; ++k) {
            cdf += p[k];
            if (target < cdf) { a = k; break; }
        }
    }
    qa = 0.f;
#pragma unroll
    for (int k = 0; k < A; ++k)
        if (k == a) qa = p[k] / psum;
    return a;
}

constexpr int LOSS_BLOCKS = 1024;
constexpr int LOSS_PSTRIDE = 16;   // doubles per block partial: 4 loss sums + up to 9 head-bias sums

struct LossAcc { double pl, vl, en, nan; };

// one sample: loss terms into `acc`, d(total)/d(logits) into dz[A], d(total)/dV returned
template <int A>
__device__ __forceinline__ float ppo_sample(const float* z, float V, int a, float lpo, float Ad, float R, float vo,
                                            float inv_n, float clip, float beta, LossAcc& acc, float* dz) {
    float p[A];
    softmax_row<A>(z, p);
    float psum = 0.f;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < A; ++k) {
        psum += p[k];
        bad |= (p[k] != p[k]);
    }
    if (bad) acc.nan += 1.0;
    // Categorical(probs): q = p / sum p, clamped to [eps, 1-eps]
    float q[A];
    float qa = 0.f;
#pragma unroll
    for (int k = 0; k < A; ++k) {
        q[k] = p[k] / psum;
        if (k == a) qa = q[k];
    }
    const bool clamp_open = (qa >= F32_EPS) && (qa <= 1.0f - F32_EPS);
    const float qc = fminf(fmaxf(qa, F32_EPS), 1.0f - F32_EPS);
    const float logp = logf(qc);
    const float ratio = expf(logp - lpo);
    const float lo = 1.0f - clip, hi = 1.0f + clip;
    const float rc = fminf(fmaxf(ratio, lo), hi);
    const float s1 = ratio * Ad, s2 = rc * Ad;
    acc.pl += (double)(-fminf(s1, s2));
    const bool in_rng = (ratio >= lo) && (ratio <= hi);
    // torch.min backward: ties split 1/2 - 1/2; clamp backward passes inside [lo,hi]
    float g_ratio;
    if (s1 < s2) g_ratio = Ad;
    else if (s1 == s2) g_ratio = 0.5f * Ad + (in_rng ? 0.5f * Ad : 0.f);
    else g_ratio = in_rng ? Ad : 0.f;
    const float g_logp = clamp_open ? (-inv_n * g_ratio * ratio) : 0.f;   // dL/dlogp

    // value loss, train_ppo2.0.py:74-78
    const float dv = V - vo;
    const float vc = vo + fminf(fmaxf(dv, -clip), clip);
    const float e1 = (V - R) * (V - R), e2 = (vc - R) * (vc - R);
    acc.vl += (double)(0.5f * fmaxf(e1, e2));
    const float d1 = 2.0f * (V - R);
    const float d2 = (dv >= -clip && dv <= clip) ? 2.0f * (vc - R) : 0.f;
    const float gV = (e1 > e2) ? d1 : ((e1 < e2) ? d2 : 0.5f * (d1 + d2));

    // entropy, train_ppo2.0.py:81 :  H = -sum p log(p + 1e-8)
    float Hs = 0.f, hk[A], ph = 0.f;
#pragma unroll
    for (int k = 0; k < A; ++k) {
        const float lg = logf(p[k] + 1e-8f);
        Hs -= p[k] * lg;
        hk[k] = -lg - p[k] / (p[k] + 1e-8f);      // dH/dp_k
        ph += p[k] * hk[k];
    }
    acc.en += (double)Hs;
    // through the softmax: d/dz_k = p_k (g_k - sum_j p_j g_j)
#pragma unroll
    for (int k = 0; k < A; ++k) {
        const float dpol = g_logp * ((k == a ? 1.0f : 0.0f) - q[k]);
        const float dent = -beta * inv_n * p[k] * (hk[k] - ph);
        dz[k] = dpol + dent;
    }
    return 0.5f * inv_n * gV;
}


// adam.hip -- U3: global-L2-norm gradient clip + Adam over ONE flat f32 parameter buffer.
//
// Reference: PPOV2.0/train_ppo2.0.py:87 (clip_grad_norm_(params, 0.5)), :88 + :114
// (torch.optim.Adam, lr 3e-5, betas (.9,.999), eps 1e-8, no weight decay).
// HBM-bound: 16 B read + 12 B write per parameter; the norm is a deterministic two-stage
// reduction whose scalar result stays on the device (no host round trip between clip and step).
#include "common.h"

constexpr int NORM_BLOCKS = 256;

__global__ __launch_bounds__(256) void sumsq_partial(const float* __restrict__ g, int64_t n,
                                                     double* __restrict__ partial) {
    __shared__ double sm[4];
    double q = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = (double)g[i];
        q += v * v;
    }
    q = block256_sum(q, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = q;
}

// scal[0] = clip coefficient, scal[1] = pre-clip norm
__global__ __launch_bounds__(256) void norm_final(const double* __restrict__ partial, int nb, float max_norm,
                                                  float* __restrict__ scal, float* __restrict__ gnorm_out) {
    __shared__ double sm[4];
    double q = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) q += partial[i];
    q = block256_sum(q, sm);
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(q);
        float coef = max_norm / (norm + 1e-6f);        // torch clip_grad_norm_
        coef = coef > 1.0f ? 1.0f : coef;
        if (!(max_norm > 0.f)) coef = 1.0f;            // max_norm <= 0: clipping disabled
        scal[0] = coef;
        scal[1] = norm;
        if (gnorm_out) *gnorm_out = norm;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   const float* __restrict__ scal, float b1, float b2,
                                                   float step_size, float sqrt_bc2, float eps) {
#pragma clang fp contract(off)
    const float coef = scal[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * coef;
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);          // exp_avg.lerp_(grad, 1-beta1)
        const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / sqrt_bc2 + eps;      // (exp_avg_sq.sqrt() / sqrt(bc2)).add_(eps)
        p[i] = p[i] - step_size * (mi / denom);
    }
}

extern "C" int uav_clip_adam(uav_ctx* ctx, float* param, const float* grad, float* exp_avg,
                             float* exp_avg_sq, int64_t n, int64_t step, float lr, float beta1, float beta2,
                             float eps, float max_norm, float* gnorm_out, uav_stream stream) {
    UAV_REQUIRE(ctx && param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "uav_clip_adam: bad argument");
    double* partial = (double*)ctx->ws;
    float* scal = (float*)((char*)ctx->ws + 4096);
    int nb = (int)((n + 1023) / 1024);
    if (nb > NORM_BLOCKS) nb = NORM_BLOCKS;
    hipLaunchKernelGGL(sumsq_partial, dim3(nb), dim3(256), 0, as_stream(stream), grad, n, partial);
    hipLaunchKernelGGL(norm_final, dim3(1), dim3(256), 0, as_stream(stream), partial, nb, max_norm, scal, gnorm_out);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float sqrt_bc2 = (float)sqrt(bc2);
    int ab = (int)((n + 255) / 256);
    if (ab > 2048) ab = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3(ab), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq, n,
                       scal, beta1, beta2, step_size, sqrt_bc2, eps);
    UAV_LAUNCH_CHECK();
    return 0;
}

// adam.hip -- U3: global-L2-norm gradient clip + Adam over ONE flat f32 parameter buffer.
//
// Reference: PPOV2.0/train_ppo2.0.py:87 (clip_grad_norm_(params, 0.5)), :88 + :114
// (torch.optim.Adam, lr 3e-5, betas (.9,.999), eps 1e-8, no weight decay).
// HBM-bound: 16 B read + 12 B write per parameter; the norm is a deterministic two-stage
// reduction whose scalar result stays on the device (no host round trip between clip and step).
#include "common.h"

constexpr int NORM_BLOCKS = 256;

__global__ __launch_bounds__(256) void sumsq_partial(const float* __restrict__ g, int64_t n,
                                                     double* __restrict__ partial, unsigned* __restrict__ pmax_bits) {
    __shared__ double sm[4];
    if (pmax_bits && blockIdx.x == 0 && threadIdx.x == 0) *pmax_bits = 0u;      // adam_kernel (next launch) maxes into it
    double q = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = (double)g[i];
        q += v * v;
    }
    q = block256_sum(q, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = q;
}

// decay = 1 - lr * weight_decay (torch.optim.AdamW: param.mul_(decay) before the Adam update; 1 for plain Adam).
// Every block first finishes the norm itself from the (<= 256) partial sums -- the same fixed-order reduction in each, so
// all blocks get the same bits -- instead of a one-block kernel launch in between; block 0 publishes the norm.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   const double* __restrict__ partial, int nb, float max_norm,
                                                   float* __restrict__ gnorm_out, float b1, float b2,
                                                   float step_size, float sqrt_bc2, float eps, float decay,
                                                   unsigned* __restrict__ pmax_bits) {
#pragma clang fp contract(off)
    __shared__ double sm[4];
    __shared__ float coef_s;
    double q = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) q += partial[i];
    q = block256_sum(q, sm);
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(q);
        float coef = max_norm / (norm + 1e-6f);        // torch clip_grad_norm_
        coef = coef > 1.0f ? 1.0f : coef;
        if (!(max_norm > 0.f)) coef = 1.0f;            // max_norm <= 0: clipping disabled
        coef_s = coef;
        if (gnorm_out && blockIdx.x == 0) *gnorm_out = norm;
    }
    __syncthreads();
    const float coef = coef_s;
    float amax = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * coef;
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);          // exp_avg.lerp_(grad, 1-beta1)
        const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / sqrt_bc2 + eps;      // (exp_avg_sq.sqrt() / sqrt(bc2)).add_(eps)
        const float pn = p[i] * decay - step_size * (mi / denom);
        p[i] = pn;
        amax = fmaxf(amax, pn == pn ? fabsf(pn) : __builtin_inff());      // a NaN parameter reads as "out of range"
    }
    if (pmax_bits) {       // max |param| after this step (bit patterns of non-negative floats order like the floats)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        __shared__ float wmax[4];
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
        __syncthreads();
        if (threadIdx.x == 0)      // ONE atomic per block: a few hundred to one address, not a few thousand
            atomicMax(pmax_bits, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
    }
}

// out[0] = max |x[i]| (NaN counts as +inf); *out zeroed by the memset in front of the launch.
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ out) {
    float amax = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = x[i];
        amax = fmaxf(amax, v == v ? fabsf(v) : __builtin_inff());
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(amax));
}

extern "C" int uav_absmax(uav_ctx* ctx, const float* x, int64_t n, float* out, uav_stream stream) {
    UAV_REQUIRE(ctx && x && out && n > 0, "uav_absmax: bad argument");
    UAV_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float), as_stream(stream)));
    int nb = (int)((n + 4095) / 4096);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(absmax_kernel, dim3(nb), dim3(256), 0, as_stream(stream), x, n, reinterpret_cast<unsigned*>(out));
    UAV_LAUNCH_CHECK();
    return 0;
}

static int clip_adam_impl(uav_ctx* ctx, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                          int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm,
                          float* gnorm_out, float* pmax_out, uav_stream stream) {
    UAV_REQUIRE(ctx && param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "uav_clip_adam: bad argument");
    unsigned* pmax_bits = reinterpret_cast<unsigned*>(pmax_out);
    double* partial = (double*)ctx->ws;
    int nb = (int)((n + 1023) / 1024);
    if (nb > NORM_BLOCKS) nb = NORM_BLOCKS;
    hipLaunchKernelGGL(sumsq_partial, dim3(nb), dim3(256), 0, as_stream(stream), grad, n, partial, pmax_bits);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float sqrt_bc2 = (float)sqrt(bc2);
    int ab = (int)((n + 255) / 256);
    if (ab > 2048) ab = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3(ab), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq, n,
                       partial, nb, max_norm, gnorm_out, beta1, beta2, step_size, sqrt_bc2, eps, 1.0f - lr * weight_decay, pmax_bits);
    UAV_LAUNCH_CHECK();
    return 0;
}

extern "C" int uav_clip_adam(uav_ctx* ctx, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                             int64_t step, float lr, float beta1, float beta2, float eps, float max_norm, float* gnorm_out,
                             float* pmax_out, uav_stream stream) {
    return clip_adam_impl(ctx, param, grad, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, 0.f, max_norm, gnorm_out,
                          pmax_out, stream);
}

extern "C" int uav_clip_adamw(uav_ctx* ctx, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                              int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm,
                              float* gnorm_out, uav_stream stream) {
    return clip_adam_impl(ctx, param, grad, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, weight_decay, max_norm,
                          gnorm_out, nullptr, stream);
}

// SmoothL1Loss(beta), reduction = mean (train_lstm.py:66): loss_sum[0] += sum_i l(pred_i - target_i) (f64, one block, fixed
// order), dpred_i = dl/dpred_i / n.  |d| < beta: 0.5 d^2 / beta, else |d| - 0.5 beta.
__global__ __launch_bounds__(256) void smooth_l1_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                        int64_t n, float beta, double* __restrict__ loss_mean,
                                                        float* __restrict__ dpred) {
    __shared__ double sm[4];
    double s = 0.0;
    const float inv_n = 1.0f / (float)n;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const float d = pred[i] - target[i], a = fabsf(d);
        if (a < beta) { s += 0.5 * (double)d * d / beta; dpred[i] = d / beta * inv_n; }
        else { s += (double)a - 0.5 * beta; dpred[i] = (d > 0.f ? 1.f : -1.f) * inv_n; }
    }
    const double r = block256_sum(s, sm);
    if (threadIdx.x == 0) loss_mean[0] = r / (double)n;
}

extern "C" int uav_smooth_l1(uav_ctx* ctx, const float* pred, const float* target, int64_t n, float beta, double* loss_mean,
                             float* dpred, uav_stream stream) {
    UAV_REQUIRE(ctx && pred && target && loss_mean && dpred && n > 0 && beta > 0.f, "uav_smooth_l1: bad argument");
    hipLaunchKernelGGL(smooth_l1_kernel, dim3(1), dim3(256), 0, as_stream(stream), pred, target, n, beta, loss_mean, dpred);
    UAV_LAUNCH_CHECK();
    return 0;
}

// PPOV2.1/train_lstm.py:110-113: loss = MSELoss(peak, y_peak) + BCELoss(sigmoid(stop_logit), y_stop), both means.
// out [n][2] = (peak, stop_logit); target [n][2]; dout [n][2] = d(loss)/d(out).  BCELoss clamps its logs at -100 (torch).
__global__ __launch_bounds__(256) void mse_bce_kernel(const float* __restrict__ out, const float* __restrict__ target, int64_t n,
                                                      double* __restrict__ loss_mean, float* __restrict__ dout) {
    __shared__ double sm[4];
    double s = 0.0;
    const float inv_n = 1.0f / (float)n;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const float d = out[2 * i] - target[2 * i];
        const float z = out[2 * i + 1], yb = target[2 * i + 1];
        const float p = 1.0f / (1.0f + expf(-z));
        const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.0f - p), -100.f);
        s += (double)d * d - ((double)yb * lp + (1.0 - (double)yb) * lq);
        dout[2 * i] = 2.0f * d * inv_n;
        dout[2 * i + 1] = (p - yb) * inv_n;
    }
    const double r = block256_sum(s, sm);
    if (threadIdx.x == 0) loss_mean[0] = r / (double)n;
}

extern "C" int uav_mse_bce(uav_ctx* ctx, const float* out, const float* target, int64_t n, double* loss_mean, float* dout,
                           uav_stream stream) {
    UAV_REQUIRE(ctx && out && target && loss_mean && dout && n > 0, "uav_mse_bce: bad argument");
    hipLaunchKernelGGL(mse_bce_kernel, dim3(1), dim3(256), 0, as_stream(stream), out, target, n, loss_mean, dout);
    UAV_LAUNCH_CHECK();
    return 0;
}

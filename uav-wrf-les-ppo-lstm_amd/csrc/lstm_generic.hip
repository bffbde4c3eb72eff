// lstm_generic.hip -- nn.LSTM forward/backward for ANY hidden size (e.g. BASELINE C5's h=256):
// the classic time-major decomposition, one recurrent GEMM (gemm.hip, exact-f32 MFMA) plus one
// pointwise kernel per time step.  Functional coverage path: the VGPR-resident persistent kernels of
// lstm.hip serve H in {64,128}; W_hh of larger layers (1 MB at H=256) does not fit one workgroup's
// registers, and an L2-streaming persistent kernel for them is future work.
// Same stash layout and semantics as lstm.hip (gates i,f,g,o | c_prev | h_prev per (n,t)).
#include "common.h"

int gemm_f32(uav_ctx* ctx, int64_t M, int64_t N, int64_t K, const float* A, int64_t sa_m, int64_t sa_k,
             const float* B, int64_t sb_k, int64_t sb_n, float* C, int64_t ldc, const float* bias,
             int accumulate, hipStream_t st);

// state <- init * keep[:,0]
__global__ void gen_init_state(const float* __restrict__ h0, const float* __restrict__ c0, const float* __restrict__ keep,
                               int N, int T, int H, float* __restrict__ hs, float* __restrict__ cs) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)N * H) return;
    const int n = (int)(i / H);
    const float k = keep ? keep[(size_t)n * T] : 1.f;
    hs[i] = h0[i] * k;
    cs[i] = c0[i] * k;
}

// gates of step t (pre-activations in the stash) -> activations, c_t, h_t; state <- masked for step t+1
__global__ void gen_cell_fwd(float* __restrict__ stash, const float* __restrict__ keep, int N, int T, int H, int t,
                             float* __restrict__ hs, float* __restrict__ cs, float* __restrict__ y,
                             float* __restrict__ hn, float* __restrict__ cn) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)N * H) return;
    const int n = (int)(i / H), u = (int)(i % H);
    float* sp = stash + ((size_t)n * T + t) * (6 * H);
    const float gi = fast_sigmoid(sp[u]), gf = fast_sigmoid(sp[H + u]);
    const float gg = fast_tanh(sp[2 * H + u]), go = fast_sigmoid(sp[3 * H + u]);
    const float cp = cs[i], hp = hs[i];
    const float c = gf * cp + gi * gg;
    const float h = go * fast_tanh(c);
    sp[u] = gi; sp[H + u] = gf; sp[2 * H + u] = gg; sp[3 * H + u] = go;
    sp[4 * H + u] = cp;
    sp[5 * H + u] = hp;
    y[((size_t)n * T + t) * H + u] = h;
    if (t == T - 1) {
        hn[i] = h;
        cn[i] = c;
    } else {
        const float kn = keep ? keep[(size_t)n * T + t + 1] : 1.f;
        hs[i] = h * kn;
        cs[i] = c * kn;
    }
}

// dh = dy_t + dh_rec ; gate gradients of step t ; dc_next, and dh_rec scaled for the mask of step t
__global__ void gen_cell_bwd(const float* __restrict__ stash, const float* __restrict__ keep, const float* __restrict__ dy,
                             int N, int T, int H, int t, const float* __restrict__ dh_rec, float* __restrict__ dc_next,
                             float* __restrict__ dgates) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)N * H) return;
    const int n = (int)(i / H), u = (int)(i % H);
    const size_t row = (size_t)n * T + t;
    const float* sp = stash + row * (6 * H);
    const float gi = sp[u], gf = sp[H + u], gg = sp[2 * H + u], go = sp[3 * H + u], cp = sp[4 * H + u];
    const float dh = dy[row * H + u] + dh_rec[i];
    const float c = gf * cp + gi * gg;
    const float tch = fast_tanh(c);
    const float dc = dh * go * (1.0f - tch * tch) + dc_next[i];
    float* gp = dgates + row * (4 * H);
    gp[u] = dc * gg * gi * (1.0f - gi);
    gp[H + u] = dc * cp * gf * (1.0f - gf);
    gp[2 * H + u] = dc * gi * (1.0f - gg * gg);
    gp[3 * H + u] = dh * tch * go * (1.0f - go);
    const float kp = keep ? keep[row] : 1.f;
    dc_next[i] = dc * gf * kp;
}

// x *= keep[:, t]  (gradient of the masked incoming state)
__global__ void gen_mask_rows(float* __restrict__ x, const float* __restrict__ keep, int N, int T, int H, int t) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)N * H) return;
    x[i] *= keep[(size_t)(i / H) * T + t];
}

__global__ void gen_fill(float* __restrict__ x, const float* __restrict__ src, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = src ? src[i] : 0.f;
}

// pre-activations of all steps must already be in the gates slot of the stash (x W_ih^T + b)
int lstm_generic_fwd(uav_ctx* ctx, const float* keep, const float* h0, const float* c0, const float* w_hh, int N, int T,
                     int H, float* y, float* hn, float* cn, float* stash, hipStream_t st) {
    const int64_t NH = (int64_t)N * H;
    UAV_REQUIRE((size_t)(2 * NH) * sizeof(float) + (64u << 20) <= ctx->ws_bytes, "lstm (generic): workspace too small");
    float* hs = (float*)((char*)ctx->ws + ctx->ws_bytes) - 2 * NH;     // recurrent state at the tail of the workspace
    float* cs = hs + NH;
    uav_ctx sub = *ctx;
    sub.ws_bytes = ctx->ws_bytes - 2 * NH * sizeof(float);
    const unsigned nb = (unsigned)((NH + 255) / 256);
    hipLaunchKernelGGL(gen_init_state, dim3(nb), dim3(256), 0, st, h0, c0, keep, N, T, H, hs, cs);
    for (int t = 0; t < T; ++t) {
        // gates_t += h_{t-1} W_hh^T   (rows = envs, row stride T*6H inside the stash)
        int rc = gemm_f32(&sub, N, 4 * H, H, hs, H, 1, w_hh, 1, H, stash + (size_t)t * 6 * H, (int64_t)T * 6 * H, nullptr, 1, st);
        if (rc) return rc;
        hipLaunchKernelGGL(gen_cell_fwd, dim3(nb), dim3(256), 0, st, stash, keep, N, T, H, t, hs, cs, y, hn, cn);
    }
    UAV_LAUNCH_CHECK();
    return 0;
}

int lstm_generic_bwd(uav_ctx* ctx, const float* keep, const float* stash, const float* w_hh, const float* dy,
                     const float* dhn, const float* dcn, int N, int T, int H, float* dgates, float* dh0, float* dc0,
                     hipStream_t st) {
    const int64_t NH = (int64_t)N * H;
    UAV_REQUIRE((size_t)(2 * NH) * sizeof(float) + (64u << 20) <= ctx->ws_bytes, "lstm (generic): workspace too small");
    float* dh = (float*)((char*)ctx->ws + ctx->ws_bytes) - 2 * NH;
    float* dc = dh + NH;
    uav_ctx sub = *ctx;
    sub.ws_bytes = ctx->ws_bytes - 2 * NH * sizeof(float);
    const unsigned nb = (unsigned)((NH + 255) / 256);
    hipLaunchKernelGGL(gen_fill, dim3(nb), dim3(256), 0, st, dh, dhn, NH);
    hipLaunchKernelGGL(gen_fill, dim3(nb), dim3(256), 0, st, dc, dcn, NH);
    for (int t = T - 1; t >= 0; --t) {
        hipLaunchKernelGGL(gen_cell_bwd, dim3(nb), dim3(256), 0, st, stash, keep, dy, N, T, H, t, dh, dc, dgates);
        // dh_{t-1} = dgates_t W_hh  ([N,4H] x [4H,H]); then the mask of step t
        int rc = gemm_f32(&sub, N, H, 4 * H, dgates + (size_t)t * 4 * H, (int64_t)T * 4 * H, 1, w_hh, H, 1, dh, H, nullptr, 0, st);
        if (rc) return rc;
        if (keep) hipLaunchKernelGGL(gen_mask_rows, dim3(nb), dim3(256), 0, st, dh, keep, N, T, H, t);
    }
    if (dh0) hipLaunchKernelGGL(gen_fill, dim3(nb), dim3(256), 0, st, dh0, dh, NH);
    if (dc0) hipLaunchKernelGGL(gen_fill, dim3(nb), dim3(256), 0, st, dc0, dc, NH);
    UAV_LAUNCH_CHECK();
    return 0;
}

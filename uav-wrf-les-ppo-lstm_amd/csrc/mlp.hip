// mlp.hip -- M2: the reference's MLP actor-critic, forward and backward.
//
// Reference: PPOV2.0/model.py:17-53  Linear(in,256)->LayerNorm->ReLU->Linear(256,128)->LayerNorm
// ->ReLU->actor Linear(128,A) | critic Linear(128,1).  The Linear layers run on gemm.hip
// (exact-f32 MFMA); LayerNorm+ReLU forward/backward are one-wave-per-row kernels (row fits one
// wave's registers, reductions by shuffles), parameter gradients by deterministic two-stage
// column reductions.  Flat parameter layout (heads contiguous so actor|critic is ONE GEMM):
//   W1[h1][in] b1[h1] g1[h1] be1[h1] W2[h2][h1] b2[h2] g2[h2] be2[h2] Wh[A+1][h2] bh[A+1]
// Stash (consumed by the backward): xhat1[B][h1] a1[B][h1] xhat2[B][h2] a2[B][h2] rstd1[B] rstd2[B].
#include "common.h"

int gemm_f32(uav_ctx* ctx, int64_t M, int64_t N, int64_t K, const float* A, int64_t sa_m, int64_t sa_k,
             const float* B, int64_t sb_k, int64_t sb_n, float* C, int64_t ldc, const float* bias,
             int accumulate, hipStream_t st);

constexpr float LN_EPS = 1e-5f;

struct MlpLayout {
    int64_t W1, b1, g1, be1, W2, b2, g2, be2, Wh, bh, total;
};
static MlpLayout mlp_layout(int in, int h1, int h2, int A) {
    MlpLayout L;
    int64_t o = 0;
    L.W1 = o; o += (int64_t)h1 * in;
    L.b1 = o; o += h1;
    L.g1 = o; o += h1;
    L.be1 = o; o += h1;
    L.W2 = o; o += (int64_t)h2 * h1;
    L.b2 = o; o += h2;
    L.g2 = o; o += h2;
    L.be2 = o; o += h2;
    L.Wh = o; o += (int64_t)(A + 1) * h2;
    L.bh = o; o += A + 1;
    L.total = o;
    return L;
}

// z (in place -> xhat), a = relu(xhat*g + b), rstd.  One wave per row, VPT values per lane.
template <int VPT>
__global__ __launch_bounds__(256) void ln_relu_fwd_kernel(float* __restrict__ z, float* __restrict__ a,
                                                          float* __restrict__ rstd_out,
                                                          const float* __restrict__ g,
                                                          const float* __restrict__ be, int64_t B) {
    constexpr int C = VPT * 64;
    const int lane = threadIdx.x & 63;
    float gg[VPT], bb[VPT];
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        gg[j] = g[j * 64 + lane];
        bb[j] = be[j * 64 + lane];
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < B; row += (int64_t)gridDim.x * 4) {
        float v[VPT], s = 0.f;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            v[j] = z[row * C + j * 64 + lane];
            s += v[j];
        }
        const float mean = wave_allsum(s) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            v[j] -= mean;
            q += v[j] * v[j];
        }
        const float rstd = 1.0f / sqrtf(wave_allsum(q) * (1.0f / C) + LN_EPS);
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const float xh = v[j] * rstd;
            z[row * C + j * 64 + lane] = xh;
            const float pre = xh * gg[j] + bb[j];
            a[row * C + j * 64 + lane] = pre < 0.f ? 0.f : pre;      // NaN propagates like torch.relu (model.py:47-49 relies on it)
        }
        if (lane == 0) rstd_out[row] = rstd;
    }
}

// d (in place: dL/da -> dL/dz), partial[block][2][C] = per-block sums of (dgamma, dbeta)
template <int VPT>
__global__ __launch_bounds__(256) void ln_relu_bwd_kernel(float* __restrict__ d, const float* __restrict__ xhat,
                                                          const float* __restrict__ rstd,
                                                          const float* __restrict__ g,
                                                          const float* __restrict__ be, int64_t B,
                                                          float* __restrict__ partial) {
    constexpr int C = VPT * 64;
    __shared__ float sm[4][2][C];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float gg[VPT], bb[VPT], dg[VPT], db[VPT];
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        gg[j] = g[j * 64 + lane];
        bb[j] = be[j * 64 + lane];
        dg[j] = 0.f;
        db[j] = 0.f;
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < B; row += (int64_t)gridDim.x * 4) {
        const float rs = rstd[row];
        float xh[VPT], dxh[VPT], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            xh[j] = xhat[row * C + j * 64 + lane];
            float dy = d[row * C + j * 64 + lane];
            if (!(xh[j] * gg[j] + bb[j] > 0.f)) dy = 0.f;     // ReLU mask (recomputed pre-activation)
            dg[j] += dy * xh[j];
            db[j] += dy;
            dxh[j] = dy * gg[j];
            s1 += dxh[j];
            s2 += dxh[j] * xh[j];
        }
        const float m1 = wave_allsum(s1) * (1.0f / C), m2 = wave_allsum(s2) * (1.0f / C);
#pragma unroll
        for (int j = 0; j < VPT; ++j) d[row * C + j * 64 + lane] = rs * (dxh[j] - m1 - xh[j] * m2);
    }
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        sm[w][0][j * 64 + lane] = dg[j];
        sm[w][1][j * 64 + lane] = db[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int which = i / C, c = i % C;
        partial[((int64_t)blockIdx.x * 2 + which) * C + c] =
            ((sm[0][which][c] + sm[1][which][c]) + sm[2][which][c]) + sm[3][which][c];
    }
}

// out[c] = sum_b partial[b][c]: 8 independent partial sums in a fixed association (deterministic; a serial
// 1024-deep chain cost 160 us per call)
__global__ __launch_bounds__(256) void rows_reduce_kernel(const float* __restrict__ partial, int nb, int C,
                                                          float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float p8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 8 <= nb; b += 8)
#pragma unroll
        for (int k = 0; k < 8; ++k) p8[k] += partial[(int64_t)(b + k) * C + c];
    for (; b < nb; ++b) p8[0] += partial[(int64_t)b * C + c];
    out[c] = ((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]));
}

// partial[block][C] = sum over the block's rows of X[row][c]   (C <= 1024)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, int64_t B, int C,
                                                             int64_t rows_per_block, float* __restrict__ partial) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < B) ? r0 + rows_per_block : B;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int64_t r = r0; r < r1; ++r) s += X[r * C + c];
        partial[(int64_t)blockIdx.x * C + c] = s;
    }
}

// the same for C % 4 == 0: a thread owns FOUR adjacent columns (dwordx4 loads) and keeps four rows in flight; fixed
// association (row r goes to accumulator r % 4), so still deterministic.  4.3 GB of [1 M][1024] gate gradients: 1.8 -> ~0.9 ms
// AMAX: also max |x| over the whole matrix as float bits (NaN sorts above inf), one atomicMax per wave: the block scale of
// the split-fp16 GEMMs that consume the same matrix next (gemm_h3.hip) comes out of the pass that reads it anyway.
template <bool AMAX>
__global__ __launch_bounds__(256) void colsum_partial4_kernel(const float* __restrict__ X, int64_t B, int C,
                                                              int64_t rows_per_block, float* __restrict__ partial,
                                                              unsigned* __restrict__ absmax_bits) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < B) ? r0 + rows_per_block : B;
    unsigned mb = 0;
    auto seen = [&](const float4& v) {
        if (AMAX) {
            const unsigned a = __builtin_bit_cast(unsigned, v.x) & 0x7fffffffu, b = __builtin_bit_cast(unsigned, v.y) & 0x7fffffffu;
            const unsigned c2 = __builtin_bit_cast(unsigned, v.z) & 0x7fffffffu, d = __builtin_bit_cast(unsigned, v.w) & 0x7fffffffu;
            const unsigned ab = a > b ? a : b, cd = c2 > d ? c2 : d, m4 = ab > cd ? ab : cd;
            mb = mb > m4 ? mb : m4;
        }
    };
    for (int c = 4 * threadIdx.x; c < C; c += 1024) {
        float4 a[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        int64_t r = r0;
        for (; r + 4 <= r1; r += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(X + (r + k) * C + c);
                a[k].x += v.x; a[k].y += v.y; a[k].z += v.z; a[k].w += v.w;
                seen(v);
            }
        }
        for (; r < r1; ++r) {
            const float4 v = *reinterpret_cast<const float4*>(X + r * C + c);
            a[0].x += v.x; a[0].y += v.y; a[0].z += v.z; a[0].w += v.w;
            seen(v);
        }
        float* pp = partial + (int64_t)blockIdx.x * C + c;
        pp[0] = (a[0].x + a[1].x) + (a[2].x + a[3].x);
        pp[1] = (a[0].y + a[1].y) + (a[2].y + a[3].y);
        pp[2] = (a[0].z + a[1].z) + (a[2].z + a[3].z);
        pp[3] = (a[0].w + a[1].w) + (a[2].w + a[3].w);
    }
    if (AMAX) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned t = (unsigned)__shfl_xor((int)mb, o, 64);
            mb = mb > t ? mb : t;
        }
        if ((threadIdx.x & 63) == 0 && mb) atomicMax(absmax_bits, mb);
    }
}

// colsum_partial4_kernel<true> that ALSO accumulates X^T x for a narrow second operand x [B][I], I <= 8 (exact f32 FMAs):
// the layer-1 input weight gradient dW_ih = dG^T x of a wide LSTM (4H x 8) rides on the pass that reads dG for the bias
// gradient instead of costing a 3 ms thin GEMM of its own.  partial [block][1 + 8][C]; x rows are wave-uniform (scalar loads).
__global__ __launch_bounds__(256) void colsum_xw_partial_kernel(const float* __restrict__ X, int64_t B, int C,
                                                                int64_t rows_per_block, const float* __restrict__ x, int I,
                                                                float* __restrict__ partial, unsigned* __restrict__ absmax_bits) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < B) ? r0 + rows_per_block : B;
    unsigned mb = 0;
    for (int c = 4 * threadIdx.x; c < C; c += 1024) {
        float4 a = {0.f, 0.f, 0.f, 0.f}, w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = float4{0.f, 0.f, 0.f, 0.f};
        auto row = [&](const float4& v, int64_t r) {
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            const unsigned b0 = __builtin_bit_cast(unsigned, v.x) & 0x7fffffffu, b1 = __builtin_bit_cast(unsigned, v.y) & 0x7fffffffu;
            const unsigned b2 = __builtin_bit_cast(unsigned, v.z) & 0x7fffffffu, b3 = __builtin_bit_cast(unsigned, v.w) & 0x7fffffffu;
            const unsigned m01 = b0 > b1 ? b0 : b1, m23 = b2 > b3 ? b2 : b3, m4 = m01 > m23 ? m01 : m23;
            mb = mb > m4 ? mb : m4;
            const float* xr = x + r * I;                 // uniform
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xj = j < I ? xr[j] : 0.f;
                w[j].x = fmaf(v.x, xj, w[j].x); w[j].y = fmaf(v.y, xj, w[j].y);
                w[j].z = fmaf(v.z, xj, w[j].z); w[j].w = fmaf(v.w, xj, w[j].w);
            }
        };
        int64_t r = r0;
        for (; r + 4 <= r1; r += 4) {
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4*>(X + (r + k) * C + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) row(v[k], r + k);
        }
        for (; r < r1; ++r) row(*reinterpret_cast<const float4*>(X + r * C + c), r);
        float* pp = partial + (int64_t)blockIdx.x * 9 * C + c;
        *reinterpret_cast<float4*>(pp) = a;
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<float4*>(pp + (int64_t)(1 + j) * C) = w[j];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)mb, o, 64);
        mb = mb > t ? mb : t;
    }
    if (absmax_bits && (threadIdx.x & 63) == 0 && mb) atomicMax(absmax_bits, mb);
}
// partial [nb][9][C] -> colsum [C] and xw [C][I] (= X^T x, row-major like dW_ih), fixed association
__global__ __launch_bounds__(256) void colsum_xw_reduce_kernel(const float* __restrict__ partial, int nb, int C, int I,
                                                               float* __restrict__ colsum_out, float* __restrict__ xw_out,
                                                               int xw_transposed) {
    const int i = blockIdx.x * 256 + threadIdx.x;         // over (1 + I) * C
    if (i >= (1 + I) * C) return;
    const int q = i / C, c = i % C;
    float p8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 8 <= nb; b += 8)
#pragma unroll
        for (int k = 0; k < 8; ++k) p8[k] += partial[((int64_t)(b + k) * 9 + q) * C + c];
    for (; b < nb; ++b) p8[0] += partial[((int64_t)b * 9 + q) * C + c];
    const float s = ((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]));
    if (q == 0) { if (colsum_out) colsum_out[c] = s; }
    else if (xw_transposed) xw_out[(int64_t)(q - 1) * C + c] = s;          // x^T X  [I][C]
    else xw_out[(int64_t)c * I + (q - 1)] = s;                             // X^T x  [C][I]
}
// colsum of X [B][C] (colsum_out may be NULL) + X^T x for x [B][I] (or x^T X with xw_transposed), I <= 8, C % 4 == 0,
// X 16-byte aligned; scratch >= 1024 * 9 * C floats
int colsum_xw(uav_ctx* ctx, const float* X, int64_t B, int C, const float* x, int I, float* colsum_out, float* xw_out,
              float* scratch, unsigned* absmax_bits, hipStream_t st, int xw_transposed = 0) {
    UAV_REQUIRE(I >= 1 && I <= 8 && C % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0, "colsum_xw: I <= 8, cols %% 4 == 0, aligned X");
    int nb = (int)((B + 255) / 256);
    if (nb > 1024) nb = 1024;
    const int64_t rpb = (B + nb - 1) / nb;
    nb = (int)((B + rpb - 1) / rpb);
    if (absmax_bits) UAV_CHECK_HIP(hipMemsetAsync(absmax_bits, 0, sizeof(unsigned), st));
    hipLaunchKernelGGL(colsum_xw_partial_kernel, dim3(nb), dim3(256), 0, st, X, B, C, rpb, x, I, scratch, absmax_bits);
    hipLaunchKernelGGL(colsum_xw_reduce_kernel, dim3(((1 + I) * C + 255) / 256), dim3(256), 0, st, scratch, nb, C, I, colsum_out, xw_out,
                       xw_transposed);
    UAV_LAUNCH_CHECK();
    return 0;
}

int colsum_absmax(uav_ctx* ctx, const float* X, int64_t B, int C, float* out, float* scratch, unsigned* absmax_bits,
                  hipStream_t st);
int colsum(uav_ctx* ctx, const float* X, int64_t B, int C, float* out, float* scratch, hipStream_t st) {
    return colsum_absmax(ctx, X, B, C, out, scratch, nullptr, st);
}
// absmax_bits (optional; C % 4 == 0 and X 16-byte aligned required with it): device word that receives the bits of max |X|
int colsum_absmax(uav_ctx* ctx, const float* X, int64_t B, int C, float* out, float* scratch, unsigned* absmax_bits,
                  hipStream_t st) {
    // column sums are tiny next to the GEMMs; keep them simple and deterministic
    int nb = (int)((B + 255) / 256);
    if (nb > 1024) nb = 1024;
    const int64_t rpb = (B + nb - 1) / nb;
    nb = (int)((B + rpb - 1) / rpb);
    const bool vec = C % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
    UAV_REQUIRE(!absmax_bits || vec, "colsum: the fused absolute maximum needs cols %% 4 == 0 and a 16-byte aligned matrix");
    if (absmax_bits) {
        UAV_CHECK_HIP(hipMemsetAsync(absmax_bits, 0, sizeof(unsigned), st));
        hipLaunchKernelGGL(colsum_partial4_kernel<true>, dim3(nb), dim3(256), 0, st, X, B, C, rpb, scratch, absmax_bits);
    } else if (vec)
        hipLaunchKernelGGL(colsum_partial4_kernel<false>, dim3(nb), dim3(256), 0, st, X, B, C, rpb, scratch, (unsigned*)nullptr);
    else
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb), dim3(256), 0, st, X, B, C, rpb, scratch);
    hipLaunchKernelGGL(rows_reduce_kernel, dim3((C + 255) / 256), dim3(256), 0, st, scratch, nb, C, out);
    UAV_LAUNCH_CHECK();
    return 0;
}

template <int VPT>
static int ln_fwd(float* z, float* a, float* rstd, const float* g, const float* be, int64_t B, hipStream_t st) {
    int nb = (int)((B + 3) / 4);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(ln_relu_fwd_kernel<VPT>, dim3(nb), dim3(256), 0, st, z, a, rstd, g, be, B);
    UAV_LAUNCH_CHECK();
    return 0;
}
static int ln_fwd_any(int C, float* z, float* a, float* rstd, const float* g, const float* be, int64_t B,
                      hipStream_t st) {
    switch (C) {
        case 64: return ln_fwd<1>(z, a, rstd, g, be, B, st);
        case 128: return ln_fwd<2>(z, a, rstd, g, be, B, st);
        case 256: return ln_fwd<4>(z, a, rstd, g, be, B, st);
        case 512: return ln_fwd<8>(z, a, rstd, g, be, B, st);
    }
    uav_set_error("mlp: LayerNorm width %d unsupported (64/128/256/512)", C);
    return 2;
}

constexpr int LNB_BLOCKS = 512;
template <int VPT>
static int ln_bwd(uav_ctx* ctx, float* d, const float* xhat, const float* rstd, const float* g, const float* be,
                  int64_t B, float* dg_out, float* db_out, float* scratch, hipStream_t st) {
    constexpr int C = VPT * 64;
    int nb = (int)((B + 3) / 4);
    if (nb > LNB_BLOCKS) nb = LNB_BLOCKS;
    hipLaunchKernelGGL(ln_relu_bwd_kernel<VPT>, dim3(nb), dim3(256), 0, st, d, xhat, rstd, g, be, B, scratch);
    // partial is [nb][2][C]: reduce as 2C columns, then split
    hipLaunchKernelGGL(rows_reduce_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, st, scratch, nb, 2 * C,
                       scratch + (int64_t)LNB_BLOCKS * 2 * C);
    UAV_CHECK_HIP(hipMemcpyAsync(dg_out, scratch + (int64_t)LNB_BLOCKS * 2 * C, C * sizeof(float),
                                 hipMemcpyDeviceToDevice, st));
    UAV_CHECK_HIP(hipMemcpyAsync(db_out, scratch + (int64_t)LNB_BLOCKS * 2 * C + C, C * sizeof(float),
                                 hipMemcpyDeviceToDevice, st));
    UAV_LAUNCH_CHECK();
    return 0;
}
static int ln_bwd_any(uav_ctx* ctx, int C, float* d, const float* xhat, const float* rstd, const float* g,
                      const float* be, int64_t B, float* dg, float* db, float* scratch, hipStream_t st) {
    switch (C) {
        case 64: return ln_bwd<1>(ctx, d, xhat, rstd, g, be, B, dg, db, scratch, st);
        case 128: return ln_bwd<2>(ctx, d, xhat, rstd, g, be, B, dg, db, scratch, st);
        case 256: return ln_bwd<4>(ctx, d, xhat, rstd, g, be, B, dg, db, scratch, st);
        case 512: return ln_bwd<8>(ctx, d, xhat, rstd, g, be, B, dg, db, scratch, st);
    }
    uav_set_error("mlp: LayerNorm width %d unsupported (64/128/256/512)", C);
    return 2;
}

extern "C" {

int uav_ln_relu(uav_ctx* ctx, float* z, float* a, float* rstd, const float* gamma, const float* beta, int64_t rows,
                int cols, uav_stream stream) {
    UAV_REQUIRE(ctx && z && a && rstd && gamma && beta && rows > 0, "uav_ln_relu: bad argument");
    return ln_fwd_any(cols, z, a, rstd, gamma, beta, rows, as_stream(stream));
}

int uav_ln_relu_bwd(uav_ctx* ctx, float* d, const float* xhat, const float* rstd, const float* gamma, const float* beta,
                    int64_t rows, int cols, float* dgamma, float* dbeta, uav_stream stream) {
    UAV_REQUIRE(ctx && d && xhat && rstd && gamma && beta && dgamma && dbeta && rows > 0, "uav_ln_relu_bwd: bad argument");
    UAV_REQUIRE(ctx->ws_bytes >= ((size_t)LNB_BLOCKS * 2 * cols + 2 * cols) * sizeof(float), "uav_ln_relu_bwd: workspace too small");
    return ln_bwd_any(ctx, cols, d, xhat, rstd, gamma, beta, rows, dgamma, dbeta, (float*)ctx->ws, as_stream(stream));
}

int uav_colsum(uav_ctx* ctx, const float* x, int64_t rows, int cols, float* out, uav_stream stream) {
    UAV_REQUIRE(ctx && x && out && rows > 0 && cols > 0 && cols <= 1024, "uav_colsum: bad argument");
    UAV_REQUIRE(ctx->ws_bytes >= (size_t)1024 * 1024 * sizeof(float), "uav_colsum: workspace too small");
    return colsum(ctx, x, rows, cols, out, (float*)ctx->ws, as_stream(stream));
}

int64_t uav_mlp_param_count(int in_dim, int h1, int h2, int n_act) { return mlp_layout(in_dim, h1, h2, n_act).total; }
int64_t uav_mlp_stash_floats(int h1, int h2) { return 2 * (int64_t)h1 + 2 * (int64_t)h2 + 2; }

int uav_mlp_fwd(uav_ctx* ctx, const float* params, const float* x, int64_t B, int in_dim, int h1, int h2,
                int n_act, float* heads, float* stash, uav_stream stream) {
    UAV_REQUIRE(ctx && params && x && heads && stash && B > 0, "uav_mlp_fwd: bad argument (stash is required)");
    const MlpLayout L = mlp_layout(in_dim, h1, h2, n_act);
    hipStream_t st = as_stream(stream);
    float* xhat1 = stash;
    float* a1 = xhat1 + B * h1;
    float* xhat2 = a1 + B * h1;
    float* a2 = xhat2 + B * h2;
    float* rstd1 = a2 + B * h2;
    float* rstd2 = rstd1 + B;
    int rc;
    // z1 = x W1^T + b1   (NT: op(B)[k][j] = W1[j][k])
    if ((rc = gemm_f32(ctx, B, h1, in_dim, x, in_dim, 1, params + L.W1, 1, in_dim, xhat1, h1, params + L.b1, 0, st))) return rc;
    if ((rc = ln_fwd_any(h1, xhat1, a1, rstd1, params + L.g1, params + L.be1, B, st))) return rc;
    if ((rc = gemm_f32(ctx, B, h2, h1, a1, h1, 1, params + L.W2, 1, h1, xhat2, h2, params + L.b2, 0, st))) return rc;
    if ((rc = ln_fwd_any(h2, xhat2, a2, rstd2, params + L.g2, params + L.be2, B, st))) return rc;
    return gemm_f32(ctx, B, n_act + 1, h2, a2, h2, 1, params + L.Wh, 1, h2, heads, n_act + 1, params + L.bh, 0, st);
}

int uav_mlp_bwd(uav_ctx* ctx, const float* params, const float* x, float* stash, const float* dheads, int64_t B,
                int in_dim, int h1, int h2, int n_act, float* grad, uav_stream stream) {
    UAV_REQUIRE(ctx && params && x && stash && dheads && grad && B > 0, "uav_mlp_bwd: bad argument");
    const MlpLayout L = mlp_layout(in_dim, h1, h2, n_act);
    hipStream_t st = as_stream(stream);
    float* xhat1 = stash;
    float* a1 = xhat1 + B * h1;
    float* xhat2 = a1 + B * h1;
    float* a2 = xhat2 + B * h2;
    float* rstd1 = a2 + B * h2;
    float* rstd2 = rstd1 + B;
    const int A1 = n_act + 1;
    // scratch for column reductions lives behind the split-K slab area of the workspace
    const size_t red_floats = (size_t)LNB_BLOCKS * 2 * 512 + 2 * 512 + 1024 * 512;
    UAV_REQUIRE(ctx->ws_bytes >= (red_floats + (1 << 18)) * sizeof(float), "uav_mlp_bwd: workspace too small");
    float* red = (float*)((char*)ctx->ws + ctx->ws_bytes) - red_floats;
    uav_ctx sub = *ctx;                       // GEMM slabs may use everything in front of `red`
    sub.ws_bytes = ctx->ws_bytes - red_floats * sizeof(float);
    int rc;
    // heads: dWh = dheads^T a2 ; dbh = colsum(dheads) ; da2 = dheads Wh   (da2 overwrites a2)
    if ((rc = gemm_f32(&sub, A1, h2, B, dheads, 1, A1, a2, h2, 1, grad + L.Wh, h2, nullptr, 0, st))) return rc;
    if ((rc = colsum(&sub, dheads, B, A1, grad + L.bh, red, st))) return rc;
    if ((rc = gemm_f32(&sub, B, h2, A1, dheads, A1, 1, params + L.Wh, h2, 1, a2, h2, nullptr, 0, st))) return rc;
    if ((rc = ln_bwd_any(&sub, h2, a2, xhat2, rstd2, params + L.g2, params + L.be2, B, grad + L.g2, grad + L.be2, red, st))) return rc;
    // layer 2: dW2 = dz2^T a1 ; db2 ; da1 = dz2 W2  (overwrites a1)
    if ((rc = gemm_f32(&sub, h2, h1, B, a2, 1, h2, a1, h1, 1, grad + L.W2, h1, nullptr, 0, st))) return rc;
    if ((rc = colsum(&sub, a2, B, h2, grad + L.b2, red, st))) return rc;
    if ((rc = gemm_f32(&sub, B, h1, h2, a2, h2, 1, params + L.W2, h1, 1, a1, h1, nullptr, 0, st))) return rc;
    if ((rc = ln_bwd_any(&sub, h1, a1, xhat1, rstd1, params + L.g1, params + L.be1, B, grad + L.g1, grad + L.be1, red, st))) return rc;
    // layer 1: dW1 = dz1^T x ; db1
    if ((rc = gemm_f32(&sub, h1, in_dim, B, a1, 1, h1, x, in_dim, 1, grad + L.W1, in_dim, nullptr, 0, st))) return rc;
    return colsum(&sub, a1, B, h1, grad + L.b1, red, st);
}

}  // extern "C"

// rows_dot_core.h -- a few output columns of a short inner product, one WAVE per row (rows_dot_kernel of gemm.hip and the
// step-wise rollout's tail kernel of rollout.hip share this arithmetic, so the policy heads a rollout step samples from are
// bit for bit the ones uav_gemm_f32 writes).
#pragma once
#include "common.h"

template <int NO, int KQ>     // NO output columns (registers), KQ float4 chunks per lane: K <= 256 KQ
struct RowsDot {
    float4 wv[NO][KQ];
    // W [n_out][ldw] row-major, k contiguous; rows >= n_out and k >= K read as zero
    __device__ __forceinline__ void load_w(const float* __restrict__ W, int64_t ldw, int n_out, int K, int lane) {
#pragma unroll
        for (int o = 0; o < NO; ++o)
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                const int k = 4 * (lane + 64 * q);
                wv[o][q] = (o < n_out && k < K) ? *reinterpret_cast<const float4*>(W + o * ldw + k) : float4{0.f, 0.f, 0.f, 0.f};
            }
    }
    // rows m0 .. m0 + 3 of A (their loads in flight together): out[r][o] = A[m0 + r][:] . W[o][:], the full sum in EVERY lane
    __device__ __forceinline__ void rows4(const float* __restrict__ A, int64_t lda, int64_t m0, int64_t M, int K, int lane,
                                          float (&out)[4][NO]) const {
        float4 av[4][KQ];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                const int k = 4 * (lane + 64 * q);
                av[r][q] = (m0 + r < M && k < K) ? *reinterpret_cast<const float4*>(A + (m0 + r) * lda + k) : float4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                float p = 0.f;
#pragma unroll
                for (int q = 0; q < KQ; ++q)
                    p += (av[r][q].x * wv[o][q].x + av[r][q].y * wv[o][q].y) + (av[r][q].z * wv[o][q].z + av[r][q].w * wv[o][q].w);
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) p += __shfl_xor(p, d, 64);
                out[r][o] = p;
            }
    }
};

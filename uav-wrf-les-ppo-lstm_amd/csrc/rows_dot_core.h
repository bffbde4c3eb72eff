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
    // rows m0 .. m0 + 3 of A (their loads in flight together): red[r] = this lane's share of row r's NO = 8 totals after a
    // reduce-scatter over the wave -- lane l ends up with the total of column home(l) = 4 (l >> 5 & 1) + 2 (l >> 4 & 1) +
    // (l >> 3 & 1), column o is read back with total(red[r], o).  Ten cross-lane moves per row instead of the 48 of eight
    // full butterflies (which made the long-M form shuffle-bound: 0.53 ms for 1 M rows x 1 KB, 2 TB/s), and the SAME sums:
    // every total is formed by the same pairings in the same order (l with l ^ 32, then ^ 16, ^ 8, ^ 4, ^ 2, ^ 1).
    __device__ __forceinline__ void rows4(const float* __restrict__ A, int64_t lda, int64_t m0, int64_t M, int K, int lane,
                                          float (&red)[4]) const {
        static_assert(NO == 8, "the reduce-scatter below is written for eight columns");
        float4 av[4][KQ];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                const int k = 4 * (lane + 64 * q);
                av[r][q] = (m0 + r < M && k < K) ? *reinterpret_cast<const float4*>(A + (m0 + r) * lda + k) : float4{0.f, 0.f, 0.f, 0.f};
            }
        const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float p[NO];
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                p[o] = 0.f;
#pragma unroll
                for (int q = 0; q < KQ; ++q)
                    p[o] += (av[r][q].x * wv[o][q].x + av[r][q].y * wv[o][q].y) + (av[r][q].z * wv[o][q].z + av[r][q].w * wv[o][q].w);
            }
            float w4[4], w2[2];
#pragma unroll
            for (int k = 0; k < 4; ++k) {           // ^ 32: the upper half-wave keeps columns 4..7, the lower 0..3
                const float mine = b5 ? p[k + 4] : p[k], send = b5 ? p[k] : p[k + 4];
                w4[k] = mine + __shfl_xor(send, 32, 64);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {           // ^ 16
                const float mine = b4 ? w4[k + 2] : w4[k], send = b4 ? w4[k] : w4[k + 2];
                w2[k] = mine + __shfl_xor(send, 16, 64);
            }
            float w1;
            {                                       // ^ 8
                const float mine = b3 ? w2[1] : w2[0], send = b3 ? w2[0] : w2[1];
                w1 = mine + __shfl_xor(send, 8, 64);
            }
#pragma unroll
            for (int d = 4; d > 0; d >>= 1) w1 += __shfl_xor(w1, d, 64);
            red[r] = w1;
        }
    }
    // column o of a row's totals, in every lane (o a compile-time constant after unrolling)
    static __device__ __forceinline__ float total(float red, int o) {
        return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, red), ((o & 4) ? 32 : 0) + ((o & 2) ? 16 : 0) + ((o & 1) ? 8 : 0)));
    }
};

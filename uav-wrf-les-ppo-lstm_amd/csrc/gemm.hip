// gemm.hip -- exact-f32 MFMA GEMM building block: C[M][N] (+)= op(A) * op(B) + bias.
//
// Used by the MLP policy (model.py:20-40 Linear layers, forward and backward), the policy
// heads and the LSTM weight gradients.  v_mfma_f32_32x32x2_f32 is bit-for-bit an f32 fmaf
// chain (no TF32 on gfx950), so results equal a CPU fp32 matmul up to summation order.
// 64x64x16 block tile, 4 waves (2x2), each wave one 32x32 accumulator tile; operands staged
// through LDS k-major so fragment reads are conflict-free ds_read_b32.  Generic strides make
// NN / NT / TN one kernel; split-K (grid.z) writes slabs to the context workspace and a second
// kernel reduces them in a fixed order (deterministic, no float atomics).
#include "common.h"
#include "rows_dot_core.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 64, BK = 16;
constexpr int LDT = BM + 4;   // padded row of the k-major LDS tiles

// Stage a [rows=64][k=16] operand tile into T[k][row].  (r, k) element = base[r*s_r + k*s_k].
__device__ __forceinline__ void stage_tile(float (*T)[LDT], const float* __restrict__ base, int64_t s_r,
                                           int64_t s_k, int64_t r0, int64_t k0, int64_t R, int64_t Kend) {
    const int t = threadIdx.x;
    if (s_k == 1) {
        // k contiguous: one float4 along k per thread (64 rows x 4 quads)
        const int r = t >> 2, kq = (t & 3) * 4;
        const int64_t gr = r0 + r, gk = k0 + kq;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (gr < R) {
            const float* p = base + gr * s_r + gk;
            if (gk + 3 < Kend && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
                const float4 q = *reinterpret_cast<const float4*>(p);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (gk + j < Kend) v[j] = p[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) T[kq + j][r] = v[j];
    } else {
        // row index contiguous (or fully generic): four consecutive rows at one k per thread
        const int k = t >> 4, rq = (t & 15) * 4;
        const int64_t gk = k0 + k, gr = r0 + rq;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (gk < Kend) {
            const float* p = base + gk * s_k + gr * s_r;
            if (s_r == 1 && gr + 3 < R && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
                const float4 q = *reinterpret_cast<const float4*>(p);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (gr + j < R) v[j] = p[j * s_r];
            }
        }
        *reinterpret_cast<float4*>(&T[k][rq]) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(int64_t M, int64_t N, int64_t K,
                                                       const float* __restrict__ A, int64_t sa_m, int64_t sa_k,
                                                       const float* __restrict__ B, int64_t sb_k, int64_t sb_n,
                                                       float* __restrict__ C, int64_t ldc,
                                                       const float* __restrict__ bias, int accumulate,
                                                       int64_t k_per_split, float* __restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) float As[BK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BK][LDT];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int64_t m0 = (int64_t)blockIdx.x * BM, n0 = (int64_t)blockIdx.y * BN;
    const int64_t kb = (int64_t)blockIdx.z * k_per_split;
    const int64_t ke = (kb + k_per_split < K) ? kb + k_per_split : K;
    f32x16 acc = {0};
    for (int64_t k0 = kb; k0 < ke; k0 += BK) {
        stage_tile(As, A, sa_m, sa_k, m0, k0, M, ke);
        stage_tile(Bs, B, sb_n, sb_k, n0, k0, N, ke);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a = As[kk + (lane >> 5)][wm * 32 + (lane & 31)];
            const float b = Bs[kk + (lane >> 5)][wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D map of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int64_t col = n0 + wn * 32 + (lane & 31);
    if (col >= N) return;
    const float bv = (bias && !slabs) ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= M) continue;
        if (slabs) {
            slabs[((int64_t)blockIdx.z * M + row) * N + col] = acc[r];
        } else {
            float* c = C + row * ldc + col;
            *c = (accumulate ? *c : 0.f) + acc[r] + bv;
        }
    }
}

// ---- 128x128x16 tile for the large shapes (MLP layers at N*T rows, per-step LSTM GEMMs of the generic path):
// each wave owns a 64x64 sub-tile (2x2 accumulators of 32x32), the next k-tile travels global -> registers
// while the current one is multiplied (4 MFMAs per 4 LDS fragment reads instead of 1 per 2).
constexpr int BM2 = 128, BN2 = 128, LDT2 = 128 + 4;

struct TileRegs { float v[2][4]; };

__device__ __forceinline__ void tile_load(TileRegs& R, const float* __restrict__ base, int64_t s_r, int64_t s_k,
                                          int64_t r0, int64_t k0, int64_t Rn, int64_t Kend) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float* v = R.v[i];
        v[0] = v[1] = v[2] = v[3] = 0.f;
        if (s_k == 1) {
            const int r = (t >> 2) + 64 * i, kq = (t & 3) * 4;
            const int64_t gr = r0 + r, gk = k0 + kq;
            if (gr < Rn) {
                const float* p = base + gr * s_r + gk;
                if (gk + 3 < Kend && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
                    const float4 q = *reinterpret_cast<const float4*>(p);
                    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (gk + j < Kend) v[j] = p[j];
                }
            }
        } else {
            const int k = (t >> 5) + 8 * i, rq = (t & 31) * 4;
            const int64_t gk = k0 + k, gr = r0 + rq;
            if (gk < Kend) {
                const float* p = base + gk * s_k + gr * s_r;
                if (s_r == 1 && gr + 3 < Rn && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
                    const float4 q = *reinterpret_cast<const float4*>(p);
                    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (gr + j < Rn) v[j] = p[j * s_r];
                }
            }
        }
    }
}
__device__ __forceinline__ void tile_store(float (*T)[LDT2], const TileRegs& R, int64_t s_k) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (s_k == 1) {
            const int r = (t >> 2) + 64 * i, kq = (t & 3) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) T[kq + j][r] = R.v[i][j];
        } else {
            const int k = (t >> 5) + 8 * i, rq = (t & 31) * 4;
            *reinterpret_cast<float4*>(&T[k][rq]) = make_float4(R.v[i][0], R.v[i][1], R.v[i][2], R.v[i][3]);
        }
    }
}

__global__ __launch_bounds__(256) void gemm_f32_128_kernel(int64_t M, int64_t N, int64_t K,
                                                           const float* __restrict__ A, int64_t sa_m, int64_t sa_k,
                                                           const float* __restrict__ B, int64_t sb_k, int64_t sb_n,
                                                           float* __restrict__ C, int64_t ldc,
                                                           const float* __restrict__ bias, int accumulate,
                                                           int64_t k_per_split, float* __restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) float As[BK][LDT2];
    __shared__ __attribute__((aligned(16))) float Bs[BK][LDT2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int64_t m0 = (int64_t)blockIdx.x * BM2, n0 = (int64_t)blockIdx.y * BN2;
    const int64_t kb = (int64_t)blockIdx.z * k_per_split;
    const int64_t ke = (kb + k_per_split < K) ? kb + k_per_split : K;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0};
    TileRegs ra, rb;
    tile_load(ra, A, sa_m, sa_k, m0, kb, M, ke);
    tile_load(rb, B, sb_n, sb_k, n0, kb, N, ke);
    for (int64_t k0 = kb; k0 < ke; k0 += BK) {
        tile_store(As, ra, sa_k);
        tile_store(Bs, rb, sb_k);
        __syncthreads();
        if (k0 + BK < ke) {                       // next tile flies under the MFMAs
            tile_load(ra, A, sa_m, sa_k, m0, k0 + BK, M, ke);
            tile_load(rb, B, sb_n, sb_k, n0, k0 + BK, N, ke);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int kr = kk + (lane >> 5), c = lane & 31;
            const float a0 = As[kr][wm * 64 + c], a1 = As[kr][wm * 64 + 32 + c];
            const float b0 = Bs[kr][wn * 64 + c], b1 = Bs[kr][wn * 64 + 32 + c];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int64_t col = n0 + wn * 64 + ni * 32 + (lane & 31);
            if (col >= N) continue;
            const float bv = (bias && !slabs) ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= M) continue;
                if (slabs) {
                    slabs[((int64_t)blockIdx.z * M + row) * N + col] = acc[mi][ni][r];
                } else {
                    float* cp = C + row * ldc + col;
                    *cp = (accumulate ? *cp : 0.f) + acc[mi][ni][r] + bv;
                }
            }
        }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int S, int64_t M,
                                                            int64_t N, float* __restrict__ C, int64_t ldc,
                                                            const float* __restrict__ bias, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * N) return;
    const int64_t row = i / N, col = i % N;
    float s = 0.f;
    for (int z = 0; z < S; ++z) s += slabs[(int64_t)z * M * N + i];
    float* c = C + row * ldc + col;
    *c = (accumulate ? *c : 0.f) + s + (bias ? bias[col] : 0.f);
}

// ---- a few output columns (policy heads: N = A + 1 <= 8) of a short inner product (K <= 512): C[m][j] = A[m][:] . W[j][:] + b[j].
// No MFMA tile pays here (a 64 x 64 tile would compute 58 unused columns and needs split-K + a reduce launch to fill the
// chip: 18.6 us per call at 4096 x 6 x 256); one WAVE per row instead: lane l holds 4 (or 8) consecutive k of the row and
// of every W row (registers), the eight column sums reduce-scattered over the wave (rows_dot_core.h).  A long M (the
// time-batched heads of uav_lstm_fwd) streams A from HBM: 0.30 ms per 1 M rows x 1 KB.
template <int NO, int KQ>     // KQ = float4 chunks per lane (K <= 256 KQ); arithmetic: rows_dot_core.h
__global__ __launch_bounds__(256) void rows_dot_kernel(int64_t M, int K, const float* __restrict__ A, int64_t lda,
                                                       const float* __restrict__ W, int64_t ldw, int n_out, float* __restrict__ C,
                                                       int64_t ldc, const float* __restrict__ bias) {
    const int lane = threadIdx.x & 63;
    RowsDot<NO, KQ> rd;
    rd.load_w(W, ldw, n_out, K, lane);
    const float bv = (bias && lane < n_out) ? bias[lane] : 0.f;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwave = (int64_t)gridDim.x * 4;
    // four rows per trip: their loads are in flight together (a long M is a stream from HBM, one row per trip would run
    // at the memory latency)
    for (int64_t m0 = wave * 4; m0 < M; m0 += nwave * 4) {
        float red[4];
        rd.rows4(A, lda, m0, M, K, lane, red);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float mine = 0.f;
#pragma unroll
            for (int o = 0; o < NO; ++o)
                if (lane == o) mine = RowsDot<NO, KQ>::total(red[r], o);
            if (m0 + r < M && lane < n_out) C[(m0 + r) * ldc + lane] = mine + bv;
        }
    }
}

int gemm_f32(uav_ctx* ctx, int64_t M, int64_t N, int64_t K, const float* A, int64_t sa_m, int64_t sa_k,
             const float* B, int64_t sb_k, int64_t sb_n, float* C, int64_t ldc, const float* bias,
             int accumulate, hipStream_t st) {
    // the few-columns form (see rows_dot_kernel): NT operands with contiguous k, 16-byte aligned rows
    if (ctx && A && B && C && N >= 1 && N <= 8 && K >= 4 && K <= 512 && K % 4 == 0 && sa_k == 1 && sb_k == 1 && !accumulate && M >= 1 &&
        sa_m % 4 == 0 && sb_n % 4 == 0 && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0) {
        int64_t nb = (M + 15) / 16;                                     // 4 waves x 4 rows per trip
        if (nb > 8 * ctx->num_cu) nb = 8 * ctx->num_cu;
        if (K <= 256)
            hipLaunchKernelGGL((rows_dot_kernel<8, 1>), dim3((unsigned)nb), dim3(256), 0, st, M, (int)K, A, sa_m, B, sb_n, (int)N, C, ldc, bias);
        else
            hipLaunchKernelGGL((rows_dot_kernel<8, 2>), dim3((unsigned)nb), dim3(256), 0, st, M, (int)K, A, sa_m, B, sb_n, (int)N, C, ldc, bias);
        UAV_LAUNCH_CHECK();
        return 0;
    }
    UAV_REQUIRE(ctx && A && B && C && M > 0 && N > 0 && K > 0, "uav_gemm_f32: bad argument");
    const bool big = (M >= 128 && N >= 128);          // 128x128 tile for the large shapes
    const int bm = big ? BM2 : BM, bn = big ? BN2 : BN;
    const int64_t tm = (M + bm - 1) / bm, tn = (N + bn - 1) / bn;
    UAV_REQUIRE(tm < (1ll << 31) && tn <= 65535, "uav_gemm_f32: grid too large");
    // split K when the tile grid cannot fill the chip and K is deep
    int64_t S = 1;
    const int64_t tiles = tm * tn;
    if (tiles < 2 * ctx->num_cu && K >= 16 * BK) {
        S = (2 * ctx->num_cu + tiles - 1) / tiles;
        const int64_t maxS_k = K / (8 * BK);
        if (S > maxS_k) S = maxS_k;
        const int64_t maxS_ws = (int64_t)(ctx->ws_bytes / sizeof(float)) / (M * N);
        if (S > maxS_ws) S = maxS_ws;
        if (S < 1) S = 1;
    }
    int64_t kps = ((K + S - 1) / S + BK - 1) / BK * BK;
    S = (K + kps - 1) / kps;
    float* slabs = (S > 1) ? (float*)ctx->ws : nullptr;
    if (big)
        hipLaunchKernelGGL(gemm_f32_128_kernel, dim3((unsigned)tm, (unsigned)tn, (unsigned)S), dim3(256), 0, st, M, N, K, A,
                           sa_m, sa_k, B, sb_k, sb_n, C, ldc, bias, accumulate, kps, slabs);
    else
        hipLaunchKernelGGL(gemm_f32_kernel, dim3((unsigned)tm, (unsigned)tn, (unsigned)S), dim3(256), 0, st, M, N, K, A,
                           sa_m, sa_k, B, sb_k, sb_n, C, ldc, bias, accumulate, kps, slabs);
    if (S > 1) {
        const int64_t nb = (M * N + 255) / 256;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, slabs, (int)S, M, N, C, ldc,
                           bias, accumulate);
    }
    UAV_LAUNCH_CHECK();
    return 0;
}

extern "C" int uav_gemm_f32(uav_ctx* ctx, int64_t M, int64_t N, int64_t K, const float* A, int64_t sa_m,
                            int64_t sa_k, const float* B, int64_t sb_k, int64_t sb_n, float* C, int64_t ldc,
                            const float* bias, int accumulate, uav_stream stream) {
    return gemm_f32(ctx, M, N, K, A, sa_m, sa_k, B, sb_k, sb_n, C, ldc, bias, accumulate, as_stream(stream));
}

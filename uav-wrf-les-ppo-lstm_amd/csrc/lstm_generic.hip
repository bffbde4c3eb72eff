// lstm_generic.hip -- nn.LSTM forward/backward for hidden sizes without a VGPR-resident persistent kernel (BASELINE
// C5's h=256: the two fp16 pieces of W_hh are 1 MB, a CU's register file is 512 KB): the time-major decomposition with
// ONE launch per time step.
//   H % 64 == 0 (h=256):  step_fwd_h3_kernel  = recurrent product on the fp16 matrix pipe (two-piece operand split, three
//                         products, f32 accuracy: common.h split2h) with the LSTM cell fused into its epilogue; weights
//                         pre-split once per call, h_t handed from step to step as fp16 piece planes (ping-pong), every
//                         workgroup a 64 units x 64 envs tile, operands straight from L2 into registers, no LDS, no barrier;
//                         step_bwd_h3_kernel  = dh_{t-1} = dG_t W_hh the same way, dG block-scaled per env by a power of
//                         two (cell_bwd_h3_kernel, which also writes the f32 dG rows uav_lstm_wgrad consumes).
//                         The launch boundary IS the cross-CU exchange of h_t (1.5-2 us): cheaper than any in-kernel
//                         flag hand-off of 16 KB per CU and step (MI355X_MICROARCH.md price list: 4+ us).
//   any other H:          gemm.hip's exact-f32 GEMM + a pointwise kernel per step (UAV_LSTM_F32_MFMA=1 forces this path).
// Same stash layout and semantics as lstm.hip (gates i,f,g,o | c_prev | h_prev per (n,t)).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

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


// ================================================================================================ fp16-split step kernels
// w [rows][cols] f32 -> pieces [2][rows][cols] fp16 (transpose = false) or [2][cols][rows] (transpose = true)
__global__ void split_weights_kernel(const float* __restrict__ w, int rows, int cols, int transpose,
                                     unsigned short* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i % cols);
    _Float16 p0, p1;
    split2h(w[i], p0, p1);
    const int64_t o = transpose ? (int64_t)c * rows + r : i;
    out[o] = h_bits(p0);
    out[(int64_t)rows * cols + o] = h_bits(p1);
}

// state <- init * keep[:,0], as f32 (hs, cs) and as fp16 piece planes of h (hp [2][N][H])
__global__ void h3_init_state(const float* __restrict__ h0, const float* __restrict__ c0, const float* __restrict__ keep,
                              int N, int T, int H, float* __restrict__ hs, float* __restrict__ cs,
                              unsigned short* __restrict__ hp) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)N * H) return;
    const int n = (int)(i / H);
    const float k = keep ? keep[(size_t)n * T] : 1.f;
    const float h = h0[i] * k;
    hs[i] = h;
    cs[i] = c0[i] * k;
    _Float16 p0, p1;
    split2h(h, p0, p1);
    hp[i] = h_bits(p0);
    hp[(int64_t)N * H + i] = h_bits(p1);
}

__device__ __forceinline__ f16x8 ldh8(const unsigned short* p) { return *reinterpret_cast<const f16x8*>(p); }

// One time step of one layer: gates = pre (x W_ih^T + b, already in the stash) + W_hh h_{t-1}, cell, outputs.
// Workgroup = 4 waves = a tile of 64 units x 64 envs; wave w owns units 16 w .. 16 w + 15 of the tile (its four gate row
// tiles) for the four 16-env column tiles: 16 main + 16 cross accumulators.  K = H in slabs of 32, next slab's fragments in
// flight while the current one multiplies (1 wave per SIMD: up to 512 VGPRs).  A = weight pieces [2][4H][H], B = h pieces
// [2][N][H]: lane (r16, kq) reads 8 consecutive k of row / env r16 -- one dwordx4 each.
template <int H>
__global__ __launch_bounds__(256) void step_fwd_h3_kernel(const unsigned short* __restrict__ wp,
                                                          const unsigned short* __restrict__ hp_in,
                                                          unsigned short* __restrict__ hp_out, float* __restrict__ hs,
                                                          float* __restrict__ cs, float* __restrict__ stash,
                                                          const float* __restrict__ keep, int N, int T, int t,
                                                          float* __restrict__ y, float* __restrict__ hn,
                                                          float* __restrict__ cn) {
    constexpr int NS = H / 32;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int e0 = blockIdx.x * 64, u0 = blockIdx.y * 64 + 16 * w;
    const size_t plane_w = (size_t)4 * H * H, plane_h = (size_t)N * H;
    const unsigned short* ap[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) ap[g] = wp + (size_t)(g * H + u0 + r16) * H + 8 * kq;
    const unsigned short* bp[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) bp[c] = hp_in + (size_t)min(e0 + 16 * c + r16, N - 1) * H + 8 * kq;

    // accumulators start from the input projection: acc[g][c][r] <-> unit u0 + 4 kq + r, env e0 + 16 c + r16
    f32x4 acc[4][4], acl[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int n = min(e0 + 16 * c + r16, N - 1);
        const float* sp = stash + ((size_t)n * T + t) * (6 * H) + u0 + 4 * kq;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(sp + g * H);
            acc[g][c] = f32x4{v.x, v.y, v.z, v.w};
            acl[g][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    f16x8 a[2][4][2], b[2][4][2];                     // [buffer][gate | col tile][piece]
#pragma unroll
    for (int g = 0; g < 4; ++g) { a[0][g][0] = ldh8(ap[g]); a[0][g][1] = ldh8(ap[g] + plane_w); }
#pragma unroll
    for (int c = 0; c < 4; ++c) { b[0][c][0] = ldh8(bp[c]); b[0][c][1] = ldh8(bp[c] + plane_h); }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        if (s + 1 < NS) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                a[nxt][g][0] = ldh8(ap[g] + 32 * (s + 1));
                a[nxt][g][1] = ldh8(ap[g] + plane_w + 32 * (s + 1));
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                b[nxt][c][0] = ldh8(bp[c] + 32 * (s + 1));
                b[nxt][c][1] = ldh8(bp[c] + plane_h + 32 * (s + 1));
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acl[g][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][g][1], b[cur][c][0], acl[g][c], 0, 0, 0);
                acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][g][0], b[cur][c][0], acc[g][c], 0, 0, 0);
                acl[g][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][g][0], b[cur][c][1], acl[g][c], 0, 0, 0);
            }
    }
    // ---- cell (gen_cell_fwd's arithmetic) and outputs
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int n = e0 + 16 * c + r16;
        if (n >= N) continue;
        const size_t row = (size_t)n * T + t, i0 = (size_t)n * H + u0 + 4 * kq;
        float* sp = stash + row * (6 * H) + u0 + 4 * kq;
        const float4 cp4 = *reinterpret_cast<const float4*>(cs + i0), hp4 = *reinterpret_cast<const float4*>(hs + i0);
        const float cp[4] = {cp4.x, cp4.y, cp4.z, cp4.w};
        float gi[4], gf[4], gg[4], go[4], cc[4], hh[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gi[r] = fast_sigmoid(acc[0][c][r] + acl[0][c][r] * H3_LO);
            gf[r] = fast_sigmoid(acc[1][c][r] + acl[1][c][r] * H3_LO);
            gg[r] = fast_tanh(acc[2][c][r] + acl[2][c][r] * H3_LO);
            go[r] = fast_sigmoid(acc[3][c][r] + acl[3][c][r] * H3_LO);
            cc[r] = gf[r] * cp[r] + gi[r] * gg[r];
            hh[r] = go[r] * fast_tanh(cc[r]);
        }
        *reinterpret_cast<float4*>(sp) = float4{gi[0], gi[1], gi[2], gi[3]};
        *reinterpret_cast<float4*>(sp + H) = float4{gf[0], gf[1], gf[2], gf[3]};
        *reinterpret_cast<float4*>(sp + 2 * H) = float4{gg[0], gg[1], gg[2], gg[3]};
        *reinterpret_cast<float4*>(sp + 3 * H) = float4{go[0], go[1], go[2], go[3]};
        *reinterpret_cast<float4*>(sp + 4 * H) = cp4;
        *reinterpret_cast<float4*>(sp + 5 * H) = hp4;
        *reinterpret_cast<float4*>(y + row * H + u0 + 4 * kq) = float4{hh[0], hh[1], hh[2], hh[3]};
        if (t == T - 1) {
            *reinterpret_cast<float4*>(hn + i0) = float4{hh[0], hh[1], hh[2], hh[3]};
            *reinterpret_cast<float4*>(cn + i0) = float4{cc[0], cc[1], cc[2], cc[3]};
        } else {
            const float kn = keep ? keep[row + 1] : 1.f;
            unsigned short q0[4], q1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                hh[r] *= kn;
                cc[r] *= kn;
                _Float16 p0, p1;
                split2h(hh[r], p0, p1);
                q0[r] = h_bits(p0);
                q1[r] = h_bits(p1);
            }
            *reinterpret_cast<float4*>(hs + i0) = float4{hh[0], hh[1], hh[2], hh[3]};
            *reinterpret_cast<float4*>(cs + i0) = float4{cc[0], cc[1], cc[2], cc[3]};
            uint2 v0, v1;
            v0.x = (unsigned)q0[0] | ((unsigned)q0[1] << 16); v0.y = (unsigned)q0[2] | ((unsigned)q0[3] << 16);
            v1.x = (unsigned)q1[0] | ((unsigned)q1[1] << 16); v1.y = (unsigned)q1[2] | ((unsigned)q1[3] << 16);
            *reinterpret_cast<uint2*>(hp_out + i0) = v0;
            *reinterpret_cast<uint2*>(hp_out + plane_h + i0) = v1;
        }
    }
}

// Gate gradients of step t (gen_cell_bwd's arithmetic), one block per env: dgates f32 [n][t][4H] for the weight-gradient
// pass AND, for the recurrent product, the same row as two fp16 planes scaled by the power of two that puts the row's
// largest magnitude in [2^13, 2^14) (gradients span dozens of binades; scaled back exactly by step_bwd_h3_kernel).
template <int H>
__global__ __launch_bounds__(H) void cell_bwd_h3_kernel(const float* __restrict__ stash, const float* __restrict__ keep,
                                                        const float* __restrict__ dy, int N, int T, int t,
                                                        const float* __restrict__ dh_rec, float* __restrict__ dc_next,
                                                        float* __restrict__ dgates, unsigned short* __restrict__ dgp,
                                                        float* __restrict__ inv_scale) {
    __shared__ float smax[H / 64];
    const int n = blockIdx.x, u = threadIdx.x;
    const size_t row = (size_t)n * T + t, i = (size_t)n * H + u;
    const float* sp = stash + row * (6 * H);
    const float gi = sp[u], gf = sp[H + u], gg = sp[2 * H + u], go = sp[3 * H + u], cp = sp[4 * H + u];
    const float dh = dy[row * H + u] + dh_rec[i];
    const float c = gf * cp + gi * gg;
    const float tch = fast_tanh(c);
    const float dc = dh * go * (1.0f - tch * tch) + dc_next[i];
    float g4[4];
    g4[0] = dc * gg * gi * (1.0f - gi);
    g4[1] = dc * cp * gf * (1.0f - gf);
    g4[2] = dc * gi * (1.0f - gg * gg);
    g4[3] = dh * tch * go * (1.0f - go);
    float* gp = dgates + row * (4 * H);
    float m = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        gp[q * H + u] = g4[q];
        m = fmaxf(m, fabsf(g4[q]));
    }
    const float kp = keep ? keep[row] : 1.f;
    dc_next[i] = dc * gf * kp;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((u & 63) == 0) smax[u >> 6] = m;
    __syncthreads();
    m = smax[0];
#pragma unroll
    for (int k = 1; k < H / 64; ++k) m = fmaxf(m, smax[k]);
    // power of two 2^e with m * 2^e in [2^13, 2^14); exponent clamped so that both the scale and its inverse are normal
    int e = 0;
    if (m > 0.f && m < 3.0e38f) {
        e = 13 - (int)((__float_as_uint(m) >> 23) & 0xff) + 127;
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
    }
    const float sc = __uint_as_float((unsigned)(127 + e) << 23), isc = __uint_as_float((unsigned)(127 - e) << 23);
    if (u == 0) inv_scale[n] = isc * kp;                    // the mask of step t rides on the scale
    const size_t plane = (size_t)N * 4 * H;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        _Float16 p0, p1;
        split2h(g4[q] * sc, p0, p1);
        dgp[(size_t)n * 4 * H + q * H + u] = h_bits(p0);
        dgp[plane + (size_t)n * 4 * H + q * H + u] = h_bits(p1);
    }
}

// dh_{t-1}[n][u] = keep[n][t] * sum_k dG_t[n][k] W_hh[k][u]: A = W_hh^T pieces [2][H][4H] (rows = units), B = the scaled dG
// pieces [2][N][4H]; tile 64 units x 64 envs, wave w: one 16-unit row tile x four env column tiles; K = 4H.
template <int H>
__global__ __launch_bounds__(256) void step_bwd_h3_kernel(const unsigned short* __restrict__ wtp,
                                                          const unsigned short* __restrict__ dgp,
                                                          const float* __restrict__ inv_scale, int N,
                                                          float* __restrict__ dh) {
    constexpr int K = 4 * H, NS = K / 32, DEPTH = 4;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int e0 = blockIdx.x * 64, u0 = blockIdx.y * 64 + 16 * w;
    const size_t plane_w = (size_t)H * K, plane_g = (size_t)N * K;
    const unsigned short* ap = wtp + (size_t)(u0 + r16) * K + 8 * kq;
    const unsigned short* bp[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) bp[c] = dgp + (size_t)min(e0 + 16 * c + r16, N - 1) * K + 8 * kq;
    f32x4 acc[4], acl[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = acl[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a[DEPTH][2], b[DEPTH][4][2];
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) {
        a[d][0] = ldh8(ap + 32 * d); a[d][1] = ldh8(ap + plane_w + 32 * d);
#pragma unroll
        for (int c = 0; c < 4; ++c) { b[d][c][0] = ldh8(bp[c] + 32 * d); b[d][c][1] = ldh8(bp[c] + plane_g + 32 * d); }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int cur = s % DEPTH, nxt = (s + DEPTH - 1) % DEPTH;
        if (s + DEPTH - 1 < NS) {
            const int sn = s + DEPTH - 1;
            a[nxt][0] = ldh8(ap + 32 * sn); a[nxt][1] = ldh8(ap + plane_w + 32 * sn);
#pragma unroll
            for (int c = 0; c < 4; ++c) { b[nxt][c][0] = ldh8(bp[c] + 32 * sn); b[nxt][c][1] = ldh8(bp[c] + plane_g + 32 * sn); }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acl[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][1], b[cur][c][0], acl[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][0], b[cur][c][0], acc[c], 0, 0, 0);
            acl[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][0], b[cur][c][1], acl[c], 0, 0, 0);
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int n = e0 + 16 * c + r16;
        if (n >= N) continue;
        const float is = inv_scale[n];
        const f32x4 v = (acc[c] + acl[c] * H3_LO) * is;
        *reinterpret_cast<float4*>(dh + (size_t)n * H + u0 + 4 * kq) = float4{v[0], v[1], v[2], v[3]};
    }
}

static bool h3_step_ok(int H) { return H == 256 && !uav_want_f32_mfma() && !getenv("UAV_LSTM_STEP_F32"); }

static int lstm_h3_fwd(uav_ctx* ctx, const float* keep, const float* h0, const float* c0, const float* w_hh, int N, int T,
                       float* y, float* hn, float* cn, float* stash, hipStream_t st) {
    constexpr int H = 256;
    const int64_t NH = (int64_t)N * H;
    // tail of the workspace: hs, cs f32 | two ping-pong sets of h pieces | W_hh pieces
    const size_t need = (size_t)(2 * NH) * 4 + (size_t)(2 * 2 * NH) * 2 + (size_t)2 * 4 * H * H * 2;
    UAV_REQUIRE(need + (64u << 20) <= ctx->ws_bytes, "lstm (h=256): workspace too small");
    char* base = (char*)ctx->ws + ctx->ws_bytes - need;
    float* hs = (float*)base;
    float* cs = hs + NH;
    unsigned short* hp0 = (unsigned short*)(cs + NH);
    unsigned short* hp1 = hp0 + 2 * NH;
    unsigned short* wp = hp1 + 2 * NH;
    const unsigned nb = (unsigned)((NH + 255) / 256);
    hipLaunchKernelGGL(split_weights_kernel, dim3(4 * H * H / 256), dim3(256), 0, st, w_hh, 4 * H, H, 0, wp);
    hipLaunchKernelGGL(h3_init_state, dim3(nb), dim3(256), 0, st, h0, c0, keep, N, T, H, hs, cs, hp0);
    const dim3 grid((N + 63) / 64, H / 64);
    for (int t = 0; t < T; ++t)
        hipLaunchKernelGGL((step_fwd_h3_kernel<H>), grid, dim3(256), 0, st, wp, (t & 1) ? hp1 : hp0, (t & 1) ? hp0 : hp1, hs, cs,
                           stash, keep, N, T, t, y, hn, cn);
    UAV_LAUNCH_CHECK();
    return 0;
}

static int lstm_h3_bwd(uav_ctx* ctx, const float* keep, const float* stash, const float* w_hh, const float* dy,
                       const float* dhn, const float* dcn, int N, int T, float* dgates, float* dh0, float* dc0,
                       hipStream_t st) {
    constexpr int H = 256;
    const int64_t NH = (int64_t)N * H;
    // tail of the workspace: dh, dc f32 | dG pieces [2][N][4H] | inv_scale [N] | W_hh^T pieces [2][H][4H]
    const size_t need = (size_t)(2 * NH) * 4 + (size_t)(2 * 4 * NH) * 2 + (size_t)((N + 63) / 64 * 64) * 4 + (size_t)2 * 4 * H * H * 2;
    UAV_REQUIRE(need + (64u << 20) <= ctx->ws_bytes, "lstm (h=256): workspace too small");
    char* base = (char*)ctx->ws + ctx->ws_bytes - need;
    float* dh = (float*)base;
    float* dc = dh + NH;
    unsigned short* dgp = (unsigned short*)(dc + NH);
    float* inv_scale = (float*)(dgp + 2 * 4 * NH);
    unsigned short* wtp = (unsigned short*)(inv_scale + (N + 63) / 64 * 64);
    const unsigned nb = (unsigned)((NH + 255) / 256);
    hipLaunchKernelGGL(split_weights_kernel, dim3(4 * H * H / 256), dim3(256), 0, st, w_hh, 4 * H, H, 1, wtp);
    hipLaunchKernelGGL(gen_fill, dim3(nb), dim3(256), 0, st, dh, dhn, NH);
    hipLaunchKernelGGL(gen_fill, dim3(nb), dim3(256), 0, st, dc, dcn, NH);
    const dim3 grid((N + 63) / 64, H / 64);
    for (int t = T - 1; t >= 0; --t) {
        hipLaunchKernelGGL((cell_bwd_h3_kernel<H>), dim3(N), dim3(H), 0, st, stash, keep, dy, N, T, t, dh, dc, dgates, dgp, inv_scale);
        hipLaunchKernelGGL((step_bwd_h3_kernel<H>), grid, dim3(256), 0, st, wtp, dgp, inv_scale, N, dh);
    }
    if (dh0) hipLaunchKernelGGL(gen_fill, dim3(nb), dim3(256), 0, st, dh0, dh, NH);
    if (dc0) hipLaunchKernelGGL(gen_fill, dim3(nb), dim3(256), 0, st, dc0, dc, NH);
    UAV_LAUNCH_CHECK();
    return 0;
}

// pre-activations of all steps must already be in the gates slot of the stash (x W_ih^T + b)
int lstm_generic_fwd(uav_ctx* ctx, const float* keep, const float* h0, const float* c0, const float* w_hh, int N, int T,
                     int H, float* y, float* hn, float* cn, float* stash, hipStream_t st) {
    if (h3_step_ok(H)) return lstm_h3_fwd(ctx, keep, h0, c0, w_hh, N, T, y, hn, cn, stash, st);
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
    if (h3_step_ok(H)) return lstm_h3_bwd(ctx, keep, stash, w_hh, dy, dhn, dcn, N, T, dgates, dh0, dc0, st);
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

// lstm_generic.hip -- nn.LSTM forward/backward for hidden sizes without a VGPR-resident persistent kernel (BASELINE
// C5's h=256: the two fp16 pieces of W_hh are 1 MB, a CU's register file is 512 KB): the time-major decomposition with
// ONE launch per time step.
//   H % 64 == 0 (h=256):  step_fwd_h3_kernel  = recurrent product on the fp16 matrix pipe (two-piece operand split, three
//                         products, f32 accuracy: common.h split2h) with the LSTM cell fused into its epilogue; weights
//                         pre-split once per call, h_t handed from step to step as fp16 piece planes (ping-pong), every
//                         workgroup a 64 units x 64 envs tile, operands straight from L2 into registers, no LDS, no barrier;
//                         step_bwd_h3_kernel  = dh_{t-1} = dG_t W_hh the same way, dG block-scaled per env by a power of
//                         two (cell_bwd_h3_kernel); since round 5 those pieces are KEPT for all steps and are what uav_lstm_wgrad
//                         consumes (common.h: DgPack, wgrad_pc.hip) -- no f32 dG rows, no h_prev slot in the stash.
//                         The launch boundary IS the cross-CU exchange of h_t (1.5-2 us): cheaper than any in-kernel
//                         flag hand-off of 16 KB per CU and step (MI355X_MICROARCH.md price list: 4+ us).
//   any other H:          gemm.hip's exact-f32 GEMM + a pointwise kernel per step (UAV_LSTM_F32_MFMA=1 forces this path).
// Same stash layout and semantics as lstm.hip (gates i,f,g,o | c_prev | h_prev per (n,t)).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef CELL_ABL
#define CELL_ABL 0      // ablation builds of cell_bwd_h3_kernel only (tools/ab_step_fwd.sh): 1 no piece stores, 2 no f32 dG stores, 3 neither
#endif
#ifndef STEP_ABL
#define STEP_ABL 0      // ablation builds of the step kernels only (tools/ab_step_fwd.sh): step_fwd_h3_kernel 1 no MFMAs, 2 no recurrent loads,
#endif                  // 3 no epilogue, 4 each wave loads ONE of the four B column tiles (what sharing B through LDS would leave on the L1
                        // path); 5 step_bwd_h3_kernel: each wave loads one of its two A and one of its two B tiles (the same question)

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
// MFMA-fragment order of an operand matrix M [R][K] (R rows = A rows or B columns, K a multiple of 32): the 8 consecutive
// k that lane (r16, kq) of a 16x16x32 MFMA feeds from row 16 rt + r16 in slab s sit at
//   frag[((rt * (K/32) + s) * 2 + piece) * 512 + (kq * 16 + r16) * 8 + i],   k = 32 s + 8 kq + i
// so one wave-wide dwordx4 load is 1 KB of consecutive bytes (row-major pieces give 16 half-used 128-B lines per load).
__host__ __device__ inline size_t frag_index(int row, int k, int K, int piece) {
    return ((size_t)((row >> 4) * (K >> 5) + (k >> 5)) * 2 + piece) * 512 + (size_t)((((k & 31) >> 3) * 16 + (row & 15)) * 8 + (k & 7));
}

// w [rows][cols] f32 (cols zero-padded to KP) -> fragment-ordered fp16 pieces of w (transpose = 0: A rows = w rows, K = KP)
// or of w^T (transpose = 1: A rows = w columns, K = rows)
__global__ void split_weights_kernel(const float* __restrict__ w, int rows, int cols, int KP, int transpose,
                                     unsigned short* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)rows * KP) return;
    const int r = (int)(i / KP), c = (int)(i % KP);
    _Float16 p0, p1;
    split2h(c < cols ? w[(int64_t)r * cols + c] : 0.f, p0, p1);
    if (!transpose) {
        out[frag_index(r, c, KP, 0)] = h_bits(p0);
        out[frag_index(r, c, KP, 1)] = h_bits(p1);
    } else {
        out[frag_index(c, r, rows, 0)] = h_bits(p0);
        out[frag_index(c, r, rows, 1)] = h_bits(p1);
    }
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
    const int u = (int)(i % H);
    hp[frag_index(n, u, H, 0)] = h_bits(p0);
    hp[frag_index(n, u, H, 1)] = h_bits(p1);
}

__device__ __forceinline__ f16x8 ldh8(const unsigned short* p) { return *reinterpret_cast<const f16x8*>(p); }

// One time step of one layer: gates = (b_ih + b_hh) + W_ih x_t + W_hh h_{t-1}, cell, outputs -- input projection included,
// so no [N][T][4H] pre-activation array is ever written or read.
// Workgroup = 8 waves = a tile of 64 units x 64 envs; wave w owns units 16 (w & 3) .. + 15 of the tile (its four gate row
// tiles) for two of the four 16-env column tiles (w >> 2): 8 main + 8 cross accumulators, two waves per SIMD.
// K runs over the input (IP = I rounded up to 32, x_t read as f32 and split in registers) and then over H (h_{t-1} as fp16
// piece planes) in slabs of 32, next slab's fragments in flight while the current one multiplies.
// A = weight pieces [2][4H][IP] and [2][4H][H], lane (r16, kq) reads 8 consecutive k of row r16 -- one dwordx4.
template <int H, int IPS, int RING>
__device__ __forceinline__ void step_fwd_h3_body(const unsigned short* __restrict__ wxp,
                                                 const unsigned short* __restrict__ wp,
                                                 const float* __restrict__ bsum, const float* __restrict__ x, int I,
                                                 const unsigned short* __restrict__ xp, int last_pieces,
                                                 const unsigned short* __restrict__ hp_in,
                                                 unsigned short* __restrict__ hp_out, float* __restrict__ hs,
                                                 float* __restrict__ cs, float* __restrict__ stash,
                                                 const float* __restrict__ keep, int64_t keep_sn, int64_t keep_off,
                                                 int N, int T, int t,
                                                 float* __restrict__ y, float* __restrict__ hn,
                                                 float* __restrict__ cn, int hslot) {
    constexpr int NS = H / 32, NC = 4;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: scalar bases
    const unsigned lo = lane * 8;                      // this lane's 16 bytes of a 1 KB fragment chunk
    const int r16 = lane & 15, kq = lane >> 4;
    const int e0 = blockIdx.x * 64, u0 = blockIdx.y * 64 + 16 * w;
    int nrow[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) nrow[c] = min(e0 + 16 * c + r16, N - 1);
    // restart mask of THIS step (keep[n][t] in {0, 1}: 0 where an episode ended in step t - 1): applied to the incoming
    // state here -- the recurrent B fragments of a masked env are ANDed to zero (exact), c_prev / h_prev are multiplied in
    // the epilogue -- so the outgoing state is written unmasked and no step needs the NEXT step's mask (which a rollout
    // only learns from the environment step in between)
    float kin[NC];
    unsigned kmask[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        kin[c] = keep ? keep[(size_t)nrow[c] * keep_sn + keep_off] : 1.f;
        kmask[c] = kin[c] != 0.f ? 0xffffffffu : 0u;
    }
    auto masked = [&](const f16x8& v, unsigned m) {
        uint4 q = __builtin_bit_cast(uint4, v);
        q.x &= m; q.y &= m; q.z &= m; q.w &= m;
        return __builtin_bit_cast(f16x8, q);
    };

    // accumulators start from the bias: acc[g][c][r] <-> unit u0 + 4 kq + r, env e0 + 16 c + r16
    f32x4 acc[4][NC], acl[4][NC];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(bsum + g * H + u0 + 4 * kq);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            acc[g][c] = f32x4{v.x, v.y, v.z, v.w};
            acl[g][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    auto mac = [&](const f16x8 (&a)[4][2], const f16x8 (&b)[NC][2]) {
#if STEP_ABL == 1
        asm volatile("" ::"v"(a[0][0]), "v"(a[1][0]), "v"(a[2][0]), "v"(a[3][0]), "v"(a[0][1]), "v"(a[1][1]), "v"(a[2][1]), "v"(a[3][1]),
                     "v"(b[0][0]), "v"(b[0][1]), "v"(b[1][0]), "v"(b[1][1]));
        return;
#endif
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                acl[g][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[g][1], b[c][0], acl[g][c], 0, 0, 0);
                acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[g][0], b[c][0], acc[g][c], 0, 0, 0);
                acl[g][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[g][0], b[c][1], acl[g][c], 0, 0, 0);
            }
    };
    if (xp) {
        // ---- input given as piece planes (xp: x_t already split, fragment order with K = 32 IPS -- the layer below's state
        // planes in a step-wise rollout, or a chunk converted by split_x_kernel in the sequence driver): ONE ring over the
        // NS recurrent and IPS input slabs, coalesced 1 KB chunk loads, no split arithmetic, no f32 row gathers
        const unsigned short* ap[4];
        const unsigned short* axp[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            ap[g] = wp + (size_t)((g * H + u0) >> 4) * NS * 1024;
            axp[g] = wxp + (size_t)((g * H + u0) >> 4) * IPS * 1024;
        }
        const unsigned short* bp[NC];
        const unsigned short* bxp[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            bp[c] = hp_in + (size_t)((e0 + 16 * c) >> 4) * NS * 1024;
            bxp[c] = xp + (size_t)((e0 + 16 * c) >> 4) * IPS * 1024;
        }
        constexpr int DEPTH = RING, NSL = NS + IPS;
        f16x8 a[DEPTH][4][2], b[DEPTH][NC][2];
        auto fetch = [&](int s, int buf) {
            const bool rec = s < NS;
            const int q = rec ? s : s - NS;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const unsigned short* p = rec ? ap[g] : axp[g];
                a[buf][g][0] = ldh8(p + (1024 * q + lo));
                a[buf][g][1] = ldh8(p + (1024 * q + 512 + lo));
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
#if STEP_ABL == 4
                if (c != w) continue;
#endif
                const unsigned short* p = rec ? bp[c] : bxp[c];
                b[buf][c][0] = ldh8(p + (1024 * q + lo));
                b[buf][c][1] = ldh8(p + (1024 * q + 512 + lo));
            }
        };
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) fetch(d, d);
#pragma unroll
        for (int s = 0; s < NSL; ++s) {
            if (s + DEPTH - 1 < NSL) fetch(s + DEPTH - 1, (s + DEPTH - 1) % DEPTH);
            __builtin_amdgcn_sched_barrier(0);
            if (s < NS) {
#pragma unroll
                for (int c = 0; c < NC; ++c) { b[s % DEPTH][c][0] = masked(b[s % DEPTH][c][0], kmask[c]); b[s % DEPTH][c][1] = masked(b[s % DEPTH][c][1], kmask[c]); }
            }
            mac(a[s % DEPTH], b[s % DEPTH]);
        }
    } else {
    // ---- recurrent product, K = H, double-buffered slabs
    {
        const unsigned short* ap[4];                   // fragment order (frag_index): 1 KB per wave-wide load
#pragma unroll
        for (int g = 0; g < 4; ++g) ap[g] = wp + (size_t)((g * H + u0) >> 4) * NS * 1024;
        const unsigned short* bp[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) bp[c] = hp_in + (size_t)((e0 + 16 * c) >> 4) * NS * 1024;
        // DEPTH - 1 slabs in flight.  The scheduler barrier pins every slab's loads where they are written: left alone
        // the machine scheduler sinks them next to their uses (fewest registers), one slab in flight, and the kernel
        // then runs at the L2 latency per slab (step_bwd_h3_kernel: 20.5 -> 14.4 us by the same change)
        // (only with a short input projection behind it: with IPS = 8 -- layer 2, whose time goes to the x gathers and
        // the stash stores -- the pinned ring measured 36.6 us per step against 34.2 for the plain double buffer)
        constexpr int DEPTH = IPS <= 2 ? RING : 2;
        f16x8 a[DEPTH][4][2], b[DEPTH][NC][2];         // [buffer][gate | col tile][piece]
        auto fetch = [&](int s, int buf) {
#pragma unroll
            for (int g = 0; g < 4; ++g) { a[buf][g][0] = ldh8(ap[g] + (1024 * s + lo)); a[buf][g][1] = ldh8(ap[g] + (1024 * s + 512 + lo)); }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
#if STEP_ABL == 4
                if (c != w) continue;
#endif
                b[buf][c][0] = ldh8(bp[c] + (1024 * s + lo)); b[buf][c][1] = ldh8(bp[c] + (1024 * s + 512 + lo));
            }
        };
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) fetch(d, d);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#if STEP_ABL == 2
            mac(a[0], b[0]);
#else
            if (s + DEPTH - 1 < NS) fetch(s + DEPTH - 1, (s + DEPTH - 1) % DEPTH);
            if (DEPTH > 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < NC; ++c) { b[s % DEPTH][c][0] = masked(b[s % DEPTH][c][0], kmask[c]); b[s % DEPTH][c][1] = masked(b[s % DEPTH][c][1], kmask[c]); }
            mac(a[s % DEPTH], b[s % DEPTH]);
#endif
        }
    }
    // ---- input projection: x_t [n][I] f32 -> two fp16 pieces per value, K = 32 IPS; the next slab's weight fragments and
    // raw x values are in flight while the current slab is split and multiplied
    {
        constexpr int IP = 32 * IPS;
        const unsigned short* axp[4];                  // fragment order: + s * 1024 per slab, + 512 for the second piece
#pragma unroll
        for (int g = 0; g < 4; ++g) axp[g] = wxp + (size_t)((g * H + u0) >> 4) * IPS * 1024;
        const float* xr[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) xr[c] = x + ((size_t)nrow[c] * T + t) * I + 8 * kq;
        constexpr int DEPTH = IPS >= 2 ? 2 : 1;        // deeper / pinned was slower here (layer 2: 40.4 against 34.2 us): the
        f16x8 a[DEPTH][4][2];                          // split arithmetic wants to slide between the neighbouring slabs' MFMAs
        float4 xv[DEPTH][NC][2];
        auto fetch = [&](int s, int buf) {
#pragma unroll
            for (int g = 0; g < 4; ++g) { a[buf][g][0] = ldh8(axp[g] + (1024 * s + lo)); a[buf][g][1] = ldh8(axp[g] + (1024 * s + 512 + lo)); }
            const int k0 = 32 * s + 8 * kq;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (k0 + 8 <= I) {
                    xv[buf][c][0] = *reinterpret_cast<const float4*>(xr[c] + 32 * s);
                    xv[buf][c][1] = *reinterpret_cast<const float4*>(xr[c] + 32 * s + 4);
                } else {
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = (k0 + i < I) ? xr[c][32 * s + i] : 0.f;
                    xv[buf][c][0] = float4{v[0], v[1], v[2], v[3]};
                    xv[buf][c][1] = float4{v[4], v[5], v[6], v[7]};
                }
            }
        };
#pragma unroll
        for (int d = 0; d < DEPTH - 1 || d == 0; ++d) fetch(d, d);
#pragma unroll
        for (int s = 0; s < IPS; ++s) {
            const int cur = s % DEPTH;
            if (DEPTH > 1 && s + DEPTH - 1 < IPS) fetch(s + DEPTH - 1, (s + DEPTH - 1) % DEPTH);
            f16x8 b[NC][2];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float v[8] = {xv[cur][c][0].x, xv[cur][c][0].y, xv[cur][c][0].z, xv[cur][c][0].w,
                                    xv[cur][c][1].x, xv[cur][c][1].y, xv[cur][c][1].z, xv[cur][c][1].w};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    _Float16 p0, p1;
                    split2h(v[i], p0, p1);
                    b[c][0][i] = p0;
                    b[c][1][i] = p1;
                }
            }
            mac(a[cur], b);
        }
        (void)IP;
    }
    }
    // ---- cell (gen_cell_fwd's arithmetic) and outputs
#if STEP_ABL == 3
    if (acc[0][0][0] == 123.456f)
#endif
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int n = e0 + 16 * c + r16;
        if (n >= N) continue;
        const size_t row = (size_t)n * T + t, i0 = (size_t)n * H + u0 + 4 * kq;
        float* sp = stash + row * (6 * H) + u0 + 4 * kq;
        // the state entering step t: from (hs, cs) at t = 0, afterwards from this row's own c_prev | h_prev slots, where
        // step t - 1 left it (the state travels through the stash rows the BPTT needs anyway: no separate f32 state
        // array is written and read back on every step)
        // (hslot = 0: the h_prev slot of the stash is neither read nor written -- the weight gradients then take h_prev from y,
        //  DgPack's iscm; 2 of the 10 KB a step moves per env and layer)
        float4 cp4 = *reinterpret_cast<const float4*>(t == 0 ? cs + i0 : sp + 4 * H);
        float4 hp4 = float4{0.f, 0.f, 0.f, 0.f};
        if (hslot) hp4 = *reinterpret_cast<const float4*>(t == 0 ? hs + i0 : sp + 5 * H);
        const float kc = kin[c];
        cp4.x *= kc; cp4.y *= kc; cp4.z *= kc; cp4.w *= kc;
        hp4.x *= kc; hp4.y *= kc; hp4.z *= kc; hp4.w *= kc;
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
        if (t == 0 || kc == 0.f) {                  // the row holds the MASKED state (what the BPTT and dW_hh read): step t - 1
            *reinterpret_cast<float4*>(sp + 4 * H) = cp4;   // left it unmasked, so a restarted env's slots are rewritten (rare)
            if (hslot) *reinterpret_cast<float4*>(sp + 5 * H) = hp4;
        }
        *reinterpret_cast<float4*>(y + row * H + u0 + 4 * kq) = float4{hh[0], hh[1], hh[2], hh[3]};
        if (t == T - 1) {
            *reinterpret_cast<float4*>(hn + i0) = float4{hh[0], hh[1], hh[2], hh[3]};
            *reinterpret_cast<float4*>(cn + i0) = float4{cc[0], cc[1], cc[2], cc[3]};
        }
        if (t < T - 1 || last_pieces) {             // last_pieces: the layer above reads this step's h from the piece planes
            unsigned short q0[4], q1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                _Float16 p0, p1;
                split2h(hh[r], p0, p1);
                q0[r] = h_bits(p0);
                q1[r] = h_bits(p1);
            }
            if (t < T - 1) {                           // state (unmasked: step t + 1 applies its own mask) -> the next row's c_prev | h_prev slots
                *reinterpret_cast<float4*>(sp + 6 * H + 4 * H) = float4{cc[0], cc[1], cc[2], cc[3]};
                if (hslot) *reinterpret_cast<float4*>(sp + 6 * H + 5 * H) = float4{hh[0], hh[1], hh[2], hh[3]};
            }
            uint2 v0, v1;
            v0.x = (unsigned)q0[0] | ((unsigned)q0[1] << 16); v0.y = (unsigned)q0[2] | ((unsigned)q0[3] << 16);
            v1.x = (unsigned)q1[0] | ((unsigned)q1[1] << 16); v1.y = (unsigned)q1[2] | ((unsigned)q1[3] << 16);
            const int u = u0 + 4 * kq;                 // four consecutive k of the next step's B fragment
            *reinterpret_cast<uint2*>(hp_out + frag_index(n, u, H, 0)) = v0;
            *reinterpret_cast<uint2*>(hp_out + frag_index(n, u, H, 1)) = v1;
        }
    }
}

template <int H, int IPS>
__global__ __launch_bounds__(256) void step_fwd_h3_kernel(const unsigned short* __restrict__ wxp,
                                                          const unsigned short* __restrict__ wp,
                                                          const float* __restrict__ bsum, const float* __restrict__ x, int I,
                                                          const unsigned short* __restrict__ xp, int last_pieces,
                                                          const unsigned short* __restrict__ hp_in,
                                                          unsigned short* __restrict__ hp_out, float* __restrict__ hs,
                                                          float* __restrict__ cs, float* __restrict__ stash,
                                                          const float* __restrict__ keep, int64_t keep_sn, int64_t keep_off,
                                                          int N, int T, int t,
                                                          float* __restrict__ y, float* __restrict__ hn,
                                                          float* __restrict__ cn, int hslot) {
    step_fwd_h3_body<H, IPS, 3>(wxp, wp, bsum, x, I, xp, last_pieces, hp_in, hp_out, hs, cs, stash, keep, keep_sn, keep_off, N, T, t, y, hn, cn,
                                hslot);
}

// TWO independent steps as one launch (blockIdx.z picks): the update's forward pass runs layer 1's step t + 1 beside layer 2's
// step t, so that two workgroups share every CU -- a step launch costs ~13-16 us of latency whatever its size
// (profiles/r05_step_share_ablation.log) and two of them side by side take 1.75-1.8 x one, not 2 x.
struct StepFwdArgs {
    const unsigned short *wxp, *wp;
    const float *bsum, *x;
    int I;
    const unsigned short* xp;
    int last_pieces;
    const unsigned short* hp_in;
    unsigned short* hp_out;
    float *hs, *cs, *stash;
    const float* keep;
    int64_t keep_sn, keep_off;
    int N, T, t;
    float *y, *hn, *cn;
    int hslot;
};
// (the workgroups do not share a SIMD -- a wave takes 418 registers for its accumulators and three-slab ring, and held to 256 it
//  spills 100 of them -- they follow one another on the CUs without a launch in between)
template <int H, int IPS_A, int IPS_B>
__global__ __launch_bounds__(256) void step_fwd_h3_pair_kernel(const StepFwdArgs a, const StepFwdArgs b) {
    if (blockIdx.z == 0)
        step_fwd_h3_body<H, IPS_A, 3>(a.wxp, a.wp, a.bsum, a.x, a.I, a.xp, a.last_pieces, a.hp_in, a.hp_out, a.hs, a.cs, a.stash, a.keep, a.keep_sn,
                                   a.keep_off, a.N, a.T, a.t, a.y, a.hn, a.cn, a.hslot);
    else
        step_fwd_h3_body<H, IPS_B, 3>(b.wxp, b.wp, b.bsum, b.x, b.I, b.xp, b.last_pieces, b.hp_in, b.hp_out, b.hs, b.cs, b.stash, b.keep, b.keep_sn,
                                   b.keep_off, b.N, b.T, b.t, b.y, b.hn, b.cn, b.hslot);
}

__global__ void add2v_kernel(const float* a0, const float* a1, float* b, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) b[i] = a0[i] + a1[i];
}

// Gate gradients of step t (gen_cell_bwd's arithmetic).  One WAVE per env (H = 256: lane l owns units 4 l .. 4 l + 3):
// the row as two fp16 planes in MFMA fragment order, scaled by the power of two that puts the row's largest magnitude in
// [2^13, 2^14) (gradients span dozens of binades; scaled back exactly by step_bwd_h3_kernel) -- for the recurrent product AND,
// kept for all steps (common.h: DgPack), for the weight-gradient pass.  dgates != NULL (UAV_DEBUG_DG_F32, the round-4 form kept
// as an A/B switch): the pieces are one step's scratch and the weight-gradient pass gets f32 rows [n][t][4H] as well.
// A workgroup = the 16 envs of one fragment row tile (16 waves): every wave parks its row's pieces in LDS and the workgroup
// then writes the tile's 64 KB of piece chunks as ONE linear block -- lanes writing their own 8 bytes straight to the
// fragment order scattered 16-byte segments over 32 lines per store, and the 16.8 MB of pieces cost 7.5 us of the step's
// 20 (profiles/r02_cell_bwd_ablation.log).  LDS image: chunk (slab, piece) at CB_CH halves, kq rows at CB_KQ: strides chosen
// so that the 32 (slab, kq) positions one store instruction touches fall on different banks (2-way at worst).
constexpr int CB_KQ = 136, CB_CH = 592;                 // halves: 272 B per kq row (16 envs x 16 B + 16), 1184 B per chunk
#ifndef BWD_DX_DEPTH
#define BWD_DX_DEPTH 5                                  // prefetch ring of step_bwd_h3_kernel<DX>: 5 slabs = 364 registers, so that two
#endif                                                  // cell_bwd waves (2 x 64) of the other layer's stream fit the SIMD's 512 beside it (6: 412)
#ifndef CELL_EPW
#define CELL_EPW 8                                      // envs per workgroup of cell_bwd_h3_kernel (16 = one workgroup per row tile)
#endif
constexpr size_t CELL_BWD_LDS = (size_t)64 * CB_CH * 2;
template <int H, int EPW>
__global__ __launch_bounds__(EPW * 64) void cell_bwd_h3_kernel(const float* __restrict__ stash, const float* __restrict__ keep,
                                                          const float* __restrict__ dy, const float* __restrict__ dheads,
                                                          const float* __restrict__ w_head, int NH, int N, int T, int t,
                                                          const float* __restrict__ dh_rec, float* __restrict__ dc_next,
                                                          float* __restrict__ dgates, unsigned short* __restrict__ dgp,
                                                          float* __restrict__ inv_scale, float* __restrict__ isc_out,
                                                          float* __restrict__ iscm_out) {
    static_assert(H == 256, "one wave per env: 64 lanes x 4 units");
    static_assert(EPW == 16 || EPW == 8, "a workgroup = a whole 16-env fragment row tile, or half of one");
    extern __shared__ __attribute__((aligned(16))) unsigned short cb_lds[];
    // EPW = 8: two workgroups of 8 waves per row tile (each writes its envs' 128-byte half of every 256-byte kq row of the tile's
    // block): two waves per SIMD at 64 registers leave room for a step_bwd_h3_kernel wave of the OTHER layer's stream beside them
    constexpr int TPB = 16 / EPW;
    const int lane = threadIdx.x & 63, r16 = (blockIdx.x % TPB) * EPW + (threadIdx.x >> 6);
    const int tile = blockIdx.x / TPB;
    const int n = tile * 16 + r16, u = 4 * lane;
    const bool live = n < N;
    // this lane's 8 bytes of chunk (slab 8 q + lane / 8, piece): kq = (lane % 8) / 2, halves 4 (lane & 1) ..
    unsigned short* lp = cb_lds + (size_t)(2 * (lane >> 3)) * CB_CH + ((lane & 7) >> 1) * CB_KQ + r16 * 8 + 4 * (lane & 1);
    if (live) {
    const size_t row = (size_t)n * T + t, i = (size_t)n * H + u;
    const float* sp = stash + row * (6 * H) + u;
    const float4 gi4 = *reinterpret_cast<const float4*>(sp), gf4 = *reinterpret_cast<const float4*>(sp + H);
    const float4 gg4 = *reinterpret_cast<const float4*>(sp + 2 * H), go4 = *reinterpret_cast<const float4*>(sp + 3 * H);
    const float4 cp4 = *reinterpret_cast<const float4*>(sp + 4 * H);
    float4 dy4;
    if (dy) dy4 = *reinterpret_cast<const float4*>(dy + row * H + u);
    else {           // top layer: dy = dheads . W_head formed here (the row's few dheads are wave-uniform), never written to HBM
        dy4 = float4{0.f, 0.f, 0.f, 0.f};
        const float* dhd = dheads + row * NH;
        for (int a = 0; a < NH; ++a) {
            const float d = dhd[a];
            const float4 wv = *reinterpret_cast<const float4*>(w_head + (size_t)a * H + u);
            dy4.x = fmaf(d, wv.x, dy4.x); dy4.y = fmaf(d, wv.y, dy4.y); dy4.z = fmaf(d, wv.z, dy4.z); dy4.w = fmaf(d, wv.w, dy4.w);
        }
    }
    const float4 dr4 = *reinterpret_cast<const float4*>(dh_rec + i);
    const float4 dn4 = *reinterpret_cast<const float4*>(dc_next + i);
    const float gi[4] = {gi4.x, gi4.y, gi4.z, gi4.w}, gf[4] = {gf4.x, gf4.y, gf4.z, gf4.w};
    const float gg[4] = {gg4.x, gg4.y, gg4.z, gg4.w}, go[4] = {go4.x, go4.y, go4.z, go4.w};
    const float cp[4] = {cp4.x, cp4.y, cp4.z, cp4.w};
    const float dyv[4] = {dy4.x, dy4.y, dy4.z, dy4.w}, drv[4] = {dr4.x, dr4.y, dr4.z, dr4.w}, dnv[4] = {dn4.x, dn4.y, dn4.z, dn4.w};
    const float kp = keep ? keep[row] : 1.f;
    float g4[4][4], dcn[4], m = 0.f;                  // g4[gate][r]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float dh = dyv[r] + drv[r];
        const float c = gf[r] * cp[r] + gi[r] * gg[r];
        const float tch = fast_tanh(c);
        const float dc = dh * go[r] * (1.0f - tch * tch) + dnv[r];
        g4[0][r] = dc * gg[r] * gi[r] * (1.0f - gi[r]);
        g4[1][r] = dc * cp[r] * gf[r] * (1.0f - gf[r]);
        g4[2][r] = dc * gi[r] * (1.0f - gg[r] * gg[r]);
        g4[3][r] = dh * tch * go[r] * (1.0f - go[r]);
        dcn[r] = dc * gf[r] * kp;
#pragma unroll
        for (int q = 0; q < 4; ++q) m = fmaxf(m, fabsf(g4[q][r]));
    }
    if (dgates) {
        float* gp = dgates + row * (4 * H) + u;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(gp + q * H) = float4{g4[q][0], g4[q][1], g4[q][2], g4[q][3]};
    }
    *reinterpret_cast<float4*>(dc_next + i) = float4{dcn[0], dcn[1], dcn[2], dcn[3]};
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    // power of two 2^e with m * 2^e in [2^13, 2^14); exponent clamped so that both the scale and its inverse are normal
    int e = 0;
    if (m > 0.f && m < 3.0e38f) {
        e = 13 - (int)((__float_as_uint(m) >> 23) & 0xff) + 127;
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
    }
    const float sc = __uint_as_float((unsigned)(127 + e) << 23), isc = __uint_as_float((unsigned)(127 - e) << 23);
    if (lane == 0) {
        inv_scale[n] = isc * kp;                            // dh_{t-1}: the mask of step t rides on the scale
        isc_out[n] = isc;                                   // dx_t (the gradient of the step's input), db and dW_ih are not masked
        if (iscm_out) iscm_out[n] = isc * kp;               // dW_hh = dG^T (y[t-1] keep[t])
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned short q0[4], q1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            _Float16 p0, p1;
            split2h(g4[q][r] * sc, p0, p1);
            q0[r] = h_bits(p0);
            q1[r] = h_bits(p1);
        }
        uint2 v0, v1;
        v0.x = (unsigned)q0[0] | ((unsigned)q0[1] << 16); v0.y = (unsigned)q0[2] | ((unsigned)q0[3] << 16);
        v1.x = (unsigned)q1[0] | ((unsigned)q1[1] << 16); v1.y = (unsigned)q1[2] | ((unsigned)q1[3] << 16);
        // gate q = slabs 8 q .. 8 q + 7 of the [N][4H] operand
        *reinterpret_cast<uint2*>(lp + (size_t)(16 * q) * CB_CH) = v0;
        *reinterpret_cast<uint2*>(lp + (size_t)(16 * q + 1) * CB_CH) = v1;
    }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {                       // a ragged last tile: rows past N stay zero
            *reinterpret_cast<uint2*>(lp + (size_t)(16 * q) * CB_CH) = uint2{0u, 0u};
            *reinterpret_cast<uint2*>(lp + (size_t)(16 * q + 1) * CB_CH) = uint2{0u, 0u};
        }
        if (lane == 0) {                                    // (the grid covers whole 64-row tiles: n < NP)
            isc_out[n] = 0.f;
            if (iscm_out) iscm_out[n] = 0.f;
        }
    }
    __syncthreads();
#if CELL_ABL != 1 && CELL_ABL != 3
    // the tile's piece block is 64 chunks x 1 KB of consecutive memory (frag_index): a linear copy, 16 bytes per thread
    // (EPW = 8: this workgroup's eight envs = one 128-byte line of each kq row)
    unsigned short* gout = dgp + (size_t)tile * 64 * 512;
    constexpr int LG = EPW == 16 ? 4 : 3;                  // log2(envs per workgroup)
    const int e_first = (blockIdx.x % TPB) * EPW;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int e8 = ps * (EPW * 64) + threadIdx.x, c = e8 >> (LG + 2), kq = (e8 >> LG) & 3, r = e_first + (e8 & (EPW - 1));
        const uint4 v = *reinterpret_cast<const uint4*>(cb_lds + (size_t)c * CB_CH + kq * CB_KQ + r * 8);
        *reinterpret_cast<uint4*>(gout + ((size_t)c * 64 + kq * 16 + r) * 8) = v;
    }
#endif
}

// dh_{t-1}[n][u] = keep[n][t] * sum_k dG_t[n][k] W_hh[k][u]: A = W_hh^T pieces [2][H][4H] (rows = units), B = the scaled dG
// pieces [2][N][4H]; tile 64 units x 64 envs, wave w: one 16-unit row tile x four env column tiles; K = 4H.
// DX: the same B fragments also multiply W_ih^T (wxtp, I = H): dx_t = dG_t W_ih, the gradient of the layer's input, written at
// [n][t] of dx -- instead of a separate [N T x 4H] x [4H x I] GEMM that re-reads all of dG (3.5 ms per epoch at C5).
template <int H, bool DX>
__global__ __launch_bounds__(256) void step_bwd_h3_kernel(const unsigned short* __restrict__ wtp,
                                                          const unsigned short* __restrict__ dgp,
                                                          const float* __restrict__ inv_scale,
                                                          const float* __restrict__ isc_t, int N,
                                                          float* __restrict__ dh, const unsigned short* __restrict__ wxtp,
                                                          float* __restrict__ dx, int T, int t) {
    // four waves, each TWO 16-unit row tiles x TWO 16-env column tiles of the 64 x 64 workgroup tile: 8 KB of fragments per
    // slab feed 12 MFMAs (one row tile x two column tiles per wave in eight waves loaded 6 KB for 6: the kernel is bound by
    // the L1 path, 48 KB per slab and workgroup then, 32 KB now); one wave per SIMD, so the ring can be deep
    constexpr int K = 4 * H, NS = K / 32, DEPTH = DX ? BWD_DX_DEPTH : 8, NR = 2, NC = 2;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const unsigned lo = lane * 8;
    const int e0 = blockIdx.x * 64 + 32 * (w >> 1), u0 = blockIdx.y * 64 + 32 * (w & 1);
    const unsigned short* ap[NR];
    const unsigned short* axp[NR];
    const unsigned short* bp[NC];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        ap[r] = wtp + (size_t)((u0 + 16 * r) >> 4) * NS * 1024;                              // fragment order (frag_index)
        axp[r] = DX ? wxtp + (size_t)((u0 + 16 * r) >> 4) * NS * 1024 : nullptr;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) bp[c] = dgp + (size_t)((e0 + 16 * c) >> 4) * NS * 1024;
    f32x4 acc[NR][NC], acl[NR][NC], xcc[NR][NC], xcl[NR][NC];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[r][c] = acl[r][c] = xcc[r][c] = xcl[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a[DEPTH][NR][2], ax[DX ? DEPTH : 1][NR][2], b[DEPTH][NC][2];
    auto fetch = [&](int s, int buf) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
#if STEP_ABL == 5
            if (r != (w >> 1)) continue;
#endif
            a[buf][r][0] = ldh8(ap[r] + (1024 * s + lo)); a[buf][r][1] = ldh8(ap[r] + (1024 * s + 512 + lo));
        }
        if (DX) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
#if STEP_ABL == 5
                if (r != (w >> 1)) continue;
#endif
                ax[buf][r][0] = ldh8(axp[r] + (1024 * s + lo)); ax[buf][r][1] = ldh8(axp[r] + (1024 * s + 512 + lo));
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#if STEP_ABL == 5
            if (c != (w & 1)) continue;
#endif
            b[buf][c][0] = ldh8(bp[c] + (1024 * s + lo)); b[buf][c][1] = ldh8(bp[c] + (1024 * s + 512 + lo));
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) fetch(d, d);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int cur = s % DEPTH;
        if (s + DEPTH - 1 < NS) fetch(s + DEPTH - 1, (s + DEPTH - 1) % DEPTH);
        // pin the loads HERE: left alone the scheduler sinks them next to their uses (56 VGPRs, one slab in flight) and the
        // kernel runs at the L2 latency per slab
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                acl[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][r][1], b[cur][c][0], acl[r][c], 0, 0, 0);
                acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][r][0], b[cur][c][0], acc[r][c], 0, 0, 0);
                acl[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cur][r][0], b[cur][c][1], acl[r][c], 0, 0, 0);
                if (DX) {
                    xcl[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ax[cur][r][1], b[cur][c][0], xcl[r][c], 0, 0, 0);
                    xcc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ax[cur][r][0], b[cur][c][0], xcc[r][c], 0, 0, 0);
                    xcl[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ax[cur][r][0], b[cur][c][1], xcl[r][c], 0, 0, 0);
                }
            }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int n = e0 + 16 * c + r16;
        if (n >= N) continue;
        const float is = inv_scale[n];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const f32x4 v = (acc[r][c] + acl[r][c] * H3_LO) * is;
            *reinterpret_cast<float4*>(dh + (size_t)n * H + u0 + 16 * r + 4 * kq) = float4{v[0], v[1], v[2], v[3]};
            if (DX) {
                const f32x4 vx = (xcc[r][c] + xcl[r][c] * H3_LO) * isc_t[n];
                *reinterpret_cast<float4*>(dx + ((size_t)n * T + t) * H + u0 + 16 * r + 4 * kq) = float4{vx[0], vx[1], vx[2], vx[3]};
            }
        }
    }
}

// x[:, t0 .. t0 + tc) f32 [N][T][I] -> per step the two fp16 piece planes of x_t in fragment order (K = IP): a wave = one
// 16-row tile x one 32-k slab, lane (kq, r16) converts 8 consecutive k of row r16 and stores 16 bytes per piece -- every
// wave store is one whole 1 KB chunk.  out[(t - t0)][frag_index(n, k, IP, piece)], rows >= N and k >= I zero.
__global__ __launch_bounds__(256) void split_x_kernel(const float* __restrict__ x, int N, int T, int I, int IP, int t0, int tc,
                                                      unsigned short* __restrict__ out) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, kq = lane >> 4;
    const int IPS = IP / 32, ntile = (N + 63) / 64 * 4;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t per_step = (int64_t)ntile * IPS;
    if (wave >= per_step * tc) return;
    const int tt = (int)(wave / per_step), rem = (int)(wave % per_step), tile = rem / IPS, s = rem % IPS;
    const int n = tile * 16 + r16, k0 = 32 * s + 8 * kq;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (n < N && k0 + i < I) ? x[((size_t)n * T + t0 + tt) * I + k0 + i] : 0.f;
    f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        _Float16 p0, p1;
        split2h(v[i], p0, p1);
        a[i] = p0; b[i] = p1;
    }
    unsigned short* o = out + (size_t)tt * 2 * ntile * 16 * IP + ((size_t)(tile * IPS + s) * 2) * 512 + lane * 8;
    *reinterpret_cast<f16x8*>(o) = a;
    *reinterpret_cast<f16x8*>(o + 512) = b;
}

// The h = 256 step kernels exist in the fp16-split form only (|w| < 65504, |x| < 4096): BOTH other modes -- exact f32 and the
// bf16 split, which callers select precisely because operands left that range -- take the generic exact-f32 step path
// (gemm_f32 + gen_cell_*), uav_lstm_bwd_caps reports 0 and the stepper refuses.
static bool h3_step_ok(int H) {
    return H == 256 && !uav_want_f32_mfma() && !uav_want_bf16x6() && !uav_debug(UAV_DEBUG_STEP_F32);
}

// the gate gradients stay in the packed form only (common.h: DgPack) unless the A/B switch asks for the round-4 f32 rows too
static bool h3_dg_packed(int H) { return h3_step_ok(H) && !uav_debug(UAV_DEBUG_DG_F32); }
bool lstm_h3_dg_packed(int H) { return h3_dg_packed(H); }

// fp16-split step path of uav_lstm_fwd for h = 256 (input projection included: the caller does NOT pre-fill the stash)
int lstm_h3_fwd(uav_ctx* ctx, const float* x, int I, const float* w_ih, const float* b_ih, const float* b_hh,
                const float* keep, const float* h0, const float* c0, const float* w_hh, int N, int T, float* y, float* hn,
                float* cn, float* stash, hipStream_t st) {
    constexpr int H = 256;
    const int64_t NH = (int64_t)N * H;
    int IP = (I + 31) / 32 * 32;
    IP = IP <= 32 ? 32 : (IP <= 64 ? 64 : (IP <= 128 ? 128 : 256));        // instantiated slab counts: 1, 2, 4, 8
    UAV_REQUIRE(I <= 256, "lstm (h=256): input width %d > 256", I);
    // tail of the workspace: hs, cs f32 | two ping-pong sets of h pieces | W_hh pieces | W_ih pieces | b_ih + b_hh
    const int64_t NP = (int64_t)(N + 63) / 64 * 64 * H;     // piece planes cover whole 64-env tiles (fragment order)
    const size_t need = (size_t)(2 * NH) * 4 + (size_t)(2 * 2 * NP) * 2 + (size_t)2 * 4 * H * H * 2 + (size_t)2 * 4 * H * IP * 2 +
                        (size_t)4 * H * 4;
    UAV_REQUIRE(need + (64u << 20) <= ctx->ws_bytes, "lstm (h=256): workspace too small");
    char* base = (char*)ctx->ws + ctx->ws_bytes - need;
    float* hs = (float*)base;
    float* cs = hs + NH;
    unsigned short* hp0 = (unsigned short*)(cs + NH);
    unsigned short* hp1 = hp0 + 2 * NP;
    unsigned short* wp = hp1 + 2 * NP;
    unsigned short* wxp = wp + (size_t)2 * 4 * H * H;
    float* bsum = (float*)(wxp + (size_t)2 * 4 * H * IP);
    const unsigned nb = (unsigned)((NH + 255) / 256);
    UAV_CHECK_HIP(hipMemsetAsync(hp0, 0, (size_t)4 * NP * 2, st));      // rows of a ragged last tile stay finite
    hipLaunchKernelGGL(split_weights_kernel, dim3(4 * H * H / 256), dim3(256), 0, st, w_hh, 4 * H, H, H, 0, wp);
    hipLaunchKernelGGL(split_weights_kernel, dim3((4 * H * IP + 255) / 256), dim3(256), 0, st, w_ih, 4 * H, I, IP, 0, wxp);
    hipLaunchKernelGGL(add2v_kernel, dim3((4 * H + 255) / 256), dim3(256), 0, st, b_ih, b_hh, bsum, 4 * H);
    hipLaunchKernelGGL(h3_init_state, dim3(nb), dim3(256), 0, st, h0, c0, keep, N, T, H, hs, cs, hp0);
    const dim3 grid((N + 63) / 64, H / 64);
    const int hslot = h3_dg_packed(H) ? 0 : 1;              // the stash's h_prev slot is only read by the f32-rows weight gradients
    // a wide input (the layer below's output, I = 64 .. 256): x is converted to piece planes a chunk of time steps at a
    // time (split_x_kernel, in front of the state in the workspace), so the step kernel's input projection streams
    // 1 KB fragment chunks like its recurrent product instead of gathering f32 rows and splitting them in every workgroup
    const int ntile = (N + 63) / 64 * 4;
    const size_t x_step = (size_t)ntile * (IP / 32) * 1024 * 2;            // bytes of one step's planes
    int TC = 0;
    if (I == IP && I >= 64 && !uav_debug(UAV_DEBUG_X_F32)) {
        const size_t room = ctx->ws_bytes - need - (64u << 20);
        TC = (int)(room / x_step < 16 ? room / x_step : 16);
    }
    unsigned short* xbuf = TC > 0 ? (unsigned short*)(base - (size_t)TC * x_step) : nullptr;
    for (int t = 0; t < T; ++t) {
        const unsigned short* xp = nullptr;
        if (TC > 0) {
            const int t0 = t / TC * TC;
            if (t == t0) {
                const int tc = T - t0 < TC ? T - t0 : TC;
                const int64_t waves = (int64_t)ntile * (IP / 32) * tc;
                hipLaunchKernelGGL(split_x_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, x, N, T, I, IP, t0, tc, xbuf);
            }
            xp = xbuf + (size_t)(t - t0) * (x_step / 2);
        }
#define LAUNCH_STEP(IPS_)                                                                                                    \
    hipLaunchKernelGGL((step_fwd_h3_kernel<H, IPS_>), grid, dim3(256), 0, st, wxp, wp, bsum, x, I, xp, 0, (t & 1) ? hp1 : hp0, \
                       (t & 1) ? hp0 : hp1, hs, cs, stash, keep, (int64_t)T, (int64_t)t, N, T, t, y, hn, cn, hslot)
        switch (IP / 32) {
            case 1: LAUNCH_STEP(1); break;
            case 2: LAUNCH_STEP(2); break;
            case 4: LAUNCH_STEP(4); break;
            default: LAUNCH_STEP(8); break;
        }
#undef LAUNCH_STEP
    }
    UAV_LAUNCH_CHECK();
    return 0;
}
bool lstm_h3_step_path(int H) { return h3_step_ok(H); }

// ---- the same step kernel, one time step per call (uav_lstm_stepper_*): an environment step can sit between two time
// steps (the rollout of a stacked / h = 256 policy), the weights are split once per rollout instead of once per step, the
// recurrent state stays in its piece planes between calls, and stash / y land in the [N][T] arrays the update's BPTT
// reads -- so PPO epoch 0 needs no forward pass.  Each step applies its OWN restart mask (keep[n][t]) to the incoming state,
// so a rollout passes the mask it learned from the previous environment step (keep_t [N]).
struct StepperLayout { size_t hs, cs, hp0, hp1, wp, wxp, bsum, total; int IP; };
static StepperLayout stepper_layout(int N, int I) {
    constexpr int H = 256;
    StepperLayout L;
    int IP = (I + 31) / 32 * 32;
    L.IP = IP <= 32 ? 32 : (IP <= 64 ? 64 : (IP <= 128 ? 128 : 256));
    const size_t NH = (size_t)N * H, NP = (size_t)(N + 63) / 64 * 64 * H;
    size_t o = 0;
    L.hs = o; o += NH * 4;
    L.cs = o; o += NH * 4;
    L.hp0 = o; o += 2 * NP * 2;
    L.hp1 = o; o += 2 * NP * 2;
    L.wp = o; o += (size_t)2 * 4 * H * H * 2;
    L.wxp = o; o += (size_t)2 * 4 * H * L.IP * 2;
    L.bsum = o; o += (size_t)4 * H * 4;
    L.total = (o + 255) / 256 * 256;
    return L;
}

extern "C" {

size_t uav_lstm_stepper_bytes(int N, int I, int H) {
    if (H != 256 || N <= 0 || I <= 0 || I > 256) return 0;
    return stepper_layout(N, I).total;
}

int uav_lstm_stepper_begin(uav_ctx* ctx, void* state, const float* w_ih, const float* w_hh, const float* b_ih,
                           const float* b_hh, const float* h0, const float* c0, int N, int I, int H, uav_stream stream) {
    UAV_REQUIRE(ctx && state && w_ih && w_hh && b_ih && b_hh && h0 && c0, "uav_lstm_stepper_begin: NULL argument");
    UAV_REQUIRE(uav_lstm_stepper_bytes(N, I, H) != 0, "uav_lstm_stepper_begin: H = %d, I = %d not supported (H = 256, I <= 256)", H, I);
    uav_enter(ctx);
    UAV_REQUIRE(h3_step_ok(H), "uav_lstm_stepper_begin: only the fp16-split arithmetic steps (uav_set_lstm_arith)");
    const StepperLayout L = stepper_layout(N, I);
    char* b = (char*)state;
    hipStream_t st = as_stream(stream);
    const int64_t NH = (int64_t)N * H;
    UAV_CHECK_HIP(hipMemsetAsync(b + L.hp0, 0, L.wp - L.hp0, st));            // rows of a ragged last tile stay finite
    hipLaunchKernelGGL(split_weights_kernel, dim3(4 * H * H / 256), dim3(256), 0, st, w_hh, 4 * H, H, H, 0, (unsigned short*)(b + L.wp));
    hipLaunchKernelGGL(split_weights_kernel, dim3((4 * H * L.IP + 255) / 256), dim3(256), 0, st, w_ih, 4 * H, I, L.IP, 0,
                       (unsigned short*)(b + L.wxp));
    hipLaunchKernelGGL(add2v_kernel, dim3((4 * H + 255) / 256), dim3(256), 0, st, b_ih, b_hh, (float*)(b + L.bsum), 4 * H);
    hipLaunchKernelGGL(h3_init_state, dim3((unsigned)((NH + 255) / 256)), dim3(256), 0, st, h0, c0, (const float*)nullptr, N, 1, H,
                       (float*)(b + L.hs), (float*)(b + L.cs), (unsigned short*)(b + L.hp0));
    UAV_LAUNCH_CHECK();
    return 0;
}

int uav_lstm_stepper_step(uav_ctx* ctx, void* state, const float* x, const void* below, const float* keep_t, int N, int T, int t,
                          int I, int H, float* y, float* stash, float* hn, float* cn, uav_stream stream) {
    UAV_REQUIRE(ctx && state && x && y && stash && hn && cn, "uav_lstm_stepper_step: NULL argument");
    UAV_REQUIRE(uav_lstm_stepper_bytes(N, I, H) != 0 && T > 0 && t >= 0 && t < T, "uav_lstm_stepper_step: bad shape (N=%d T=%d t=%d I=%d H=%d)", N, T, t, I, H);
    uav_enter(ctx);
    UAV_REQUIRE(h3_step_ok(H), "uav_lstm_stepper_step: only the fp16-split arithmetic steps (uav_set_lstm_arith)");
    const StepperLayout L = stepper_layout(N, I);
    char* b = (char*)state;
    hipStream_t st = as_stream(stream);
    const unsigned short* wxp = (const unsigned short*)(b + L.wxp);
    const unsigned short* wp = (const unsigned short*)(b + L.wp);
    const float* bsum = (const float*)(b + L.bsum);
    unsigned short* hp0 = (unsigned short*)(b + L.hp0);
    unsigned short* hp1 = (unsigned short*)(b + L.hp1);
    float* hs = (float*)(b + L.hs);
    float* cs = (float*)(b + L.cs);
    const dim3 grid((N + 63) / 64, 256 / 64);
    const int hslot = h3_dg_packed(H) ? 0 : 1;
    // `below`: the stepper state of the layer below (same N, its H = this I = 256), already stepped to t: its piece
    // planes of h_t (parity (t + 1) & 1, not yet masked) ARE this layer's input in fragment order
    const unsigned short* xp = nullptr;
    if (below) {
        UAV_REQUIRE(I == 256, "uav_lstm_stepper_step: `below` needs I = 256 (the layer below's hidden size), got %d", I);
        const StepperLayout LB = stepper_layout(N, 8);        // hp0 / hp1 offsets do not depend on the input width
        xp = (const unsigned short*)((const char*)below + (((t + 1) & 1) ? LB.hp1 : LB.hp0));
    }
#define LAUNCH_STEP(IPS_)                                                                                                     \
    hipLaunchKernelGGL((step_fwd_h3_kernel<256, IPS_>), grid, dim3(256), 0, st, wxp, wp, bsum, x, I, xp, 1, (t & 1) ? hp1 : hp0, \
                       (t & 1) ? hp0 : hp1, hs, cs, stash, keep_t, (int64_t)1, (int64_t)0, N, T, t, y, hn, cn, hslot)
    switch (L.IP / 32) {
        case 1: LAUNCH_STEP(1); break;
        case 2: LAUNCH_STEP(2); break;
        case 4: LAUNCH_STEP(4); break;
        default: LAUNCH_STEP(8); break;
    }
#undef LAUNCH_STEP
    UAV_LAUNCH_CHECK();
    return 0;
}

// the arguments of one stepper step, as uav_lstm_stepper_step forms them
static int stepper_args(const uav_stepper_call& c, int N, int T, int H, StepFwdArgs& A, int& IPS) {
    UAV_REQUIRE(c.state && c.x && c.y && c.stash && c.hn && c.cn, "uav_lstm_stepper_step_pair: NULL argument");
    UAV_REQUIRE(uav_lstm_stepper_bytes(N, c.I, H) != 0 && T > 0 && c.t >= 0 && c.t < T, "uav_lstm_stepper_step_pair: bad shape (N=%d T=%d t=%d I=%d H=%d)", N, T, c.t, c.I, H);
    const StepperLayout L = stepper_layout(N, c.I);
    char* b = (char*)c.state;
    unsigned short* hp0 = (unsigned short*)(b + L.hp0);
    unsigned short* hp1 = (unsigned short*)(b + L.hp1);
    const unsigned short* xp = nullptr;
    if (c.below) {
        UAV_REQUIRE(c.I == 256, "uav_lstm_stepper_step_pair: `below` needs I = 256, got %d", c.I);
        const StepperLayout LB = stepper_layout(N, 8);
        xp = (const unsigned short*)((const char*)c.below + (((c.t + 1) & 1) ? LB.hp1 : LB.hp0));
    }
    A = StepFwdArgs{(const unsigned short*)(b + L.wxp), (const unsigned short*)(b + L.wp), (const float*)(b + L.bsum), c.x, c.I, xp, 1,
                    (c.t & 1) ? hp1 : hp0, (c.t & 1) ? hp0 : hp1, (float*)(b + L.hs), (float*)(b + L.cs), c.stash, c.keep_t, (int64_t)1,
                    (int64_t)0, N, T, c.t, c.y, c.hn, c.cn, h3_dg_packed(H) ? 0 : 1};
    IPS = L.IP / 32;
    return 0;
}

int uav_lstm_stepper_step_pair(uav_ctx* ctx, const uav_stepper_call* a, const uav_stepper_call* b, int N, int T, int H,
                               uav_stream stream) {
    UAV_REQUIRE(ctx && a && b, "uav_lstm_stepper_step_pair: NULL argument");
    uav_enter(ctx);
    UAV_REQUIRE(h3_step_ok(H), "uav_lstm_stepper_step_pair: only the fp16-split arithmetic steps (uav_set_lstm_arith)");
    StepFwdArgs A, B;
    int ia, ib, rc;
    if ((rc = stepper_args(*a, N, T, H, A, ia)) || (rc = stepper_args(*b, N, T, H, B, ib))) return rc;
    hipStream_t st = as_stream(stream);
    if (ia == 1 && ib == 8) {                 // (layer 1 of a narrow input, layer 2 on the layer below's planes): the one instantiated pair
        hipLaunchKernelGGL((step_fwd_h3_pair_kernel<256, 1, 8>), dim3((N + 63) / 64, 256 / 64, 2), dim3(256), 0, st, A, B);
        UAV_LAUNCH_CHECK();
        return 0;
    }
    // any other pair of widths: the two steps one after the other (same results)
    if ((rc = uav_lstm_stepper_step(ctx, a->state, a->x, a->below, a->keep_t, N, T, a->t, a->I, H, a->y, a->stash, a->hn, a->cn, stream))) return rc;
    return uav_lstm_stepper_step(ctx, b->state, b->x, b->below, b->keep_t, N, T, b->t, b->I, H, b->y, b->stash, b->hn, b->cn, stream);
}

}  // extern "C"


// One layer's BPTT on the step path: its slice of the workspace, the launches before the time loop, one time step, the end.
struct H3Bwd {
    static constexpr int H = 256;
    const float *keep, *stash, *dy, *dheads, *w_head;
    int n_heads, N, T;
    float *dgates, *dx;
    bool packed;
    float *dh, *dc, *inv_scale;
    unsigned short *dgp, *wtp, *wxtp;
    // dh, dc f32 | inverse scales [masked | unmasked][NP] | W_hh^T, W_ih^T pieces [2][H][4H] each | (f32-rows form only) one step's
    // dG pieces [2][NP][4H]
    static size_t need(int N) {
        const size_t NH = (size_t)N * H, NP = (size_t)(N + 63) / 64 * 64;
        return (2 * NH) * 4 + NP * 4 * 2 + (size_t)2 * 2 * 4 * H * H * 2 + (2 * 4 * NP * H) * 2;
    }
    int prepare(char* base, const float* w_hh, const float* w_ih, const float* dhn, const float* dcn, hipStream_t st) {
        const int64_t NH = (int64_t)N * H, NP = (int64_t)(N + 63) / 64 * 64;
        dh = (float*)base;
        dc = dh + NH;
        inv_scale = dc + NH;
        wtp = (unsigned short*)(inv_scale + 2 * NP);
        wxtp = wtp + (size_t)2 * 4 * H * H;
        dgp = wxtp + (size_t)2 * 4 * H * H;
        const unsigned nb = (unsigned)((NH + 255) / 256);
        hipLaunchKernelGGL(split_weights_kernel, dim3(4 * H * H / 256), dim3(256), 0, st, w_hh, 4 * H, H, H, 1, wtp);
        if (dx) hipLaunchKernelGGL(split_weights_kernel, dim3(4 * H * H / 256), dim3(256), 0, st, w_ih, 4 * H, H, H, 1, wxtp);
        hipLaunchKernelGGL(gen_fill, dim3(nb), dim3(256), 0, st, dh, dhn, NH);
        hipLaunchKernelGGL(gen_fill, dim3(nb), dim3(256), 0, st, dc, dcn, NH);
        return 0;
    }
    // a time step = the gate-gradient kernel (HBM-bound) + the recurrent product (latency-bound): two launches, so that the
    // pipeline of a stack can put an event between them
    void step(int t, hipStream_t st) const { cell(t, st); prod(t, st); }
    void cell(int t, hipStream_t st) const {
        const DgPack P(N, T);
        // the gate-gradient kernel's grid covers whole 64-env tiles: rows past N are written as zero pieces, scale 0
        unsigned short* pc = packed ? P.pieces((void*)dgates, t) : dgp;
        float* isc_t = packed ? P.isc((void*)dgates, t) : inv_scale + P.NP;
        hipLaunchKernelGGL((cell_bwd_h3_kernel<H, CELL_EPW>), dim3(P.RT * (16 / CELL_EPW)), dim3(CELL_EPW * 64), CELL_BWD_LDS, st, stash, keep, dy, dheads, w_head,
                           n_heads, N, T, t, dh, dc, packed ? (float*)nullptr : dgates, pc, inv_scale, isc_t,
                           packed ? P.iscm((void*)dgates, t) : (float*)nullptr);
    }
    void prod(int t, hipStream_t st) const {
        const dim3 grid((N + 63) / 64, H / 64);
        const DgPack P(N, T);
        unsigned short* pc = packed ? P.pieces((void*)dgates, t) : dgp;
        float* isc_t = packed ? P.isc((void*)dgates, t) : inv_scale + P.NP;
        if (dx) hipLaunchKernelGGL((step_bwd_h3_kernel<H, true>), grid, dim3(256), 0, st, wtp, pc, inv_scale, isc_t, N, dh, wxtp, dx, T, t);
        else hipLaunchKernelGGL((step_bwd_h3_kernel<H, false>), grid, dim3(256), 0, st, wtp, pc, inv_scale, isc_t, N, dh,
                                (const unsigned short*)nullptr, (float*)nullptr, T, t);
    }
    void finish(float* dh0, float* dc0, hipStream_t st) const {
        const int64_t NH = (int64_t)N * H;
        const unsigned nb = (unsigned)((NH + 255) / 256);
        if (dh0) hipLaunchKernelGGL(gen_fill, dim3(nb), dim3(256), 0, st, dh0, dh, NH);
        if (dc0) hipLaunchKernelGGL(gen_fill, dim3(nb), dim3(256), 0, st, dc0, dc, NH);
    }
};
static int h3_bwd_attr() {
    UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(&cell_bwd_h3_kernel<256, CELL_EPW>), (int)CELL_BWD_LDS));
    return 0;
}

static int lstm_h3_bwd(uav_ctx* ctx, const float* keep, const float* stash, const float* w_hh, const float* dy,
                       const float* dheads, const float* w_head, int n_heads,
                       const float* dhn, const float* dcn, int N, int T, float* dgates, float* dh0, float* dc0,
                       const float* w_ih, float* dx, hipStream_t st) {
    const size_t need = H3Bwd::need(N);
    UAV_REQUIRE(need + (64u << 20) <= ctx->ws_bytes, "lstm (h=256): workspace too small");
    int rc;
    if ((rc = h3_bwd_attr())) return rc;
    H3Bwd L{keep, stash, dy, dheads, w_head, n_heads, N, T, dgates, dx, h3_dg_packed(H3Bwd::H)};
    uav_dg_record(ctx, dgates, L.packed);
    if ((rc = L.prepare((char*)ctx->ws + ctx->ws_bytes - need, w_hh, w_ih, dhn, dcn, st))) return rc;
    for (int t = T - 1; t >= 0; --t) L.step(t, st);
    L.finish(dh0, dc0, st);
    UAV_LAUNCH_CHECK();
    return 0;
}

// The BPTTs of a stack of layers as a pipeline: the layer below can run step t as soon as the layer above has produced
// dx[:, t] (its step_bwd<DX>), so each layer gets its own stream and follows the one above by one step.  The gate-gradient
// kernel is HBM-bound and the recurrent product waits on L2 latency: two layers' kernels share the CUs and overlap (two
// independent passes on two streams: 53.9 us per step pair against 63.9 on one, tools/perf_bwd_overlap.py).  Same kernels,
// same arithmetic, same results as one uav_lstm_bwd per layer, top down.
int lstm_h3_bwd_stack(uav_ctx* ctx, int nl, const uav_lstm_bwd_layer* layers, const float* dy, const float* dheads,
                      const float* w_head, int n_heads, int N, int T, hipStream_t st) {
    UAV_REQUIRE(nl >= 1 && nl <= 4, "uav_lstm_bwd_stack: 1..4 layers, got %d", nl);
    const size_t need = H3Bwd::need(N);
    UAV_REQUIRE((size_t)nl * need + (64u << 20) <= ctx->ws_bytes, "uav_lstm_bwd_stack: workspace too small for %d layers", nl);
    int rc;
    if ((rc = h3_bwd_attr())) return rc;
    // every argument is checked BEFORE the side streams are forked: an error return must not leave work queued on them
    for (int l = 0; l < nl; ++l) {
        const uav_lstm_bwd_layer& a = layers[l];
        UAV_REQUIRE(a.stash && a.w_hh && a.dgates, "uav_lstm_bwd_stack: layer %d: NULL stash / w_hh / dgates", l);
        UAV_REQUIRE(l + 1 == nl || (a.w_ih && a.dx), "uav_lstm_bwd_stack: layer %d feeds the layer below: w_ih and dx are required", l);
    }
    for (int l = 0; l + 1 < nl; ++l)
        if (!ctx->side[l]) UAV_CHECK_HIP(hipStreamCreateWithFlags(&ctx->side[l], hipStreamNonBlocking));
    for (auto& e : ctx->side_ev)
        if (!e) UAV_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipStream_t str[4] = {st, ctx->side[0], ctx->side[1], ctx->side[2]};
    // fork: the side streams start behind everything the caller has queued
    UAV_CHECK_HIP(hipEventRecord(ctx->side_ev[0], st));
    for (int l = 1; l < nl; ++l) UAV_CHECK_HIP(hipStreamWaitEvent(str[l], ctx->side_ev[0], 0));
    // join: the caller's stream continues behind everything queued on the side streams (also on the error paths below)
    auto join = [&]() -> int {
        for (int l = 1; l < nl; ++l) {
            UAV_CHECK_HIP(hipEventRecord(ctx->side_ev[1], str[l]));
            UAV_CHECK_HIP(hipStreamWaitEvent(st, ctx->side_ev[1], 0));
        }
        return 0;
    };
    H3Bwd L[4];
    for (int l = 0; l < nl; ++l) {
        const uav_lstm_bwd_layer& a = layers[l];
        L[l] = H3Bwd{a.keep, a.stash, l == 0 ? dy : layers[l - 1].dx, l == 0 ? dheads : nullptr, l == 0 ? w_head : nullptr,
                     l == 0 ? n_heads : 0, N, T, a.dgates, a.dx, h3_dg_packed(H3Bwd::H)};
        uav_dg_record(ctx, a.dgates, L[l].packed);
        if ((rc = L[l].prepare((char*)ctx->ws + ctx->ws_bytes - (size_t)(l + 1) * need, a.w_hh, a.w_ih, a.dhn, a.dcn, str[l]))) {
            (void)join();
            return rc;
        }
    }
    // the wave front: at round r layer l runs its step T - 1 - (r - l); the hand-off events form a ring per boundary.
    // (The layers run in lockstep -- a layer's step t starts when the layer above ends its step t and starts its step t - 1, so
    //  gate-gradient kernel meets gate-gradient kernel and product meets product.  Shifting the layer below by half a step with a
    //  second event per step, so that an HBM-bound kernel always runs beside a latency-bound one, was measured in round 5 and is
    //  SLOWER: 14.4 against 12.4 ms per C5 epoch -- the streaming kernel evicts the L2 lines the product lives on.)
    hipEvent_t* ring = ctx->side_ev + 2;                       // 6 events: two per boundary
    for (int r = 0; r < T + nl - 1; ++r)
        for (int l = 0; l < nl; ++l) {
            const int t = T - 1 - (r - l);
            if (t < 0 || t >= T) continue;
            if (l > 0) UAV_CHECK_HIP(hipStreamWaitEvent(str[l], ring[2 * (l - 1) + (t & 1)], 0));   // dx[:, t] of the layer above
            L[l].step(t, str[l]);
            if (l + 1 < nl) UAV_CHECK_HIP(hipEventRecord(ring[2 * l + (t & 1)], str[l]));
        }
    for (int l = 0; l < nl; ++l) L[l].finish(layers[l].dh0, layers[l].dc0, str[l]);
    if ((rc = join())) return rc;
    UAV_LAUNCH_CHECK();
    return 0;
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

// the step path forms dx = dG W_ih itself (no separate GEMM) when the input is as wide as the state
bool lstm_h3_bwd_fuses_dx(int I, int H) { return h3_step_ok(H) && I == H; }

// what the generic / step paths of uav_lstm_bwd can do themselves: bit 0 form dx (I == H on the step path), bit 1 take dheads
int lstm_generic_bwd_caps(int I, int H) { return h3_step_ok(H) ? ((I == H ? 1 | 4 : 0) | 2) : 0; }
bool lstm_h3_stack_ok(int H) { return h3_step_ok(H); }

int lstm_generic_bwd(uav_ctx* ctx, const float* keep, const float* stash, const float* w_hh, const float* dy,
                     const float* dheads, const float* w_head, int n_heads,
                     const float* dhn, const float* dcn, int N, int T, int H, float* dgates, float* dh0, float* dc0,
                     const float* w_ih, int I, float* dx, hipStream_t st) {
    UAV_REQUIRE(dy || h3_step_ok(H), "uav_lstm_bwd: this hidden size / arithmetic takes dy (form dheads . w_head with uav_gemm_f32); "
                "see uav_lstm_bwd_caps");
    UAV_REQUIRE(!dx || (w_ih && lstm_h3_bwd_fuses_dx(I, H)), "uav_lstm_bwd: dx is formed here only when uav_lstm_bwd_caps(ctx, I, H) "
                "says so (H = 256 = I on the fp16-split arithmetic); otherwise ask uav_lstm_wgrad for it");
    if (h3_step_ok(H)) return lstm_h3_bwd(ctx, keep, stash, w_hh, dy, dheads, w_head, n_heads, dhn, dcn, N, T, dgates, dh0, dc0, w_ih, dx, st);
    const int64_t NH = (int64_t)N * H;
    UAV_REQUIRE((size_t)(2 * NH) * sizeof(float) + (64u << 20) <= ctx->ws_bytes, "lstm (generic): workspace too small");
    uav_dg_record(ctx, dgates, 0);
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

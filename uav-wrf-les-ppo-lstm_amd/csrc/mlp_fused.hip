// mlp_fused.hip -- the reference's OWN policy (PPOV2.0/model.py:17-53: Linear(6,256) -> LayerNorm -> ReLU ->
// Linear(256,128) -> LayerNorm -> ReLU -> actor Linear(128,5) | critic Linear(128,1)) as two persistent kernels that keep
// every activation on chip:
//
//   rollout_mlp_kernel   R1 for the MLP policy (train_ppo2.0.py:157-198 for N environments x T steps in ONE launch):
//                        policy forward -> Categorical sample -> env step -> store, 16 environments per workgroup.
//   mlp_ppo_grad_kernel  U1-U3 up to the gradient (train_ppo2.0.py:55-86): forward, clipped-PPO loss, backward through the
//                        whole network for tiles of 32 samples; LayerNorm statistics, activations and their gradients
//                        never leave the CU, weight gradients accumulate in registers over all of a workgroup's tiles and
//                        leave as one slab per workgroup.  HBM traffic = the 44 algorithmic bytes per sample.
//
// Arithmetic: exact f32 (v_mfma_f32_16x16x4_f32 = a k-ordered fmaf chain), the reference's dtype; products are laid out
// "units on rows, samples on columns" (weights are the A operand), so the accumulator of a lane holds 4 consecutive units
// of ONE sample and LayerNorm's per-sample reductions are a 4-lane shuffle + one LDS exchange between the 8 waves.
// Both kernels run the same forward code in the same order: the rollout's log-probabilities are bit-identical to the
// update's first forward pass.
#include "env_core.h"
#include "loss_core.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

int env_params_from_cfg(const uav_ctx* ctx, const uav_env_cfg* cfg, int n_env, EnvParams& P);
int launch_loss_final(double* partial, int nb, int n_heads, double* loss_sums, float* dhead_bias, hipStream_t st);

#ifdef UAV_X6_PROFILE
// phase timing (instrumented build only, tools/build_prof.sh): cycles per phase summed over the tiles of workgroup 0,
// waves 0 (loss wave) and 1
__device__ unsigned long long g_mlp_prof[2][16];
#define M_PROF_DECL unsigned long long pm_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pl_ = __builtin_readcyclecounter()
#define M_PROF_MARK(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); pm_[i] += n_ - pl_; pl_ = n_; } while (0)
#define M_PROF_FLUSH() do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 2) for (int i_ = 0; i_ < 16; ++i_) g_mlp_prof[threadIdx.x >> 6][i_] = pm_[i_]; } while (0)
extern "C" int uav_mlp_prof_read(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_prof), sizeof(g_mlp_prof)) == hipSuccess ? 0 : 1;
}
#else
#define M_PROF_DECL
#define M_PROF_MARK(i)
#define M_PROF_FLUSH()
#endif

namespace {

constexpr int MT = 16;                  // samples (environments) per tile = one MFMA column tile
constexpr int IN = 6, H1 = 256, H2 = 128, NA = 5, NH = 6;
constexpr int NWAVE = 8;
constexpr int AS1 = 260, AS2 = 132;     // row strides (floats) of the a1 / a2 tiles: float4-aligned rows
constexpr int WS2 = 260;                // row stride of the rollout's LDS image of wave 0's W2 rows (float4-aligned)
constexpr int RS1 = H1 + 8;             // row stride (halves) of the a1 piece planes: conflict-free ds_read_b128
constexpr int RS2 = H2 + 8;             // row stride (halves) of the dz2 piece planes
constexpr float LN_EPS_F = 1e-5f;
constexpr float R_EPS = 1.1920928955078125e-07f;

// flat parameter layout of csrc/mlp.hip: W1 b1 g1 be1 W2 b2 g2 be2 Wh bh
constexpr int O_W1 = 0, O_B1 = O_W1 + H1 * IN, O_G1 = O_B1 + H1, O_BE1 = O_G1 + H1, O_W2 = O_BE1 + H1,
              O_B2 = O_W2 + H2 * H1, O_G2 = O_B2 + H2, O_BE2 = O_G2 + H2, O_WH = O_BE2 + H2, O_BH = O_WH + NH * H2,
              NPARAM = O_BH + NH;
static_assert(NPARAM == 36230, "MLP parameter count (SURVEY 8a M1)");

// LDS regions common to both kernels (floats).  NC = column tiles (of 16 samples) a workgroup advances together: 1 in the
// rollout (16 environments per workgroup keep all 256 CUs busy at 4096 environments), 2 in the update (every weight
// fragment fetched from L2 then feeds two MFMAs, and the per-tile latencies -- barriers, the loss lanes -- are shared by 32
// samples).
template <int NC>
struct Tiles {
    static constexpr int MS = MT * NC;
    float* prm;      // b1 g1 be1 [256 each] | b2 g2 be2 [128 each]
    float* X;        // [MS][8]   observations of the tile, columns 6, 7 zero
    float* A1;       // [MS][AS1] a1 = relu(LN1(z1)); later dz1
    float* A2;       // [MS][AS2] a2 = relu(LN2(z2)); later dz2
    double* red;     // [2][8 waves][MS] per-wave partial sums of the LayerNorm reductions
    f32x4* HP;       // [8 waves][NC][32] partial head tiles (K split over the waves; rows 0..7 = lanes with kq < 2)
    float* HD;       // [MS][8]   heads: logits | value
    unsigned short* P1;   // [2 pieces][MS][RS1] a1 as two fp16 planes (the B operand of the split layer-2 product)
    static constexpr int FLOATS = 3 * H1 + 3 * H2 + MS * 8 + MS * AS1 + MS * AS2 + 2 * (2 * NWAVE * MS) + NWAVE * NC * 32 * 4 + MS * 8 + MS * RS1;
    __device__ __forceinline__ explicit Tiles(float* base) {
        prm = base;
        X = prm + 3 * H1 + 3 * H2;
        A1 = X + MS * 8;
        A2 = A1 + MS * AS1;
        red = reinterpret_cast<double*>(A2 + MS * AS2);
        HP = reinterpret_cast<f32x4*>(red + 2 * NWAVE * MS);
        HD = reinterpret_cast<float*>(HP + NWAVE * NC * 32);
        P1 = reinterpret_cast<unsigned short*>(HD + MS * 8);
    }
};

__device__ __forceinline__ f32x4 ld4(const float* p) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    return f32x4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void st4(float* p, const f32x4& v) { *reinterpret_cast<float4*>(p) = float4{v[0], v[1], v[2], v[3]}; }
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// sum over the 16 lanes of a row (the 16 samples lane & 15 = 0..15 that share lane >> 4): four DPP adds -- xor 1, xor 2,
// half-row mirror, row mirror -- every lane ends with the same total (a + b == b + a bitwise at every stage)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));   // row_mirror
    return v;
}
__device__ __forceinline__ f32x4 row16_sum(f32x4 v) { return f32x4{row16_sum(v[0]), row16_sum(v[1]), row16_sum(v[2]), row16_sum(v[3])}; }

// max over all 64 lanes of a wave (DPP within the rows of 16, then the two row swaps); every lane gets the result
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true)));
    const auto s16 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    v = fmaxf(__builtin_bit_cast(float, (unsigned)s16[0]), __builtin_bit_cast(float, (unsigned)s16[1]));
    const auto s32 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)s32[0]), __builtin_bit_cast(float, (unsigned)s32[1]));
}

// Transposed LDS read (gfx950 ds_read_b64_tr_b16): an MFMA 16x16x32 fragment -- 8 consecutive K per lane -- from a plane
// stored [K rows][16+ columns of halves] with row stride `rs` halves: lane (i = lane & 15, kq) gets rows 8 kq .. 8 kq + 7 of
// column i.  Lane 4 q + p of a 16-lane group supplies the address of row q, columns 4 p .. 4 p + 3 of each 4-row block.
// EXEC must be all ones (the gather crosses lanes): only called from uniform code.
__device__ __forceinline__ f16x8 tr_frag(const unsigned short* plane, int rs, int lane) {
    typedef short v4s __attribute__((__vector_size__(4 * sizeof(short))));
    typedef __attribute__((address_space(3))) v4s lds_v4s;
    const int li = lane & 15, kq = lane >> 4;
    const unsigned short* a = plane + (8 * kq + (li >> 2)) * rs + 4 * (li & 3);
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)a);
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(a + 4 * rs));
    typedef short v8s __attribute__((__vector_size__(8 * sizeof(short))));
    const v8s r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(f16x8, r);
}

// sum over the four lanes that hold the same sample (lane & 15) of a wave
// (the two gfx950 row-swap instructions instead of ds_bpermute: v_permlane16_swap / v_permlane32_swap hand every lane its
//  xor-16 / xor-32 partner's dword without touching LDS; a + b == b + a bitwise, so the four lanes agree)
__device__ __forceinline__ float quad_sum(float v) {
    {
        const auto s = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
        v = __builtin_bit_cast(float, (unsigned)s[0]) + __builtin_bit_cast(float, (unsigned)s[1]);
    }
    {
        const auto s = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
        v = __builtin_bit_cast(float, (unsigned)s[0]) + __builtin_bit_cast(float, (unsigned)s[1]);
    }
    return v;
}
__device__ __forceinline__ double quad_sum(double v) {
    {
        const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
        const double x = __builtin_bit_cast(double, (unsigned long long)(unsigned)lo[0] | ((unsigned long long)(unsigned)hi[0] << 32));
        const double y = __builtin_bit_cast(double, (unsigned long long)(unsigned)lo[1] | ((unsigned long long)(unsigned)hi[1] << 32));
        v = x + y;
    }
    {
        const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
        const double x = __builtin_bit_cast(double, (unsigned long long)(unsigned)lo[0] | ((unsigned long long)(unsigned)hi[0] << 32));
        const double y = __builtin_bit_cast(double, (unsigned long long)(unsigned)lo[1] | ((unsigned long long)(unsigned)hi[1] << 32));
        v = x + y;
    }
    return v;
}

// One LayerNorm reduction round: every wave leaves (a, b) for its sample columns, then every lane reads the 8 partial
// pairs of ITS samples in a fixed order.  One barrier inside; the caller syncs before `red` is written again.
template <int NC>
__device__ __forceinline__ void ln_exchange(const Tiles<NC>& L, int w, int j, int kq, const double (&a)[NC], const double (&b)[NC],
                                            double (&A)[NC], double (&B)[NC]) {
    constexpr int MS = MT * NC;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const double ta = quad_sum(a[c]), tb = quad_sum(b[c]);
        if (kq == 0) {
            L.red[w * MS + 16 * c + j] = ta;
            L.red[(NWAVE + w) * MS + 16 * c + j] = tb;
        }
    }
    lds_barrier();
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        A[c] = 0.0;
        B[c] = 0.0;
#pragma unroll
        for (int i = 0; i < NWAVE; ++i) {
            A[c] += L.red[i * MS + 16 * c + j];
            B[c] += L.red[(NWAVE + i) * MS + 16 * c + j];
        }
    }
}

// The same exchange in f32 for the LayerNorm BACKWARD sums (mean of dxhat, mean of dxhat * xhat): they enter the result
// linearly, so there is no cancellation to protect and the reference's own backward accumulates them in f32.
template <int NC>
__device__ __forceinline__ void ln_exchange_f32(const Tiles<NC>& L, int w, int j, int kq, const float (&a)[NC], const float (&b)[NC],
                                                float (&A)[NC], float (&B)[NC]) {
    constexpr int MS = MT * NC;
    float* red = reinterpret_cast<float*>(L.red);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float ta = quad_sum(a[c]), tb = quad_sum(b[c]);
        if (kq == 0) {
            red[w * MS + 16 * c + j] = ta;
            red[(NWAVE + w) * MS + 16 * c + j] = tb;
        }
    }
    lds_barrier();
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        A[c] = 0.f;
        B[c] = 0.f;
#pragma unroll
        for (int i = 0; i < NWAVE; ++i) {
            A[c] += red[i * MS + 16 * c + j];
            B[c] += red[(NWAVE + i) * MS + 16 * c + j];
        }
    }
}

// ---------------------------------------------------------------------------------------------------- forward
// z1 -> LN1 -> a1 (to LDS).  Lane (j = lane & 15, kq = lane >> 4) of wave w holds units 32 w + 16 t + 4 kq + r, t < 2,
// r < 4 of samples 16 c + j.  Leaves xhat1 / rstd1 for the backward.  One barrier inside (ln_exchange); the caller's
// barrier after it publishes A1.
// MODE: 0 = a1 as f32 rows (A1), 1 = a1 as two fp16 piece planes (P1), 2 = both.
template <int NC, int MODE = 0>
__device__ __forceinline__ void layer1(const Tiles<NC>& L, const float (&w1a)[2][2], int w, int j, int kq, f32x4 (&xh)[NC][2],
                                       float (&rstd)[NC]) {
    f32x4 z[NC][2];
    double s[NC], q[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float xb0 = L.X[(16 * c + j) * 8 + kq], xb1 = L.X[(16 * c + j) * 8 + 4 + kq];
        s[c] = 0.0;
        q[c] = 0.0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int u = 32 * w + 16 * t + 4 * kq;
            f32x4 acc = ld4(L.prm + u);                      // b1
            acc = mfma4(w1a[t][0], xb0, acc);
            acc = mfma4(w1a[t][1], xb1, acc);
            z[c][t] = acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[c] += (double)acc[r];
                q[c] += (double)acc[r] * (double)acc[r];
            }
        }
    }
    double S[NC], Q[NC];
    ln_exchange<NC>(L, w, j, kq, s, q, S, Q);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const double mean = S[c] * (1.0 / H1);
        double var = Q[c] * (1.0 / H1) - mean * mean;        // f64: no cancellation at f32 data precision
        var = var > 0.0 ? var : 0.0;
        const float mf = (float)mean;
        rstd[c] = 1.0f / sqrtf((float)var + LN_EPS_F);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int u = 32 * w + 16 * t + 4 * kq;
            const f32x4 g = ld4(L.prm + H1 + u), be = ld4(L.prm + 2 * H1 + u);
            f32x4 a;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xh[c][t][r] = (z[c][t][r] - mf) * rstd[c];
                const float pre = xh[c][t][r] * g[r] + be[r];
                a[r] = pre < 0.f ? 0.f : pre;                // NaN propagates like torch.relu
            }
            if (MODE != 1) st4(L.A1 + (16 * c + j) * AS1 + u, a);
            if (MODE != 0) {
                unsigned short b[2][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    _Float16 p0, p1;
                    split2h(a[r], p0, p1);
                    b[0][r] = h_bits(p0); b[1][r] = h_bits(p1);
                }
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
                    uint2 v;
                    v.x = (unsigned)b[pc][0] | ((unsigned)b[pc][1] << 16);
                    v.y = (unsigned)b[pc][2] | ((unsigned)b[pc][3] << 16);
                    *reinterpret_cast<uint2*>(L.P1 + (pc * (MT * NC) + 16 * c + j) * RS1 + u) = v;
                }
            }
        }
    }
}

// z2 -> LN2 -> a2 (to LDS).  Lane holds units 16 w + 4 kq + r of samples 16 c + j.  The K = 256 sum runs in 16 slabs of 16:
// in slab s the lane group kq multiplies k = 16 s + 4 kq + i, i < 4 -- a permutation of the MFMA's natural k order (any
// order is a valid dot product as long as A and B agree) that makes BOTH operands four consecutive floats per lane:
// wa4(s) = W2[16 w + (lane & 15)][16 s + 4 kq .. + 3] is ONE dwordx4 load (registers, LDS or L2), the a1 fragment one
// ds_read_b128.
template <int NC, class WA>
__device__ __forceinline__ void layer2(const Tiles<NC>& L, WA&& wa4, int w, int j, int kq, f32x4 (&xh)[NC], float (&rstd)[NC]) {
    const int u = 16 * w + 4 * kq;
    f32x4 acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = ld4(L.prm + 3 * H1 + u);       // b2
    const float* brow = L.A1 + j * AS1 + 4 * kq;
    // four slabs of weights in flight ahead of the MFMAs that use them (L2 latency), never all sixteen (registers)
    f32x4 wq[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wq[0][i] = wa4(i);
#pragma unroll
    for (int ch = 0; ch < H1 / 64; ++ch) {
        if (ch + 1 < H1 / 64) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wq[(ch + 1) & 1][i] = wa4(4 * (ch + 1) + i);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 a = wq[ch & 1][i];
            f32x4 b[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) b[c] = ld4(brow + 16 * c * AS1 + 16 * (4 * ch + i));
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] = mfma4(a[k], b[c][k], acc[c]);
        }
        asm volatile("" ::: "memory");
    }
    double s1[NC], q1[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        s1[c] = 0.0;
        q1[c] = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s1[c] += (double)acc[c][r];
            q1[c] += (double)acc[c][r] * (double)acc[c][r];
        }
    }
    double S[NC], Q[NC];
    ln_exchange<NC>(L, w, j, kq, s1, q1, S, Q);
    const f32x4 g = ld4(L.prm + 3 * H1 + H2 + u), be = ld4(L.prm + 3 * H1 + 2 * H2 + u);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const double mean = S[c] * (1.0 / H2);
        double var = Q[c] * (1.0 / H2) - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float mf = (float)mean;
        rstd[c] = 1.0f / sqrtf((float)var + LN_EPS_F);
        f32x4 a;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xh[c][r] = (acc[c][r] - mf) * rstd[c];
            const float pre = xh[c][r] * g[r] + be[r];
            a[r] = pre < 0.f ? 0.f : pre;
        }
        st4(L.A2 + (16 * c + j) * AS2 + u, a);
    }
}

// The same layer on the fp16 matrix pipe at f32 accuracy (common.h split2h: two fp16 pieces per operand, three products per
// K = 32 slab into a main and a cross accumulator): 8 slabs x 3 x 16 cycles instead of 64 k-steps x 32.  wa8(s, piece) =
// the A fragment W2[16 w + (lane & 15)][32 s + 8 kq .. + 7] as fp16 piece `piece`; B fragments from the a1 piece planes.
// Operand ranges: |W2| < 65504 and a1 <= sqrt(255) |g1| + |be1| < 65504, i.e. max |param| < 2048 (the caller's guard).
template <int NC, class WA>
__device__ __forceinline__ void layer2_h3(const Tiles<NC>& L, WA&& wa8, int w, int j, int kq, f32x4 (&xh)[NC], float (&rstd)[NC]) {
    constexpr int MS = MT * NC;
    const int u = 16 * w + 4 * kq;
    f32x4 acc[NC], acl[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        acc[c] = ld4(L.prm + 3 * H1 + u);                    // b2
        acl[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const unsigned short* brow = L.P1 + j * RS1 + 8 * kq;
    // two slabs of weight fragments in flight ahead of the MFMAs that use them (L2 latency in the update kernel)
    f16x8 wq[2][2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { wq[0][i][0] = wa8(i, 0); wq[0][i][1] = wa8(i, 1); }
#pragma unroll
    for (int ch = 0; ch < H1 / 64; ++ch) {
        if (ch + 1 < H1 / 64) {
#pragma unroll
            for (int i = 0; i < 2; ++i) { wq[(ch + 1) & 1][i][0] = wa8(2 * (ch + 1) + i, 0); wq[(ch + 1) & 1][i][1] = wa8(2 * (ch + 1) + i, 1); }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int s = 2 * ch + i;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const f16x8 b0 = *reinterpret_cast<const f16x8*>(brow + (16 * c) * RS1 + 32 * s);
                const f16x8 b1 = *reinterpret_cast<const f16x8*>(brow + (MS + 16 * c) * RS1 + 32 * s);
                acl[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[ch & 1][i][1], b0, acl[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[ch & 1][i][0], b0, acc[c], 0, 0, 0);
                acl[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[ch & 1][i][0], b1, acl[c], 0, 0, 0);
            }
        }
        asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = acc[c] + acl[c] * H3_LO;
    double s1[NC], q1[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        s1[c] = 0.0;
        q1[c] = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s1[c] += (double)acc[c][r];
            q1[c] += (double)acc[c][r] * (double)acc[c][r];
        }
    }
    double S[NC], Q[NC];
    ln_exchange<NC>(L, w, j, kq, s1, q1, S, Q);
    const f32x4 g = ld4(L.prm + 3 * H1 + H2 + u), be = ld4(L.prm + 3 * H1 + 2 * H2 + u);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const double mean = S[c] * (1.0 / H2);
        double var = Q[c] * (1.0 / H2) - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float mf = (float)mean;
        rstd[c] = 1.0f / sqrtf((float)var + LN_EPS_F);
        f32x4 a;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xh[c][r] = (acc[c][r] - mf) * rstd[c];
            const float pre = xh[c][r] * g[r] + be[r];
            a[r] = pre < 0.f ? 0.f : pre;
        }
        st4(L.A2 + (16 * c + j) * AS2 + u, a);
    }
}

// heads = Wh a2 + bh with K = 128 split over the 8 waves (16 each); wave 0 adds the partial tiles in a fixed order and
// leaves heads[sample][0..5] in HD.  wh[s] = Wh[lane & 15][16 w + 4 s + kq] (0 for rows >= 6).  Barrier inside.
template <int NC>
__device__ __forceinline__ void heads_fwd(const Tiles<NC>& L, const float (&wh)[4], const float* __restrict__ bh, int w, int lane) {
    const int j = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        f32x4 hp = {0.f, 0.f, 0.f, 0.f};
        const float* brow = L.A2 + (16 * c + j) * AS2 + 16 * w + kq;
#pragma unroll
        for (int s = 0; s < 4; ++s) hp = mfma4(wh[s], brow[4 * s], hp);
        if (kq < 2) L.HP[(w * NC + c) * 32 + lane] = hp;
    }
    lds_barrier();
    if (w == 0 && kq < 2) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            f32x4 t = L.HP[c * 32 + lane];
#pragma unroll
            for (int i = 1; i < NWAVE; ++i) t = t + L.HP[(i * NC + c) * 32 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int hd = 4 * kq + r;
                L.HD[(16 * c + j) * 8 + hd] = t[r] + (hd < NH ? bh[hd] : 0.f);
            }
        }
    }
}

// ==================================================================================================== rollout
struct MlpRollBufs {
    float* cur_obs; float* obs; int32_t* act; float* rew; float* val; float* logp; float* done; uint8_t* flags;
    float* last_val; const int32_t* forced_act; const double* noise; int32_t* nan_count; float* info; float* heads;
};

constexpr size_t ROLL_LDS = (size_t)(Tiles<1>::FLOATS + 16 * WS2) * sizeof(float);

// H3: the 256 x 128 layer on the fp16 matrix pipe (layer2_h3) -- the update kernel of the same arithmetic runs the same
// code, so the rollout's log-probabilities stay bit-identical to the update's first forward pass.
template <bool H3>
__global__ __launch_bounds__(512) void rollout_mlp_kernel(EnvParams P_arg, EnvBlob blob, int N, int T, uint64_t iter,
                                                            const float* __restrict__ params, MlpRollBufs B) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Tiles<1> L(smem);
    float* W2i = smem + Tiles<1>::FLOATS;                        // [16][WS2] rows 0..15 of W2: wave 0 (the env role) keeps no weight VGPRs
    __shared__ unsigned short vis[MT * NVIS];
    __shared__ EnvState es_s[MT];
    __shared__ double env_tab[ENV_LDS_TABLE_DOUBLES];            // pow(vc, 0.75) | ripple factors (env_core.h)
    static_assert(ROLL_LDS + sizeof(vis) + sizeof(es_s) + sizeof(env_tab) <= 160 * 1024, "rollout_mlp_kernel: LDS over 160 KB per workgroup");
    EnvParams P = P_arg;
    env_params_refresh(P);
    env_tables_to_lds(P, env_tab, threadIdx.x, 512);

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * MT;
    const int my_env = n0 + lane;
    const bool env_lane = (w == 0) && lane < MT && my_env < N;
    unsigned short* myvis = vis + (lane & 15) * NVIS;

    // ---- parameters
    for (int i = threadIdx.x; i < H1; i += 512) {
        L.prm[i] = params[O_B1 + i];
        L.prm[H1 + i] = params[O_G1 + i];
        L.prm[2 * H1 + i] = params[O_BE1 + i];
    }
    for (int i = threadIdx.x; i < H2; i += 512) {
        L.prm[3 * H1 + i] = params[O_B2 + i];
        L.prm[3 * H1 + H2 + i] = params[O_G2 + i];
        L.prm[3 * H1 + 2 * H2 + i] = params[O_BE2 + i];
    }
    unsigned short* W2p = reinterpret_cast<unsigned short*>(W2i);   // H3: rows 0..15 as piece fragments [8 slabs][2][64 lanes][8]
    if (H3) {
        for (int i = threadIdx.x; i < 16 * H1; i += 512) {
            const int row = i >> 8, k = i & 255, sl = k >> 5, q8 = (k >> 3) & 3, e = k & 7;
            _Float16 p0, p1;
            split2h(params[O_W2 + i], p0, p1);
            unsigned short* d = W2p + ((sl * 2) * 64 + q8 * 16 + row) * 8 + e;
            d[0] = h_bits(p0);
            d[64 * 8] = h_bits(p1);
        }
    } else {
        for (int i = threadIdx.x; i < 16 * H1; i += 512) W2i[(i >> 8) * WS2 + (i & 255)] = params[O_W2 + i];
    }
    float w1a[2][2], wh[4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int k = 4 * s + kq;
            w1a[t][s] = k < IN ? params[O_W1 + (32 * w + 16 * t + j) * IN + k] : 0.f;
        }
#pragma unroll
    for (int s = 0; s < 4; ++s) wh[s] = j < NH ? params[O_WH + j * H2 + 16 * w + 4 * s + kq] : 0.f;
    const float* bh = params + O_BH;

    if (w == 0 && lane < MT) {
        const int n = min(my_env, N - 1);
        es_s[lane] = env_load(blob, n);
        for (int k = 0; k < NVIS; ++k) myvis[k] = blob.visited[(size_t)n * NVIS + k];
#pragma unroll
        for (int f = 0; f < 8; ++f) L.X[lane * 8 + f] = f < IN ? B.cur_obs[(size_t)n * IN + f] : 0.f;
    }
    lds_barrier();

    const float* w2row = W2i + j * WS2 + 4 * kq;
    const int steps = T + (B.last_val ? 1 : 0);              // one extra value-only pass for V(s_T)
    if (w != 0) {
        // ------------------------------------------------------------------ waves 1..7: their rows of W2 in registers
        f32x4 wA[H3 ? 1 : H1 / 16];
        f16x8 wP[H3 ? H1 / 32 : 1][2];
        if (H3) {
#pragma unroll
            for (int sl = 0; sl < H1 / 32; ++sl) {
                const float* src = params + O_W2 + (16 * w + j) * H1 + 32 * sl + 8 * kq;
                const f32x4 v0 = ld4(src), v1 = ld4(src + 4);
                const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    _Float16 p0, p1;
                    split2h(v[e], p0, p1);
                    wP[sl][0][e] = p0; wP[sl][1][e] = p1;
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < H1 / 16; ++s) wA[s] = ld4(params + O_W2 + (16 * w + j) * H1 + 16 * s + 4 * kq);
        }
        for (int t = 0; t < steps; ++t) {
            f32x4 xh1[1][2], xh2[1];
            float r1[1], r2[1];
            layer1<1, H3 ? 1 : 0>(L, w1a, w, j, kq, xh1, r1);
            lds_barrier();                                   // a1 visible
            if constexpr (H3) layer2_h3<1>(L, [&](int sl, int pc) { return wP[sl][pc]; }, w, j, kq, xh2, r2);
            else layer2<1>(L, [&](int s) { return wA[s]; }, w, j, kq, xh2, r2);
            lds_barrier();                                   // a2 visible
            heads_fwd<1>(L, wh, bh, w, lane);
            lds_barrier();                                   // wave 0 has stepped the environments: next observations visible
        }
    } else {
        // ------------------------------------------------------------------ wave 0: its rows from LDS + the env role
        for (int t = 0; t < steps; ++t) {
            const bool value_only = (t == T);
            f32x4 xh1[1][2], xh2[1];
            float r1[1], r2[1];
            layer1<1, H3 ? 1 : 0>(L, w1a, w, j, kq, xh1, r1);
            lds_barrier();                                       // a1 visible
            if constexpr (H3) layer2_h3<1>(L, [&](int sl, int pc) { return *reinterpret_cast<const f16x8*>(W2p + ((sl * 2 + pc) * 64 + lane) * 8); },
                                           w, j, kq, xh2, r2);
            else layer2<1>(L, [&](int s) { return ld4(w2row + 16 * s); }, w, j, kq, xh2, r2);
            lds_barrier();                                       // a2 visible
            heads_fwd<1>(L, wh, bh, w, lane);
            if (w == 0) {
                __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): HD written (single wave)
                __builtin_amdgcn_wave_barrier();
                if (lane < MT) {
                    float z[NA], p[NA];
#pragma unroll
                    for (int a = 0; a < NA; ++a) z[a] = L.HD[lane * 8 + a];
                    const float V = L.HD[lane * 8 + NA];
                    if (value_only) {
                        if (env_lane) B.last_val[my_env] = V;
                    } else {
                        // softmax + Categorical(probs) sample / log_prob (train_ppo2.0.py:161-163,189), as rollout.hip
                        float m = z[0];
#pragma unroll
                        for (int a = 1; a < NA; ++a) m = fmaxf(m, z[a]);
                        float ssum = 0.f;
#pragma unroll
                        for (int a = 0; a < NA; ++a) { p[a] = __expf(z[a] - m); ssum += p[a]; }
                        float psum = 0.f;
                        bool bad = false;
#pragma unroll
                        for (int a = 0; a < NA; ++a) { p[a] = p[a] / ssum; psum += p[a]; bad |= (p[a] != p[a]); }
                        if (bad && env_lane) atomicAdd(B.nan_count, 1);
                        const int eg = P.env_offset + my_env;
                        const size_t row = (size_t)min(my_env, N - 1) * T + t;
                        int a_sel;
                        if (B.forced_act) {
                            a_sel = B.forced_act[row];
                        } else {
                            const uint32_t u_act = philox4x32_10(P.seed, (uint32_t)t, (uint32_t)eg, (uint32_t)iter, RNG_ACTION).x;
                            const float target = u01_f32(u_act) * psum;
                            float cdf = 0.f;
                            a_sel = NA - 1;
                            bool found = false;
#pragma unroll
                            for (int a = 0; a < NA; ++a) {       // inverse CDF: first a with target < cdf
                                cdf += p[a];
                                if (!found && target < cdf) { a_sel = a; found = true; }
                            }
                        }
                        a_sel = a_sel < 0 ? 0 : (a_sel > NA - 1 ? NA - 1 : a_sel);
                        float psel = 0.f;
#pragma unroll
                        for (int a = 0; a < NA; ++a) if (a == a_sel) psel = p[a];
                        const float lp = __logf(fminf(fmaxf(psel / psum, R_EPS), 1.0f - R_EPS));
                        // environment step (f64, env_core.h) + auto reset
                        EnvState es = es_s[lane];
                        double z0, z1, wind_x, wind_y;
                        if (B.noise) { z0 = B.noise[2 * row]; z1 = B.noise[2 * row + 1]; }
                        else env_step_noise(P, eg, es, z0, z1);
                        env_step_wind(es, z0, z1, wind_x, wind_y);
                        StepOut so;
                        env_step_core(P, eg, es, myvis, a_sel, wind_x, wind_y, so);
                        if (env_lane) {
#pragma unroll
                            for (int f = 0; f < IN; ++f) B.obs[row * IN + f] = L.X[lane * 8 + f];
                            B.act[row] = a_sel;
                            B.rew[row] = (float)so.reward;
                            B.val[row] = V;
                            B.logp[row] = lp;
                            B.done[row] = so.done ? 1.f : 0.f;
                            B.flags[row] = (uint8_t)((so.done ? 1 : 0) | (so.reached ? 2 : 0));
                            if (B.heads) {
#pragma unroll
                                for (int a = 0; a < NA; ++a) B.heads[row * NH + a] = z[a];
                                B.heads[row * NH + NA] = V;
                            }
                            if (B.info) {
#pragma unroll
                                for (int f = 0; f < 5; ++f) B.info[row * 10 + f] = (float)so.info[f];
                                B.info[row * 10 + 5] = so.obs[2];
                                B.info[row * 10 + 6] = es.px;    // agent_pos after the move (before any auto-reset)
                                B.info[row * 10 + 7] = es.py;
                                B.info[row * 10 + 8] = (float)es.sx;
                                B.info[row * 10 + 9] = (float)es.sy;
                            }
                        }
                        if (so.done) {
                            es.episode += 1;
                            env_begin_episode(P, eg, es, myvis);
                            env_obs(P, es, myvis, so.obs);
                        }
#pragma unroll
                        for (int f = 0; f < IN; ++f) L.X[lane * 8 + f] = so.obs[f];
                        es_s[lane] = es;
                    }
                }
            }
            lds_barrier();                                       // next observations visible; HD / HP free again
        }
    }
    if (env_lane) {
        env_store(blob, my_env, es_s[lane]);
        for (int k = 0; k < NVIS; ++k) blob.visited[(size_t)my_env * NVIS + k] = myvis[k];
#pragma unroll
        for (int f = 0; f < IN; ++f) B.cur_obs[(size_t)my_env * IN + f] = L.X[lane * 8 + f];
    }
}

// ==================================================================================================== update
// LDS of the update kernel: the tiles + the per-unit accumulators of the LayerNorm / bias gradients
//   ACC1 [2][256] (dg1, dbe1)   ACC2 [3][128] (dg2, dbe2, db2)   DH [MS][8] dheads
// (each tile's contribution is summed over its 16 sample lanes by row16_sum first; db1 rides on the dW1 product as a
//  ones column of X.  Round 2 kept one slot per (sample lane, unit): 74 KB of LDS and 18 LDS read-modify-writes per tile.)
constexpr int UNC = 2;                  // column tiles per workgroup step: 32 samples
constexpr int UMS = MT * UNC;
constexpr int LS_STRIDE = 10;             // doubles per loss lane: policy / value / entropy / NaN sums + 6 head-bias sums
// + (H3) P2 [2 pieces][MS][RS2] dz2 as scaled fp16 planes, MX [8 waves][MS] per-sample maxima for their scale
constexpr int UPD_FLOATS = Tiles<UNC>::FLOATS + 2 * H1 + 3 * H2 + UMS * 8 + 2 * UMS * LS_STRIDE + H1 * 8 + 8 * H2 + UMS * RS2 + NWAVE * UMS + UMS * RS2;
constexpr size_t UPD_LDS = (size_t)UPD_FLOATS * sizeof(float);
constexpr int SLAB = NPARAM;            // one gradient slab per workgroup, flat parameter layout

// H3: the two K = 256 / K = 128 products (layer 2 forward, da1 = W2^T dz2) on the fp16 matrix pipe at f32 accuracy; `w2t`
// is then the pre-split weights in MFMA fragment order (mlp_w2_pieces_kernel) instead of the f32 transpose.  dW2 (K = the
// tile's 32 samples) stays on exact-f32 MFMA: its operands would need a second, transposed set of piece planes.
template <bool H3>
__global__ __launch_bounds__(512) void mlp_ppo_grad_kernel(
    const float* __restrict__ params, const float* __restrict__ obs, const int32_t* __restrict__ act,
    const float* __restrict__ logp_old, const float* __restrict__ adv, const float* __restrict__ ret,
    const float* __restrict__ val_old, int64_t Bn, float inv_n, float clip, float beta, double* __restrict__ loss_partial,
    float* __restrict__ slabs, const float* __restrict__ w2t) {
    constexpr int NC = UNC, MS = UMS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Tiles<NC> L(smem);
    float* ACC1 = smem + Tiles<NC>::FLOATS;
    float* ACC2 = ACC1 + 2 * H1;
    float* DH = ACC2 + 3 * H2;
    // the loss lanes' running sums live in LDS, not in registers: only wave 0 touches them, once per tile, but as registers
    // they were live in EVERY wave across the whole tile loop (14 VGPRs of a kernel that spills)
    double* LS = reinterpret_cast<double*>(DH + MS * 8);
    // W1 [256][8] (columns 6, 7 zero) and the head weights [8][128] (rows 6, 7 zero): their per-lane MFMA fragments are
    // re-read from here where they are used instead of living in 10 VGPRs across the whole tile loop
    float* W1L = reinterpret_cast<float*>(LS + MS * LS_STRIDE);
    float* WHL = W1L + H1 * 8;
    unsigned short* P2 = reinterpret_cast<unsigned short*>(WHL + 8 * H2);
    float* MX = reinterpret_cast<float*>(P2 + 2 * MS * RS2);
    // dz2 once more for dW2 = dz2^T a1 (K = the tile's samples, so the scale must not depend on the sample): plane 0 =
    // fp16(x), plane 1 = fp16(x - plane 0) with x = dz2 * 2^e_run, e_run the WAVE's running power of two (below)
    unsigned short* P2w = reinterpret_cast<unsigned short*>(MX + NWAVE * MS);

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, kq = lane >> 4;

    // ---- parameters: LayerNorm / bias vectors in LDS, the small weight fragments in registers
    for (int i = threadIdx.x; i < H1; i += 512) {
        L.prm[i] = params[O_B1 + i];
        L.prm[H1 + i] = params[O_G1 + i];
        L.prm[2 * H1 + i] = params[O_BE1 + i];
    }
    for (int i = threadIdx.x; i < H2; i += 512) {
        L.prm[3 * H1 + i] = params[O_B2 + i];
        L.prm[3 * H1 + H2 + i] = params[O_G2 + i];
        L.prm[3 * H1 + 2 * H2 + i] = params[O_BE2 + i];
    }
    for (int i = threadIdx.x; i < 2 * H1 + 3 * H2; i += 512) ACC1[i] = 0.f;
    for (int i = threadIdx.x; i < H1 * 8; i += 512) W1L[i] = (i & 7) < IN ? params[O_W1 + (i >> 3) * IN + (i & 7)] : 0.f;
    for (int i = threadIdx.x; i < 8 * H2; i += 512) WHL[i] = (i / H2) < NH ? params[O_WH + i] : 0.f;
    // W2 is NOT held in registers: the 64 + 64 VGPRs of its two orientations beside the 64 of the dW2 accumulators spill
    // (measured: 85 VGPRs to scratch).  Both orientations are streamed from L2 instead -- 256 KB per 32 samples and CU,
    // dwordx4 per lane thanks to the permuted k order -- W2 row-major for the forward, its transpose (w2t, made once per
    // call) for da1 = W2^T dz2.
    const float* wfw = params + O_W2 + (16 * w + j) * H1 + 4 * kq;           // forward:  W2[16 w + j][16 s + 4 kq ..]
    const float* wbw0 = w2t + (32 * w + j) * H2 + 4 * kq;                    // backward: W2^T[32 w + 16 t + j][16 s + 4 kq ..]
    const float* wbw1 = wbw0 + 16 * H2;
    // H3: fragment-ordered pieces.  forward: [row tile w][8 slabs][2 pieces][64 lanes][8]; backward (after the 64 K halves of
    // the forward set): [row tile 2 w + t][4 slabs][2][64][8]
    const unsigned short* w2f = reinterpret_cast<const unsigned short*>(w2t) + (size_t)w * 8 * 2 * 512 + lane * 8;
    const unsigned short* w2b = reinterpret_cast<const unsigned short*>(w2t) + (size_t)H1 * H2 * 2 + (size_t)(2 * w) * 4 * 2 * 512 + lane * 8;
    const float* bh = params + O_BH;

    f32x4 dW2[16], dW1[2], dWh;
#pragma unroll
    for (int i = 0; i < 16; ++i) dW2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    dW1[0] = dW1[1] = dWh = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < MS * LS_STRIDE; i += 512) LS[i] = 0.0;

    // H3: the dW2 accumulators of this wave hold dW2 * 2^e_run: dz2 rows of this wave's 16 units enter the fp16 product
    // scaled by a power of two that puts the largest magnitude seen SO FAR in [2^13, 2^14); when a tile brings a larger one
    // the exponent drops and the accumulators are rescaled (exact)
    int e_run = 100;
    const int64_t ntile = (Bn + MS - 1) / MS;
    M_PROF_DECL;
    for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int64_t s0 = tile * MS;
        // ---- stage the tile's observations; the loss lanes fetch their per-sample scalars early
        if (threadIdx.x < MS * 8) {
            const int sj = threadIdx.x >> 3, f = threadIdx.x & 7;
            // column 6 = 1: W1 has no column 6 (the forward multiplies it by zero), and dW1 = dz1^T X then carries
            // db1 = dz1^T 1 in its seventh column for free
            L.X[threadIdx.x] = (s0 + sj < Bn) ? (f < IN ? obs[(s0 + sj) * IN + f] : (f == IN ? 1.f : 0.f)) : 0.f;
        }
        int a_s = 0;
        float lpo = 0.f, Ad = 0.f, Rt = 0.f, vo = 0.f;
        const bool loss_lane = (w == 0) && lane < MS && s0 + lane < Bn;
        if (loss_lane) {
            a_s = act[s0 + lane]; lpo = logp_old[s0 + lane]; Ad = adv[s0 + lane]; Rt = ret[s0 + lane]; vo = val_old[s0 + lane];
        }
        lds_barrier();
        M_PROF_MARK(0);
        // ---- forward
        f32x4 xh1[NC][2], xh2[NC];
        float r1[NC], r2[NC];
        {
            float w1a[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) w1a[t][s] = W1L[(32 * w + 16 * t + j) * 8 + 4 * s + kq];
            layer1<NC, H3 ? 1 : 0>(L, w1a, w, j, kq, xh1, r1);
        }
        lds_barrier();
        M_PROF_MARK(1);
        if constexpr (H3) {
            // ONE address register pair, advanced 4 KB per two-slab chunk, the four fragments of a chunk at immediate offsets:
            // sixteen separate 64-bit addresses (what the compiler makes of w2f + constant) are sixteen spilled pairs
            const unsigned short* pw = w2f;
            asm volatile("" : "+v"(pw));
            layer2_h3<NC>(L, [&](int sl, int pc) {
                if (sl > 0 && (sl & 1) == 0 && pc == 0) { pw += 2048; asm volatile("" : "+v"(pw)); }
                return *reinterpret_cast<const f16x8*>(pw + ((sl & 1) * 2 + pc) * 512);
            }, w, j, kq, xh2, r2);
        } else layer2<NC>(L, [&](int s) { return ld4(wfw + 16 * s); }, w, j, kq, xh2, r2);
        lds_barrier();
        M_PROF_MARK(2);
        {
            float wh[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) wh[s] = j < 8 ? WHL[j * H2 + 16 * w + 4 * s + kq] : 0.f;
            heads_fwd<NC>(L, wh, bh, w, lane);
        }
        M_PROF_MARK(3);
        // ---- loss + d(total)/d(heads) for the tile's samples (lanes 0..MS-1 of wave 0)
        if (w == 0) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            if (lane < MS) {
                float dz[NA], dV = 0.f;
#pragma unroll
                for (int a = 0; a < NA; ++a) dz[a] = 0.f;
                if (loss_lane) {
                    float z[NA];
#pragma unroll
                    for (int a = 0; a < NA; ++a) z[a] = L.HD[lane * 8 + a];
                    double* ls = LS + lane * LS_STRIDE;
                    LossAcc lacc{ls[0], ls[1], ls[2], ls[3]};
                    dV = ppo_sample<NA>(z, L.HD[lane * 8 + NA], a_s, lpo, Ad, Rt, vo, inv_n, clip, beta, lacc, dz);
                    ls[0] = lacc.pl; ls[1] = lacc.vl; ls[2] = lacc.en; ls[3] = lacc.nan;
                    // head-bias gradient sums: f32 running sums as before (kept as f32 VALUES in the f64 slots)
#pragma unroll
                    for (int a = 0; a < NA; ++a) ls[4 + a] = (double)((float)ls[4 + a] + dz[a]);
                    ls[4 + NA] = (double)((float)ls[4 + NA] + dV);
                }
#pragma unroll
                for (int a = 0; a < NA; ++a) DH[lane * 8 + a] = dz[a];
                DH[lane * 8 + NA] = dV;
                DH[lane * 8 + 6] = 0.f;
                DH[lane * 8 + 7] = 0.f;
            }
        }
        lds_barrier();
        M_PROF_MARK(4);
        // ---- heads backward: dWh += dheads^T a2 (K = the tile's samples), da2 = Wh^T dheads
#pragma unroll
        for (int s = 0; s < MS / 4; ++s) {
            const float a = j < 8 ? DH[(4 * s + kq) * 8 + j] : 0.f;
            dWh = mfma4(a, L.A2[(4 * s + kq) * AS2 + 16 * w + j], dWh);
        }
        // ---- LayerNorm 2 + ReLU backward (mlp.hip: ln_relu_bwd_kernel's arithmetic)
        f32x4 dz2[NC];
        float isc2[NC];                                      // H3: inverse fp16 scale of samples 16 c + j
#pragma unroll
        for (int c = 0; c < NC; ++c) isc2[c] = 1.f;
        {
            const int u = 16 * w + 4 * kq;
            const f32x4 g = ld4(L.prm + 3 * H1 + H2 + u), be = ld4(L.prm + 3 * H1 + 2 * H2 + u);
            f32x4 dxh[NC];
            // per-lane partial sums in f32 (4 terms, fixed order), widened to f64 only for the cross-lane / cross-wave sum:
            // the reference's own LayerNorm backward accumulates in f32; 48 f64 conversions and FMAs per tile and lane
            // here and in LayerNorm 1 were a quarter of these phases' issue time
            float p1[NC], p2[NC];
            f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sb = sg, sz = sg;
            float* a2p = ACC2 + u;
            const float whT[2] = {WHL[kq * H2 + 16 * w + j], WHL[(4 + kq) * H2 + 16 * w + j]};
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                f32x4 da2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) da2 = mfma4(whT[s], DH[(16 * c + j) * 8 + 4 * s + kq], da2);
                float q1 = 0.f, q2 = 0.f, mx = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dy = (xh2[c][r] * g[r] + be[r] > 0.f) ? da2[r] : 0.f;
                    sg[r] += dy * xh2[c][r];
                    sb[r] += dy;
                    dxh[c][r] = dy * g[r];
                    q1 += dxh[c][r];
                    q2 += dxh[c][r] * xh2[c][r];
                    mx = fmaxf(mx, fabsf(dxh[c][r]));
                }
                p1[c] = q1;
                p2[c] = q2;
                if (H3) {        // this wave's max |dxhat| of sample 16 c + j, for the sample's fp16 scale (below)
                    {
                        const auto s16 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, mx), __builtin_bit_cast(unsigned, mx), false, false);
                        mx = fmaxf(__builtin_bit_cast(float, (unsigned)s16[0]), __builtin_bit_cast(float, (unsigned)s16[1]));
                        const auto s32 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, mx), __builtin_bit_cast(unsigned, mx), false, false);
                        mx = fmaxf(__builtin_bit_cast(float, (unsigned)s32[0]), __builtin_bit_cast(float, (unsigned)s32[1]));
                    }
                    if (kq == 0) MX[w * MS + 16 * c + j] = mx;
                }
            }
            sg = row16_sum(sg);                              // dg2, dbe2: summed over the tile's 16 sample lanes, then ONE lane per
            sb = row16_sum(sb);                              // kq group adds them to the per-unit accumulators
            if (j == 0) {
                st4(a2p, ld4(a2p) + sg);
                st4(a2p + H2, ld4(a2p + H2) + sb);
            }
            float S1[NC], S2[NC];
            ln_exchange_f32<NC>(L, w, j, kq, p1, p2, S1, S2);    // barrier inside: every wave's dWh reads of A2 are done
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float m1 = S1[c] * (1.0f / H2), m2 = S2[c] * (1.0f / H2);
#pragma unroll
                for (int r = 0; r < 4; ++r) dz2[c][r] = r2[c] * (dxh[c][r] - m1 - xh2[c][r] * m2);
                if (!H3) st4(L.A2 + (16 * c + j) * AS2 + u, dz2[c]);  // dz2 replaces a2 (the f32 dW2 product reads it)
                sz = sz + dz2[c];
                if (H3) {
                    // The sample's dz2 row as two fp16 planes for da1 = W2^T dz2.  Gradients span dozens of binades, fp16
                    // five bits of exponent: the row is scaled by the power of two that puts a BOUND on its largest
                    // magnitude -- |dz2| <= rstd (max |dxhat| + |m1| + sqrt(127) |m2|), identical in every wave -- into
                    // [2^13, 2^14), and the sample's column of da1 is scaled back exactly.
                    float M = MX[16 * c + j];
#pragma unroll
                    for (int i = 1; i < NWAVE; ++i) M = fmaxf(M, MX[i * MS + 16 * c + j]);
                    const float bound = r2[c] * (M + fabsf(m1) + 11.3f * fabsf(m2));
                    int e = 0;
                    if (bound > 0.f && bound < 3.0e38f) {
                        e = 13 - (int)((__float_as_uint(bound) >> 23) & 0xff) + 127;
                        e = e > 100 ? 100 : (e < -100 ? -100 : e);
                    }
                    const float sc = __uint_as_float((unsigned)(127 + e) << 23);
                    isc2[c] = __uint_as_float((unsigned)(127 - e) << 23);
                    unsigned short b[2][4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        _Float16 p0, p1;
                        split2h(dz2[c][r] * sc, p0, p1);
                        b[0][r] = h_bits(p0); b[1][r] = h_bits(p1);
                    }
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        uint2 v;
                        v.x = (unsigned)b[pc][0] | ((unsigned)b[pc][1] << 16);
                        v.y = (unsigned)b[pc][2] | ((unsigned)b[pc][3] << 16);
                        *reinterpret_cast<uint2*>(P2 + (pc * MS + 16 * c + j) * RS2 + u) = v;
                    }
                }
            }
            sz = row16_sum(sz);
            if (j == 0) st4(a2p + 2 * H2, ld4(a2p + 2 * H2) + sz);
            if constexpr (H3) {
                float m = 0.f;
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(dz2[c][r]));
                m = wave_max(m);
                int e = 100;
                if (m > 0.f && m < 3.0e38f) {
                    e = 13 - (int)((__float_as_uint(m) >> 23) & 0xff) + 127;
                    e = e > 100 ? 100 : (e < -100 ? -100 : e);
                }
                if (e < e_run) {                             // wave-uniform: a larger magnitude than any tile before
                    if (e_run < 100) {
#pragma unroll
                        for (int tj = 0; tj < 16; ++tj)
#pragma unroll
                            for (int r = 0; r < 4; ++r) dW2[tj][r] = __builtin_amdgcn_ldexpf(dW2[tj][r], e - e_run);
                    }
                    e_run = e;
                }
                const float sc = __builtin_amdgcn_ldexpf(1.0f, e_run);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    unsigned short b[2][4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x = dz2[c][r] * sc;
                        const _Float16 p0 = (_Float16)x;
                        const _Float16 p1 = (_Float16)(x - (float)p0);
                        b[0][r] = h_bits(p0); b[1][r] = h_bits(p1);
                    }
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        uint2 v;
                        v.x = (unsigned)b[pc][0] | ((unsigned)b[pc][1] << 16);
                        v.y = (unsigned)b[pc][2] | ((unsigned)b[pc][3] << 16);
                        *reinterpret_cast<uint2*>(P2w + (pc * MS + 16 * c + j) * RS2 + u) = v;
                    }
                }
            }
        }
        lds_barrier();                                       // dz2 visible
        M_PROF_MARK(5);
        // ---- dW2 += dz2^T a1 (row tile w, all 16 column tiles; K = the tile's samples), da1 = W2^T dz2
        if constexpr (H3) {
            // one K = 32 slab (the tile's samples): A = this wave's 16 dz2 rows from P2w, B = a1 from the SAME planes layer 2
            // read row-wise, both through the transposing LDS read.  a b = A0 q0 + 2^-11 A0 q1 + r q0 (+ O(2^-22)) with
            // A0 = fp16(x), r = fp16(x - A0), q0 / q1 = split2h(a1): ONE accumulator, the 2^-11 folded into an operand.
            const f16x8 a0 = tr_frag(P2w + 16 * w, RS2, lane);
            const f16x8 ar = tr_frag(P2w + MS * RS2 + 16 * w, RS2, lane);
            const f16x8 a0s = a0 * (_Float16)(1.0f / 2048.0f);
#pragma unroll
            for (int tj = 0; tj < 16; ++tj) {
                const f16x8 q0 = tr_frag(L.P1 + 16 * tj, RS1, lane);
                const f16x8 q1 = tr_frag(L.P1 + MS * RS1 + 16 * tj, RS1, lane);
                dW2[tj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, q0, dW2[tj], 0, 0, 0);
                dW2[tj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0s, q1, dW2[tj], 0, 0, 0);
                dW2[tj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ar, q0, dW2[tj], 0, 0, 0);
                if ((tj & 3) == 3) asm volatile("" ::: "memory");
            }
        } else {
            float af[MS / 4];
#pragma unroll
            for (int s = 0; s < MS / 4; ++s) af[s] = L.A2[(4 * s + kq) * AS2 + 16 * w + j];
#pragma unroll
            for (int tj = 0; tj < 16; ++tj) {
#pragma unroll
                for (int s = 0; s < MS / 4; ++s) dW2[tj] = mfma4(af[s], L.A1[(4 * s + kq) * AS1 + 16 * tj + j], dW2[tj]);
                if ((tj & 3) == 3) asm volatile("" ::: "memory");      // four column tiles' fragments at a time
            }
        }
        M_PROF_MARK(6);
        f32x4 da1[NC][2];
#pragma unroll
        for (int c = 0; c < NC; ++c) da1[c][0] = da1[c][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (H3) {
            // da1^T tile (row tiles 2 w, 2 w + 1 of the 256 units) = W2^T pieces x the scaled dz2 planes, K = 128 in 4 slabs
            const unsigned short* brow = P2 + j * RS2 + 8 * kq;
            f32x4 acl[NC][2];
#pragma unroll
            for (int c = 0; c < NC; ++c) acl[c][0] = acl[c][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            f16x8 wq[2][2][2];                               // [buffer][row tile][piece]: one slab ahead
            const unsigned short* pb0 = w2b;                 // one address pair per row tile, advanced 2 KB per slab
            const unsigned short* pb1 = w2b + 4 * 1024;
            asm volatile("" : "+v"(pb0), "+v"(pb1));
            wq[0][0][0] = *reinterpret_cast<const f16x8*>(pb0); wq[0][0][1] = *reinterpret_cast<const f16x8*>(pb0 + 512);
            wq[0][1][0] = *reinterpret_cast<const f16x8*>(pb1); wq[0][1][1] = *reinterpret_cast<const f16x8*>(pb1 + 512);
#pragma unroll
            for (int sl = 0; sl < H2 / 32; ++sl) {
                if (sl + 1 < H2 / 32) {
                    pb0 += 1024; pb1 += 1024;
                    asm volatile("" : "+v"(pb0), "+v"(pb1));
                    wq[(sl + 1) & 1][0][0] = *reinterpret_cast<const f16x8*>(pb0); wq[(sl + 1) & 1][0][1] = *reinterpret_cast<const f16x8*>(pb0 + 512);
                    wq[(sl + 1) & 1][1][0] = *reinterpret_cast<const f16x8*>(pb1); wq[(sl + 1) & 1][1][1] = *reinterpret_cast<const f16x8*>(pb1 + 512);
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const f16x8 b0 = *reinterpret_cast<const f16x8*>(brow + (16 * c) * RS2 + 32 * sl);
                    const f16x8 b1 = *reinterpret_cast<const f16x8*>(brow + (MS + 16 * c) * RS2 + 32 * sl);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        acl[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[sl & 1][t][1], b0, acl[c][t], 0, 0, 0);
                        da1[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[sl & 1][t][0], b0, da1[c][t], 0, 0, 0);
                        acl[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[sl & 1][t][0], b1, acl[c][t], 0, 0, 0);
                    }
                }
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int t = 0; t < 2; ++t) da1[c][t] = (da1[c][t] + acl[c][t] * H3_LO) * isc2[c];
        } else {
            const float* brow = L.A2 + j * AS2 + 4 * kq;
            f32x4 wq[2][2][2];                               // [buffer][slab in chunk][row tile]: two slabs ahead
#pragma unroll
            for (int i = 0; i < 2; ++i) { wq[0][i][0] = ld4(wbw0 + 16 * i); wq[0][i][1] = ld4(wbw1 + 16 * i); }
#pragma unroll
            for (int ch = 0; ch < H2 / 32; ++ch) {
                if (ch + 1 < H2 / 32) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        wq[(ch + 1) & 1][i][0] = ld4(wbw0 + 16 * (2 * (ch + 1) + i));
                        wq[(ch + 1) & 1][i][1] = ld4(wbw1 + 16 * (2 * (ch + 1) + i));
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f32x4 b[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c) b[c] = ld4(brow + 16 * c * AS2 + 16 * (2 * ch + i));
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            da1[c][0] = mfma4(wq[ch & 1][i][0][k], b[c][k], da1[c][0]);
                            da1[c][1] = mfma4(wq[ch & 1][i][1][k], b[c][k], da1[c][1]);
                        }
                }
                asm volatile("" ::: "memory");
            }
        }
        M_PROF_MARK(7);
        // ---- LayerNorm 1 + ReLU backward
        {
            f32x4 dxh[NC][2];                                // da1 becomes dxhat in place
            float q1[NC], q2[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) q1[c] = q2[c] = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int u = 32 * w + 16 * t + 4 * kq;
                const f32x4 g = ld4(L.prm + H1 + u), be = ld4(L.prm + 2 * H1 + u);
                f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sb = sg;
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dy = (xh1[c][t][r] * g[r] + be[r] > 0.f) ? da1[c][t][r] : 0.f;
                        sg[r] += dy * xh1[c][t][r];
                        sb[r] += dy;
                        dxh[c][t][r] = dy * g[r];
                        q1[c] += dxh[c][t][r];
                        q2[c] += dxh[c][t][r] * xh1[c][t][r];
                    }
                sg = row16_sum(sg);
                sb = row16_sum(sb);
                if (j == 0) {
                    float* a1p = ACC1 + u;                   // dg1, dbe1
                    st4(a1p, ld4(a1p) + sg);
                    st4(a1p + H1, ld4(a1p + H1) + sb);
                }
            }
            float S1[NC], S2[NC];
            ln_exchange_f32<NC>(L, w, j, kq, q1, q2, S1, S2);    // barrier inside: every wave's dW2 reads of A1 are done
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int u = 32 * w + 16 * t + 4 * kq;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float m1 = S1[c] * (1.0f / H1), m2 = S2[c] * (1.0f / H1);
                    f32x4 dz1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dz1[r] = r1[c] * (dxh[c][t][r] - m1 - xh1[c][t][r] * m2);
                    st4(L.A1 + (16 * c + j) * AS1 + u, dz1);  // dz1 replaces a1 (db1 = its column sums: the dW1 product below)
                }
            }
        }
        lds_barrier();                                       // dz1 visible
        M_PROF_MARK(8);
        // ---- dW1 += dz1^T x
#pragma unroll
        for (int s = 0; s < MS / 4; ++s) {
            const float b = j < 8 ? L.X[(4 * s + kq) * 8 + j] : 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t) dW1[t] = mfma4(L.A1[(4 * s + kq) * AS1 + 32 * w + 16 * t + j], b, dW1[t]);
        }
        lds_barrier();                                       // X, A1, A2, DH free for the next tile
        M_PROF_MARK(9);
    }
    M_PROF_FLUSH();

    // ---- this workgroup's gradient slab (flat parameter layout) and loss partial
    float* slab = slabs + (size_t)blockIdx.x * SLAB;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
            if (j < IN) slab[O_W1 + (32 * w + 16 * t + 4 * kq + r) * IN + j] = dW1[t][r];
            else if (j == IN) slab[O_B1 + 32 * w + 16 * t + 4 * kq + r] = dW1[t][r];       // the ones column: db1
        }
#pragma unroll
    for (int tj = 0; tj < 16; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            slab[O_W2 + (16 * w + 4 * kq + r) * H1 + 16 * tj + j] = H3 ? __builtin_amdgcn_ldexpf(dW2[tj][r], -e_run) : dW2[tj][r];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (4 * kq + r < NH) slab[O_WH + (4 * kq + r) * H2 + 16 * w + j] = dWh[r];
    // LayerNorm / bias gradients from the per-unit accumulators
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * H1; i += 512) slab[(i < H1 ? O_G1 : O_BE1) + (i % H1)] = ACC1[i];
    for (int i = threadIdx.x; i < 3 * H2; i += 512) {
        const int q = i / H2, u = i % H2;
        slab[(q == 0 ? O_G2 : (q == 1 ? O_BE2 : O_B2)) + u] = ACC2[i];
    }
    if (w == 0) {
        // loss sums and the head-bias gradient: the loss lanes park their partial sums, lane 0 adds them in lane order
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            double* pp = loss_partial + (size_t)LOSS_PSTRIDE * blockIdx.x;
            for (int k = 0; k < 4 + NH; ++k) {
                double t = 0.0;
                for (int sj = 0; sj < MS; ++sj) t += LS[sj * LS_STRIDE + k];
                pp[k] = t;
                if (k >= 4) slab[O_BH + k - 4] = (float)t;
            }
        }
    }
}

// w2t[u1][u2] = W2[u2][u1]  (128 KB; once per gradient call)
__global__ __launch_bounds__(256) void mlp_w2_transpose_kernel(const float* __restrict__ params, float* __restrict__ w2t) {
    const int i = blockIdx.x * 256 + threadIdx.x;           // i = u1 * H2 + u2
    if (i < H1 * H2) w2t[i] = params[O_W2 + (i % H2) * H1 + i / H2];
}

// H3: both orientations of W2 as fp16 piece fragments (once per gradient call): forward set [8 row tiles][8 slabs][2][64][8],
// element (i, kq, e) of fragment (w, s) = W2[16 w + i][32 s + 8 kq + e]; backward set [16 row tiles][4 slabs][2][64][8],
// element = W2[32 s + 8 kq + e][16 rt + i].
__global__ __launch_bounds__(256) void mlp_w2_pieces_kernel(const float* __restrict__ params, unsigned short* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;           // over 2 x H1 * H2 elements
    if (i >= 2 * H1 * H2) return;
    const bool bwd = i >= H1 * H2;
    const int q = bwd ? i - H1 * H2 : i;
    const int e = q & 7, lane = (q >> 3) & 63, row = lane & 15, kq = lane >> 4;
    float v;
    size_t o;
    if (!bwd) {
        const int sl = (q >> 9) & 7, wt = q >> 12;
        v = params[O_W2 + (16 * wt + row) * H1 + 32 * sl + 8 * kq + e];
        o = ((size_t)(wt * 8 + sl) * 2) * 512 + lane * 8 + e;
    } else {
        const int sl = (q >> 9) & 3, rt = q >> 11;
        v = params[O_W2 + (32 * sl + 8 * kq + e) * H1 + 16 * rt + row];
        o = (size_t)H1 * H2 * 2 + ((size_t)(rt * 4 + sl) * 2) * 512 + lane * 8 + e;
    }
    _Float16 p0, p1;
    split2h(v, p0, p1);
    out[o] = h_bits(p0);
    out[o + 512] = h_bits(p1);
}

// grad[i] = sum over the workgroups' slabs, 8 independent partial sums in a fixed association (deterministic)
__global__ __launch_bounds__(256) void mlp_slab_reduce_kernel(const float* __restrict__ slabs, int nb, int n, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    float p8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 8 <= nb; b += 8)
#pragma unroll
        for (int k = 0; k < 8; ++k) p8[k] += slabs[(size_t)(b + k) * n + c];
    for (; b < nb; ++b) p8[0] += slabs[(size_t)b * n + c];
    out[c] = ((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]));
}

}  // namespace

// ---- entry used by uav_rollout (rollout.hip) for policy_kind 0
int launch_rollout_mlp(uav_ctx* ctx, void* env_state, int n_env, const uav_env_cfg* cfg, const float* params, int horizon,
                       uint64_t iter, float* cur_obs, float* obs, int32_t* act, float* rew, float* val, float* logp,
                       float* done, uint8_t* flags, float* last_val, const int32_t* forced_act, const double* noise,
                       int32_t* nan_count, float* info, float* heads, hipStream_t st) {
    EnvParams P;
    int rc = env_params_from_cfg(ctx, cfg, n_env, P);
    if (rc) return rc;
    UAV_REQUIRE(P.trend_k == 0, "uav_rollout: the fused MLP rollout has 6 observation features (trend_k = 0)");
    MlpRollBufs B{cur_obs, obs, act, rew, val, logp, done, flags, last_val, forced_act, noise, nan_count, info, heads};
    EnvBlob blob = env_blob_view(env_state, n_env);
    // the handle's arithmetic (uav_set_lstm_arith): FP16X3 = the 256 x 128 layer on the fp16 matrix pipe (max |param| < 2048);
    // anything else = exact-f32 MFMA
    if (ctx->lstm_arith == UAV_ARITH_FP16X3) {
        UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(&rollout_mlp_kernel<true>), (int)ROLL_LDS));
        hipLaunchKernelGGL(rollout_mlp_kernel<true>, dim3((n_env + MT - 1) / MT), dim3(512), ROLL_LDS, st, P, blob, n_env, horizon,
                           iter, params, B);
    } else {
        UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(&rollout_mlp_kernel<false>), (int)ROLL_LDS));
        hipLaunchKernelGGL(rollout_mlp_kernel<false>, dim3((n_env + MT - 1) / MT), dim3(512), ROLL_LDS, st, P, blob, n_env, horizon,
                           iter, params, B);
    }
    UAV_LAUNCH_CHECK();
    return 0;
}

extern "C" int uav_mlp_ppo_grad(uav_ctx* ctx, const float* params, const float* obs, const int32_t* act,
                                const float* logp_old, const float* adv, const float* ret, const float* val_old, int64_t n,
                                int in_dim, int h1, int h2, int n_act, float inv_n, float clip, float ent_beta,
                                double* loss_sums, float* grad, uav_stream stream) {
    UAV_REQUIRE(ctx && params && obs && act && logp_old && adv && ret && val_old && loss_sums && grad && n > 0,
                "uav_mlp_ppo_grad: bad argument");
    UAV_REQUIRE(in_dim == IN && h1 == H1 && h2 == H2 && n_act == NA,
                "uav_mlp_ppo_grad: the fused kernel is the reference's network (6-256-128, 5 actions); got %d-%d-%d, %d",
                in_dim, h1, h2, n_act);
    hipStream_t st = as_stream(stream);
    const int64_t ntile = (n + UMS - 1) / UMS;
    int nb = ctx->num_cu;
    if (nb > ntile) nb = (int)ntile;
    const bool h3 = ctx->lstm_arith == UAV_ARITH_FP16X3;       // the handle's arithmetic (uav_set_lstm_arith); else exact-f32 MFMA
    const size_t head = 65536, w2t_bytes = (size_t)H1 * H2 * sizeof(float) * 2;  // loss partials | W2^T or both piece sets | slabs
    UAV_REQUIRE(ctx->ws_bytes >= head + w2t_bytes + (size_t)nb * SLAB * sizeof(float), "uav_mlp_ppo_grad: workspace too small");
    double* partial = (double*)ctx->ws;
    float* w2t = (float*)((char*)ctx->ws + head);
    float* slabs = (float*)((char*)ctx->ws + head + w2t_bytes);
    if (h3) {
        hipLaunchKernelGGL(mlp_w2_pieces_kernel, dim3(2 * H1 * H2 / 256), dim3(256), 0, st, params, reinterpret_cast<unsigned short*>(w2t));
        UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(&mlp_ppo_grad_kernel<true>), (int)UPD_LDS));
        hipLaunchKernelGGL(mlp_ppo_grad_kernel<true>, dim3(nb), dim3(512), UPD_LDS, st, params, obs, act, logp_old, adv, ret, val_old,
                           n, inv_n, clip, ent_beta, partial, slabs, w2t);
    } else {
        hipLaunchKernelGGL(mlp_w2_transpose_kernel, dim3(H1 * H2 / 256), dim3(256), 0, st, params, w2t);
        UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(&mlp_ppo_grad_kernel<false>), (int)UPD_LDS));
        hipLaunchKernelGGL(mlp_ppo_grad_kernel<false>, dim3(nb), dim3(512), UPD_LDS, st, params, obs, act, logp_old, adv, ret, val_old,
                           n, inv_n, clip, ent_beta, partial, slabs, w2t);
    }
    hipLaunchKernelGGL(mlp_slab_reduce_kernel, dim3((SLAB + 255) / 256), dim3(256), 0, st, slabs, nb, SLAB, grad);
    UAV_LAUNCH_CHECK();
    return launch_loss_final(partial, nb, NH, loss_sums, nullptr, st);
}

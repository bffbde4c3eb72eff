// lstm.hip -- L1: nn.LSTM-semantics sequence kernels (forward with BPTT stash, backward).
//
// Reference semantics: torch.nn.LSTM as the reference uses it (PPOV2.0/model.py:206-212,
// PPOV2.1/model.py:263,284,311,330): gates = x W_ih^T + b_ih + h W_hh^T + b_hh, chunk order
// i,f,g,o, c' = s(f) c + s(i) tanh(g), h' = s(o) tanh(c').
//
// MI355X design (persistent over time):
//   * one workgroup owns 16 env sequences for ALL T steps; grid = N/16 (256 WGs at N=4096);
//   * the recurrent weights stay on chip for the whole sequence (VGPRs, plus a wave-private LDS slab for the part
//     that does not fit): they are read from HBM once per launch, not once per step;
//   * the i,f,g,o pre-activations of one (env, unit) land in ONE lane's accumulators, so the gate pointwise and the
//     cell state never leave registers; h_t / the gate gradients are exchanged through LDS, one or two barriers a step.
// Kernel families in this file (dispatch: launch_fwd, lstm_bwd_seq):
//   DEFAULT  lstm_fwd_h3_kernel, lstm_bwd_h3k_kernel: the matrix products on the fp16 pipe at f32 accuracy (two-piece
//            operand split, three products, common.h split2h; gate gradients block-scaled per env), weights-as-A
//            orientation (a lane owns one env and four consecutive units: dwordx4 stores), stash by LDS-DMA in the backward;
//   BF16 SPLIT lstm_fwd_x6_kernel, lstm_bwd_x6k_kernel: the same kernels with a three-piece bf16 split and six
//            products (f32's exponent range: no operand limits, twice the matrix work); uav_set_lstm_arith(UAV_ARITH_BF16X6);
//   EXACT-F32 lstm_fwd_kernel, lstm_bwd_kernel (also the plain-dy backward of stacked layers): v_mfma_f32_16x16x4_f32
//            with 128 weight VGPRs per lane; uav_set_lstm_arith(UAV_ARITH_F32_MFMA).
//   (Earlier generations -- an output-split bf16 backward exchanging dG through three LDS planes, an LDS-DMA form of
//    the exact-f32 backward -- were measured, superseded and removed in round 3; their numbers are in profiles/README.md.)
//   (A half-step stagger of waves 4-7 against 0-3 was built and measured on the exact-f32 forward: no gain,
//    tools/stagger_probe.hip; what paced the step was the IEEE division sequence inside the activations, now v_rcp_f32.)
//   * the K=I<=8 input projection rides along as two exact-f32 MFMA k-steps; wider inputs (stacked layers) use a
//     time-batched GEMM into the stash first.
#include <stdlib.h>
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

int gemm_f32(uav_ctx* ctx, int64_t M, int64_t N, int64_t K, const float* A, int64_t sa_m, int64_t sa_k,
             const float* B, int64_t sb_k, int64_t sb_n, float* C, int64_t ldc, const float* bias,
             int accumulate, hipStream_t st);
int colsum(uav_ctx* ctx, const float* X, int64_t B, int C, float* out, float* scratch, hipStream_t st);
int colsum_absmax(uav_ctx* ctx, const float* X, int64_t B, int C, float* out, float* scratch, unsigned* absmax_bits,
                  hipStream_t st);
int colsum_xw(uav_ctx* ctx, const float* X, int64_t B, int C, const float* x, int I, float* colsum_out, float* xw_out,
              float* scratch, unsigned* absmax_bits, hipStream_t st, int xw_transposed = 0);
bool gemm_h3_ok(int64_t M, int64_t N, int64_t K, const float* A, int64_t sa_m, int64_t sa_k, const float* B, int64_t sb_k,
                int64_t sb_n);
int gemm_h3(uav_ctx* ctx, int64_t M, int64_t N, int64_t K, const float* A, int64_t sa_m, int64_t sa_k, const float* B,
            int64_t sb_k, int64_t sb_n, float* C, int64_t ldc, const float* bias, int accumulate, const unsigned* a_absmax,
            hipStream_t st);
int lstm_generic_fwd(uav_ctx* ctx, const float* keep, const float* h0, const float* c0, const float* w_hh, int N, int T,
                     int H, float* y, float* hn, float* cn, float* stash, hipStream_t st);
int lstm_h3_fwd(uav_ctx* ctx, const float* x, int I, const float* w_ih, const float* b_ih, const float* b_hh,
                const float* keep, const float* h0, const float* c0, const float* w_hh, int N, int T, float* y, float* hn,
                float* cn, float* stash, hipStream_t st);
bool lstm_h3_step_path(int H);
int lstm_generic_bwd(uav_ctx* ctx, const float* keep, const float* stash, const float* w_hh, const float* dy,
                     const float* dheads, const float* w_head, int n_heads,
                     const float* dhn, const float* dcn, int N, int T, int H, float* dgates, float* dh0, float* dc0,
                     const float* w_ih, int I, float* dx, hipStream_t st);
int lstm_generic_bwd_caps(int I, int H);
bool lstm_h3_stack_ok(int H);
bool lstm_h3_dg_packed(int H);
int lstm_pc_unpack(const void* dgates, int N, int T, float* out, hipStream_t st);
int lstm_pc_hprev_rows(const float* y, const float* keep, const float* h0, int N, int T, float* out, hipStream_t st);
int lstm_pc_wgrad(uav_ctx* ctx, const float* x, const float* y, const float* h0, const void* dgates, int N, int T, int I, float* dw_ih,
                  float* dw_hh, float* db, float* db_hh, hipStream_t st);
int lstm_h3_bwd_stack(uav_ctx* ctx, int nl, const uav_lstm_bwd_layer* layers, const float* dy, const float* dheads,
                      const float* w_head, int n_heads, int N, int T, hipStream_t st);
int lstm_wgrad_fused(uav_ctx* ctx, const float* dgates, const float* y_prev_src, const float* keep, const float* h0,
                     const float* x, int I, const float* ytop, const float* dheads, int NH, int N, int T, int H,
                     float* dw_ih, float* dw_hh, float* db, float* db_hh, float* dw_head, hipStream_t st);

constexpr int MT = 16;      // env rows per workgroup (MFMA M)
// UAV_LSTM_F32_MFMA=1 selects the exact v_mfma_f32_16x16x4_f32 kernels (the A/B reference of the split ones)
static bool f32_mfma_requested() { return uav_want_f32_mfma(); }   // the handle's mode (uav_set_lstm_arith)
static bool bf16x6_requested() { return uav_want_bf16x6(); }        // the predecessor of the fp16 split
constexpr int TC = 32;      // time steps staged per chunk

#define sigmoidf_ fast_sigmoid
#define tanhf_ fast_tanh

template <int H>
struct FwdGeom {
    static constexpr int NW = H / 16;          // waves per workgroup
    static constexpr int KS = H / 4;           // MFMA k-steps over the hidden dimension
    static constexpr int SEG = KS + 4;         // padded quarter-row (conflict-free ds_read_b128)
    static constexpr int S = 4 * SEG + 8;      // padded row stride of the h tile
    static constexpr size_t LDS = (2 * MT * S + TC * MT * 8 + (TC + 1) * MT) * sizeof(float);
};
// position of hidden unit u inside a padded h row
template <int H>
__device__ __forceinline__ int hpos(int u) { return (u / (H / 4)) * FwdGeom<H>::SEG + (u % (H / 4)); }

template <int H, bool FUSE_X>
__global__ __launch_bounds__(H * 4) void lstm_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ keep, const float* __restrict__ h0,
    const float* __restrict__ c0, const float* __restrict__ w_ih, const float* __restrict__ w_hh,
    const float* __restrict__ b_ih, const float* __restrict__ b_hh, int N, int T, int I,
    float* __restrict__ y, float* __restrict__ hn, float* __restrict__ cn, float* __restrict__ stash) {
    using G = FwdGeom<H>;
    constexpr int KS = G::KS, SEG = G::SEG, S = G::S;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* hbuf = smem;                        // [2][MT][S]
    float* xbuf = hbuf + 2 * MT * S;           // [TC][MT][8]
    float* kbuf = xbuf + TC * MT * 8;          // [TC+1][MT]

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int u = 16 * w + j;                  // hidden unit of this lane's accumulator column
    const int n0 = blockIdx.x * MT;

    // ---- this wave's slice of the weights, resident in VGPRs for the whole sequence
    float wh[4][KS];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float* src = w_hh + (size_t)(q * H + u) * H + kq * KS;
#pragma unroll
        for (int s = 0; s < KS; s += 4) {
            const float4 v = *reinterpret_cast<const float4*>(src + s);
            wh[q][s] = v.x; wh[q][s + 1] = v.y; wh[q][s + 2] = v.z; wh[q][s + 3] = v.w;
        }
    }
    float wx[4][2], bias[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bias[q] = FUSE_X ? (b_ih[q * H + u] + b_hh[q * H + u]) : 0.f;   // non-fused: bias is in `pre`
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int k = 2 * kq + s;
            wx[q][s] = (FUSE_X && k < I) ? w_ih[(size_t)(q * H + u) * I + k] : 0.f;
        }
    }

    // ---- incoming state of this lane's (env row r, unit u) pairs: rows e = 4*kq + r
    float c_reg[4], hin[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int e = 4 * kq + r;
        const int n = min(n0 + e, N - 1);
        const float k0 = keep ? keep[(size_t)n * T] : 1.f;
        c_reg[r] = c0[(size_t)n * H + u] * k0;
        hin[r] = h0[(size_t)n * H + u] * k0;
        hbuf[e * S + hpos<H>(u)] = hin[r];
    }
    int cur = 0;
    float h_last[4] = {0.f, 0.f, 0.f, 0.f};

    for (int t0 = 0; t0 < T; t0 += TC) {
        const int tc = min(TC, T - t0);
        // ---- stage x[n0..n0+15][t0..t0+tc) (zero-padded to 8 features) and keep[..][t0..t0+tc]
        if (FUSE_X) {
            for (int idx = threadIdx.x; idx < MT * tc * 8; idx += blockDim.x) {
                const int e = idx / (tc * 8), rem = idx % (tc * 8), tt = rem >> 3, f = rem & 7;
                const int n = min(n0 + e, N - 1);
                xbuf[(tt * MT + e) * 8 + f] = (f < I) ? x[((size_t)n * T + t0 + tt) * I + f] : 0.f;
            }
        }
        for (int idx = threadIdx.x; idx < MT * (tc + 1); idx += blockDim.x) {
            const int e = idx / (tc + 1), tt = idx % (tc + 1);
            const int n = min(n0 + e, N - 1);
            kbuf[tt * MT + e] = (keep && t0 + tt < T) ? keep[(size_t)n * T + t0 + tt] : 1.f;
        }
        lds_barrier();

        for (int tt = 0; tt < tc; ++tt) {
            const int t = t0 + tt;
            f32x4 acc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (FUSE_X) {
                    acc[q] = f32x4{bias[q], bias[q], bias[q], bias[q]};
                } else {
                    // pre-activations x W_ih^T + b from the time-batched GEMM (gates slot of the stash)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = min(n0 + 4 * kq + r, N - 1);
                        acc[q][r] = stash[((size_t)n * T + t) * (6 * H) + q * H + u];
                    }
                }
            }
            if (FUSE_X) {
                const float2 ax = *reinterpret_cast<const float2*>(&xbuf[(tt * MT + j) * 8 + 2 * kq]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax.x, wx[q][0], acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax.y, wx[q][1], acc[q], 0, 0, 0);
                }
            }
            // recurrent part: A = h_{t-1}[env j][k], k permuted so that lane (j,kq) reads a contiguous run
            const float* hrow = hbuf + cur * MT * S + j * S + kq * SEG;
#pragma unroll
            for (int s = 0; s < KS; s += 4) {
                const float4 a = *reinterpret_cast<const float4*>(hrow + s);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wh[q][s], acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wh[q][s + 1], acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wh[q][s + 2], acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wh[q][s + 3], acc[q], 0, 0, 0);
            }
            // ---- gate pointwise, all in this lane's registers (C/D map: row = 4*kq + r, col = j)
            float* hnext = hbuf + (cur ^ 1) * MT * S;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = 4 * kq + r;
                const int n = n0 + e;
                const float gi = sigmoidf_(acc[0][r]), gf = sigmoidf_(acc[1][r]);
                const float gg = tanhf_(acc[2][r]), go = sigmoidf_(acc[3][r]);
                const float cp = c_reg[r];
                const float c = gf * cp + gi * gg;
                const float h = go * tanhf_(c);
                const float kn = kbuf[(tt + 1) * MT + e];     // keep of step t+1 (1 past the end)
                if (n < N) {
                    const size_t row = (size_t)n * T + t;
                    y[row * H + u] = h;
                    if (stash) {
                        float* sp = stash + row * (6 * H);
                        sp[u] = gi; sp[H + u] = gf; sp[2 * H + u] = gg; sp[3 * H + u] = go;
                        sp[4 * H + u] = cp;
                        if (I > 6) sp[5 * H + u] = hin[r];   // h_prev: only the generic wgrad path (I > 6) reads it
                    }
                }
                h_last[r] = h;
                if (t == T - 1) c_reg[r] = c;                 // cn is the unmasked final cell state
                else c_reg[r] = c * kn;
                hin[r] = h * kn;
                hnext[e * S + hpos<H>(u)] = hin[r];
            }
            cur ^= 1;
            lds_barrier();
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + 4 * kq + r;
        if (n < N) {
            hn[(size_t)n * H + u] = h_last[r];
            cn[(size_t)n * H + u] = c_reg[r];
        }
    }
}

// ------------------------------------------------------------------------- forward, split-bf16 MFMA
// Same recurrence, but the h W_hh^T product runs on the bf16 matrix pipe at f32 accuracy: every f32 operand is
// split into three bf16 pieces (a = a0 + a1 + a2, 8 significand bits each, so the split is exact) and the six
// piece products a_i b_j with i + j <= 2 are accumulated in f32 by v_mfma_f32_16x16x32_bf16.  The dropped
// products are below 2^-24 |a b|, i.e. under the rounding of the f32 accumulation itself: measured error equals
// the exact-f32 MFMA chain's (tools/bf16x6_probe.hip: 2.69e-7 vs 2.75e-7 on a K=128 dot product) at 2.6x its
// rate (6 x 16 cycles per K=32 slab against 8 x 32).  W_hh pieces stay in VGPRs (192 at H=128); h_t is split
// once by the lane that produces it and exchanged through three bf16 LDS planes.


template <int H>
struct FwdX6Geom {
    static constexpr int NS = H / 32;          // K = 32 slabs over the hidden dimension
    static constexpr int RS = H + 8;           // bf16 elements per padded row: 16 lanes x ds_read_b128 conflict-free
    static constexpr int PLANE = MT * RS;
    // three bf16 pieces of W_hh are 1.5x its f32 size = 3/4 of the CU's register file at H=128: the smallest
    // piece of QL of the four gates lives in a wave-private LDS slab instead (read back as lane-contiguous b128)
    static constexpr int QL = (H >= 128) ? 3 : 0;
    static constexpr int WPARK = QL * NS * 64 * 8;                       // bf16 elements per wave
    static constexpr int TCX = 8;                                        // steps of x / keep staged per chunk
    static constexpr int XPT = (MT * TCX * 8 + H * 4 - 1) / (H * 4);     // staged x elements per thread
    static constexpr int HPL = 8 * RS;                                   // one piece plane of the (<= 8) head weight rows
    static constexpr size_t LDS = (2 * 3 * PLANE + (H / 16) * WPARK + 3 * HPL) * sizeof(unsigned short) +
                                  (2 * TCX * MT * 8 + 2 * TCX * MT + (H / 16) * 8 * 64 + 4 * H + 8) * sizeof(float);
};

#ifdef UAV_X6_PROFILE
// phase timing of the split-bf16 forward (instrumented build only, -DUAV_X6_PROFILE): s_memtime at four marks
// per step, summed over the sequence by wave 0 and the last wave of workgroup 0
__device__ unsigned long long g_x6_prof[2][8];
#define X6_PROF_DECL unsigned long long pm_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pl_ = __builtin_readcyclecounter()
#define X6_PROF_MARK(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); pm_[i] += n_ - pl_; pl_ = n_; } while (0)
#define X6_PROF_DEP(v) asm volatile("" ::"v"(v))
#define X6_PROF_FLUSH() do { if (blockIdx.x == 0 && lane == 0 && (w == 0 || w == H / 16 - 1)) \
        for (int i_ = 0; i_ < 4; ++i_) g_x6_prof[w ? 1 : 0][i_] = pm_[i_]; } while (0)
#define X6_PROF_FLUSH8() do { if (blockIdx.x == 0 && lane == 0 && (w == 0 || w == H / 16 - 1)) \
        for (int i_ = 0; i_ < 8; ++i_) g_x6_prof[w ? 1 : 0][i_] = pm_[i_]; } while (0)
extern "C" int uav_x6_prof_read(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_x6_prof), sizeof(g_x6_prof)) == hipSuccess ? 0 : 1;
}
#else
#define X6_PROF_DECL
#define X6_PROF_MARK(i)
#define X6_PROF_DEP(v)
#define X6_PROF_FLUSH()
#define X6_PROF_FLUSH8()
#endif

// Orientation: gates^T = W h^T, i.e. the WEIGHTS are the MFMA A operand (16 units of one gate) and h^T the B
// operand (16 envs), so a lane's four accumulator rows are four CONSECUTIVE units of ONE env: y and the five
// stash values leave as dwordx4 stores (6 per step instead of 24 dword stores -- the vector-memory issue rate,
// 16 cycles per wave-instruction, paced the step as much as the MFMAs did), h_t is parked with one ds_write_b64
// per piece, and keep is one value per lane.
template <int H, bool FUSE_X>
__global__ __launch_bounds__(H * 4) void lstm_fwd_x6_kernel(
    const float* __restrict__ x, const float* __restrict__ keep, const float* __restrict__ h0,
    const float* __restrict__ c0, const float* __restrict__ w_ih, const float* __restrict__ w_hh,
    const float* __restrict__ b_ih, const float* __restrict__ b_hh, int N, int T, int I,
    float* __restrict__ y, float* __restrict__ hn, float* __restrict__ cn, float* __restrict__ stash,
    const float* __restrict__ w_head, const float* __restrict__ b_head, int NHD, float* __restrict__ heads) {
    using G = FwdX6Geom<H>;
    constexpr int NS = G::NS, RS = G::RS, PLANE = G::PLANE, QL = G::QL, WPARK = G::WPARK, TC = G::TCX, XPT = G::XPT;
    constexpr int NT = H * 4, HPL = G::HPL, NW = H / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* hpl = reinterpret_cast<unsigned short*>(smem);       // [2][3 pieces][MT][RS] bf16
    unsigned short* wpark = hpl + 2 * 3 * PLANE;                         // [waves][QL][NS][64 lanes][8] bf16
    float* xbuf = reinterpret_cast<float*>(wpark + (H / 16) * WPARK);    // [2][TC][MT][8]
    float* kbuf = xbuf + 2 * TC * MT * 8;                                // [2][TC][MT]: keep[t + 1] of the chunk's steps
    float* wxl = kbuf + 2 * TC * MT;                                     // [waves][4 gates][2 k-steps][64 lanes]
    float* bl = wxl + (H / 16) * 512;                                    // [4H] b_ih + b_hh | [8] head bias
    unsigned short* whp = reinterpret_cast<unsigned short*>(bl + 4 * H + 8);   // [3 pieces][8 heads][RS] bf16

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int uw = 16 * w + j;                 // unit whose weight row this lane holds (A operand row)
    const int uo = 16 * w + 4 * kq;            // first of this lane's four output units; its env is j
    const int n0 = blockIdx.x * MT;
    const int n = min(n0 + j, N - 1);
    const bool live = n0 + j < N;

    // A fragments of 16x16x32: lane (j, kq) holds k = 32 s + 8 kq .. + 7 of unit uw, per gate q and piece p
    bf16x8 wb[4][NS][2], wb2[4 - QL][NS];
    bf16x8* const wpk = reinterpret_cast<bf16x8*>(wpark + w * WPARK) + lane;      // + (q * NS + s) * 64
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float* src = w_hh + (size_t)(q * H + uw) * H + 32 * s + 8 * kq;
            const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            bf16x8 p2v;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __bf16 p0, p1, p2;
                split3(v[i], p0, p1, p2);
                wb[q][s][0][i] = p0; wb[q][s][1][i] = p1; p2v[i] = p2;
            }
            if (q < QL) wpk[(q * NS + s) * 64] = p2v;
            else wb2[q < QL ? 0 : q - QL][s] = p2v;
        }
    float* const wxw = wxl + w * 512 + lane;                             // this lane's W_ih fragments: + (2 q + s) * 64
    if (FUSE_X) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int k = 2 * kq + s;
                wxw[(2 * q + s) * 64] = (k < I) ? w_ih[(size_t)(q * H + uw) * I + k] : 0.f;
            }
        for (int idx = threadIdx.x; idx < 4 * H; idx += NT) bl[idx] = b_ih[idx] + b_hh[idx];
    }
    if (heads) {                                                         // actor / critic head rows as bf16 piece planes
        for (int idx = threadIdx.x; idx < 8 * H; idx += NT) {
            const int hdx = idx / H, uu = idx % H;
            __bf16 p0, p1, p2;
            split3((hdx < NHD) ? w_head[(size_t)hdx * H + uu] : 0.f, p0, p1, p2);
            unsigned short* d = whp + hdx * RS + uu;
            d[0] = bf_bits(p0); d[HPL] = bf_bits(p1); d[2 * HPL] = bf_bits(p2);
        }
        if (threadIdx.x < 8) bl[4 * H + threadIdx.x] = (int)threadIdx.x < NHD ? b_head[threadIdx.x] : 0.f;
    }
    // heads of the h held in plane set `buf` (= h_t, UNMASKED): D[head 4kq + r][env j] = W_head h^T + b, stored to
    // heads[env][t][NHD].  One wave does it, beside its own recurrent MFMAs of the next step; the loss then reads
    // NHD floats per sample instead of the H floats of y.
    auto emit_heads = [&](int buf, int t) {
        f32x4 ha = {0.f, 0.f, 0.f, 0.f};
        const unsigned short* hrow = hpl + buf * 3 * PLANE + j * RS + 8 * kq;
        const unsigned short* wrow = whp + (j & 7) * RS + 8 * kq;
        auto hp = [&](int pc, int s) { return *reinterpret_cast<const bf16x8*>(hrow + pc * PLANE + 32 * s); };
        auto wp = [&](int pc, int s) { return *reinterpret_cast<const bf16x8*>(wrow + pc * HPL + 32 * s); };
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            ha = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp(0, s), hp(2, s), ha, 0, 0, 0);
            ha = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp(1, s), hp(1, s), ha, 0, 0, 0);
            ha = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp(2, s), hp(0, s), ha, 0, 0, 0);
            ha = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp(0, s), hp(1, s), ha, 0, 0, 0);
            ha = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp(1, s), hp(0, s), ha, 0, 0, 0);
            ha = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp(0, s), hp(0, s), ha, 0, 0, 0);
            asm volatile("" ::: "memory");               // one slab's fragments at a time
        }
        if (live && kq < 2) {
            float* dst = heads + ((size_t)n * T + t) * NHD;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * kq + r < NHD) dst[4 * kq + r] = ha[r] + bl[4 * H + 4 * kq + r];
        }
    };
    auto put_h = [&](unsigned short* plane0, const float (&hv)[4]) {     // split and park h[env j][uo .. uo+3]
        unsigned short b[3][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            __bf16 p0, p1, p2;
            split3(hv[r], p0, p1, p2);
            b[0][r] = bf_bits(p0); b[1][r] = bf_bits(p1); b[2][r] = bf_bits(p2);
        }
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
            uint2 v;
            v.x = (unsigned)b[pc][0] | ((unsigned)b[pc][1] << 16);
            v.y = (unsigned)b[pc][2] | ((unsigned)b[pc][3] << 16);
            *reinterpret_cast<uint2*>(plane0 + pc * PLANE + j * RS + uo) = v;
        }
    };

    // chunk staging through registers: chunk c + 1 is loaded while chunk c runs, committed to the other buffer
    float xr[XPT], kr = 1.f;
    auto stage_load = [&](int t0) {
        if (FUSE_X) {
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                const int idx = threadIdx.x + i * NT;                    // (e, tt, f) with f fastest
                const int e = min(idx / (TC * 8), MT - 1), tt = (idx >> 3) % TC, f = idx & 7;
                const int ne = min(n0 + e, N - 1), t = min(t0 + tt, T - 1);
                xr[i] = (f < I) ? x[((size_t)ne * T + t) * I + f] : 0.f;
            }
        }
        if (threadIdx.x < TC * MT) {
            const int e = threadIdx.x / TC, tt = threadIdx.x % TC;
            const int ne = min(n0 + e, N - 1), t = t0 + tt + 1;
            kr = (keep && t < T) ? keep[(size_t)ne * T + t] : 1.f;
        }
    };
    auto stage_commit = [&](int buf) {
        if (FUSE_X) {
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                const int idx = threadIdx.x + i * NT;
                const int e = idx / (TC * 8), tt = (idx >> 3) % TC, f = idx & 7;
                if (e < MT) xbuf[((buf * TC + tt) * MT + e) * 8 + f] = xr[i];
            }
        }
        if (threadIdx.x < TC * MT) kbuf[(buf * TC + threadIdx.x % TC) * MT + threadIdx.x / TC] = kr;
    };

    float c_reg[4];
    {
        const float k0 = keep ? keep[(size_t)n * T] : 1.f;
        const float4 cv = *reinterpret_cast<const float4*>(c0 + (size_t)n * H + uo);
        const float4 hv4 = *reinterpret_cast<const float4*>(h0 + (size_t)n * H + uo);
        c_reg[0] = cv.x * k0; c_reg[1] = cv.y * k0; c_reg[2] = cv.z * k0; c_reg[3] = cv.w * k0;
        const float hv[4] = {hv4.x * k0, hv4.y * k0, hv4.z * k0, hv4.w * k0};
        put_h(hpl, hv);
        if (stash && I > 6 && live)                                      // h_prev of step 0 (generic wgrad path)
            *reinterpret_cast<float4*>(stash + ((size_t)n * T) * (6 * H) + 5 * H + uo) = float4{hv[0], hv[1], hv[2], hv[3]};
    }
    stage_load(0);
    stage_commit(0);
    int cur = 0;
    float kcur = 1.f;            // keep of the step about to run (the mask on the incoming h); h0 is masked above
    X6_PROF_DECL;
    lds_barrier();

    const int nchunk = (T + TC - 1) / TC;
    for (int ch = 0; ch < nchunk; ++ch) {
        const int t0 = ch * TC, tc = min(TC, T - t0), xb = ch & 1;
        if (ch + 1 < nchunk) stage_load(t0 + TC);
        for (int tt = 0; tt < tc; ++tt) {
            const int t = t0 + tt;
            const size_t row = (size_t)n * T + t;
            X6_PROF_MARK(0);
            if (heads && w == NW - 1 && t > 0) emit_heads(cur, t - 1);   // heads of h_{t-1}, while no accumulator is live
            // the planes hold h_{t-1} UNMASKED (the heads need it so); the episode mask k_t is a per-env scalar, so it
            // is applied to the finished h-part of the accumulator: acc = k_t (W_hh h_{t-1}) + bias + W_ih x_t
            f32x4 acc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            const unsigned short* hrow = hpl + cur * 3 * PLANE + j * RS + 8 * kq;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(hrow + 32 * s);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(hrow + PLANE + 32 * s);
                const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(hrow + 2 * PLANE + 32 * s);
                // smallest products first; four independent accumulators between dependent MFMAs
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[q][s][0], a2, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[q][s][1], a1, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bf16x8 b2 = (q < QL) ? wpk[(q * NS + s) * 64] : wb2[q < QL ? 0 : q - QL][s];
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b2, a0, acc[q], 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[q][s][0], a1, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[q][s][1], a0, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[q][s][0], a0, acc[q], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = FUSE_X ? *reinterpret_cast<const float4*>(bl + q * H + uo)
                                        : *reinterpret_cast<const float4*>(stash + row * (6 * H) + q * H + uo);
                acc[q] = acc[q] * kcur + f32x4{v.x, v.y, v.z, v.w};
            }
            if (FUSE_X) {       // K = I <= 8 input projection: two exact-f32 k-steps
                const float2 ax = *reinterpret_cast<const float2*>(&xbuf[((xb * TC + tt) * MT + j) * 8 + 2 * kq]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wxw[(2 * q) * 64], ax.x, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wxw[(2 * q + 1) * 64], ax.y, acc[q], 0, 0, 0);
                }
            }
            X6_PROF_DEP(acc[0][0]); X6_PROF_DEP(acc[1][1]); X6_PROF_DEP(acc[2][2]); X6_PROF_DEP(acc[3][3]);
            X6_PROF_MARK(1);
            const float kn = kbuf[(xb * TC + tt) * MT + j];              // keep of step t+1 (1 past the end)
            // the stash stores are issued as soon as their values exist, so the write stream starts under the rest of
            // the gate math instead of in one burst before the barrier
            float gi[4], gf[4], gg[4], go[4], cp[4], hh[4], hm[4], cc[4];
            float* sp = stash + row * (6 * H) + uo;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                gi[r] = sigmoidf_(acc[0][r]); gf[r] = sigmoidf_(acc[1][r]); gg[r] = tanhf_(acc[2][r]);
                cp[r] = c_reg[r];
                cc[r] = gf[r] * cp[r] + gi[r] * gg[r];
            }
            if (live && stash) {
                *reinterpret_cast<float4*>(sp) = float4{gi[0], gi[1], gi[2], gi[3]};
                *reinterpret_cast<float4*>(sp + H) = float4{gf[0], gf[1], gf[2], gf[3]};
                *reinterpret_cast<float4*>(sp + 2 * H) = float4{gg[0], gg[1], gg[2], gg[3]};
                *reinterpret_cast<float4*>(sp + 4 * H) = float4{cp[0], cp[1], cp[2], cp[3]};
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                go[r] = sigmoidf_(acc[3][r]);
                hh[r] = go[r] * tanhf_(cc[r]);
                hm[r] = hh[r] * kn;
                c_reg[r] = (t == T - 1) ? cc[r] : cc[r] * kn;            // cn is the unmasked final cell state
            }
            put_h(hpl + (cur ^ 1) * 3 * PLANE, hh);
            kcur = kn;
            if (live) {
                *reinterpret_cast<float4*>(y + row * H + uo) = float4{hh[0], hh[1], hh[2], hh[3]};
                if (stash) {
                    *reinterpret_cast<float4*>(sp + 3 * H) = float4{go[0], go[1], go[2], go[3]};
                    if (I > 6 && t + 1 < T)                              // h_prev of step t+1 (generic wgrad path)
                        *reinterpret_cast<float4*>(sp + 6 * H + 5 * H) = float4{hm[0], hm[1], hm[2], hm[3]};
                }
                if (t == T - 1) {
                    *reinterpret_cast<float4*>(hn + (size_t)n * H + uo) = float4{hh[0], hh[1], hh[2], hh[3]};
                    *reinterpret_cast<float4*>(cn + (size_t)n * H + uo) = float4{c_reg[0], c_reg[1], c_reg[2], c_reg[3]};
                }
            }
            cur ^= 1;
            X6_PROF_MARK(2);
            lds_barrier();
            X6_PROF_MARK(3);
        }
        if (ch + 1 < nchunk) {
            stage_commit(xb ^ 1);
            lds_barrier();
        }
    }
    if (heads && w == NW - 1) emit_heads(cur, T - 1);                     // planes `cur` hold h_{T-1}
    X6_PROF_FLUSH();
}


// ------------------------------------------------------------------------- forward, split-fp16 MFMA ("h3")
// The same kernel with the recurrent product as THREE fp16 MFMA products per K = 32 slab (common.h, split2h): W_hh and
// h_t are carried as two fp16 pieces each, main and cross products accumulate separately and are combined once per step.
// Half the matrix instructions of the bf16 split, the weight pieces take exactly the f32 weights' 128 VGPRs (no LDS
// slab), two h planes instead of three -- and a smaller error (tools/f16x3_probe.hip).  h is in (-1, 1) and the
// weights are O(1), far inside fp16's range; values below its normal range lose relative, not absolute, accuracy
// (absolute error <= 2^-36), which is what a dot product needs.
template <int H>
struct FwdH3Geom {
    static constexpr int NS = H / 32;
    static constexpr int RS = H + 8;
    static constexpr int PLANE = MT * RS;
    static constexpr int TCX = 8;
    static constexpr int XPT = (MT * TCX * 8 + H * 4 - 1) / (H * 4);
    static constexpr int HPL = 8 * RS;
    static constexpr size_t LDS = (2 * 2 * PLANE + 2 * HPL) * sizeof(unsigned short) +
                                  (2 * TCX * MT * 8 + 2 * TCX * MT + (H / 16) * 8 * 64 + 4 * H + 8) * sizeof(float);
};

template <int H, bool FUSE_X>
__global__ __launch_bounds__(H * 4) void lstm_fwd_h3_kernel(
    const float* __restrict__ x, const float* __restrict__ keep, const float* __restrict__ h0,
    const float* __restrict__ c0, const float* __restrict__ w_ih, const float* __restrict__ w_hh,
    const float* __restrict__ b_ih, const float* __restrict__ b_hh, int N, int T, int I,
    float* __restrict__ y, float* __restrict__ hn, float* __restrict__ cn, float* __restrict__ stash,
    const float* __restrict__ w_head, const float* __restrict__ b_head, int NHD, float* __restrict__ heads) {
    using G = FwdH3Geom<H>;
    constexpr int NS = G::NS, RS = G::RS, PLANE = G::PLANE, TC = G::TCX, XPT = G::XPT;
    constexpr int NT = H * 4, HPL = G::HPL, NW = H / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* hpl = reinterpret_cast<unsigned short*>(smem);       // [2][2 pieces][MT][RS] fp16
    float* xbuf = reinterpret_cast<float*>(hpl + 2 * 2 * PLANE);         // [2][TC][MT][8]
    float* kbuf = xbuf + 2 * TC * MT * 8;                                // [2][TC][MT]: keep[t + 1] of the chunk's steps
    float* wxl = kbuf + 2 * TC * MT;                                     // [waves][4 gates][2 k-steps][64 lanes]
    float* bl = wxl + (H / 16) * 512;                                    // [4H] b_ih + b_hh | [8] head bias
    unsigned short* whp = reinterpret_cast<unsigned short*>(bl + 4 * H + 8);   // [2 pieces][8 heads][RS] fp16

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int uw = 16 * w + j;                 // unit whose weight row this lane holds (A operand row)
    const int uo = 16 * w + 4 * kq;            // first of this lane's four output units; its env is j
    const int n0 = blockIdx.x * MT;
    const int n = min(n0 + j, N - 1);
    const bool live = n0 + j < N;

    // A fragments of 16x16x32: lane (j, kq) holds k = 32 s + 8 kq .. + 7 of unit uw, per gate q and piece p
    f16x8 wb[4][NS][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float* src = w_hh + (size_t)(q * H + uw) * H + 32 * s + 8 * kq;
            const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                _Float16 p0, p1;
                split2h(v[i], p0, p1);
                wb[q][s][0][i] = p0; wb[q][s][1][i] = p1;
            }
        }
    float* const wxw = wxl + w * 512 + lane;                             // this lane's W_ih fragments: + (2 q + s) * 64
    if (FUSE_X) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int k = 2 * kq + s;
                wxw[(2 * q + s) * 64] = (k < I) ? w_ih[(size_t)(q * H + uw) * I + k] : 0.f;
            }
        for (int idx = threadIdx.x; idx < 4 * H; idx += NT) bl[idx] = b_ih[idx] + b_hh[idx];
    }
    if (heads) {                                                         // actor / critic head rows as fp16 piece planes
        for (int idx = threadIdx.x; idx < 8 * H; idx += NT) {
            const int hdx = idx / H, uu = idx % H;
            _Float16 p0, p1;
            split2h((hdx < NHD) ? w_head[(size_t)hdx * H + uu] : 0.f, p0, p1);
            unsigned short* d = whp + hdx * RS + uu;
            d[0] = h_bits(p0); d[HPL] = h_bits(p1);
        }
        if (threadIdx.x < 8) bl[4 * H + threadIdx.x] = (int)threadIdx.x < NHD ? b_head[threadIdx.x] : 0.f;
    }
    // heads of the h held in plane set `buf` (= h_t, UNMASKED): D[head 4kq + r][env j] = W_head h^T + b, stored to
    // heads[env][t][NHD].  One wave does it, beside its own recurrent MFMAs of the next step.
    auto emit_heads = [&](int buf, int t) {
        f32x4 ha = {0.f, 0.f, 0.f, 0.f}, hb = ha;
        const unsigned short* hrow = hpl + buf * 2 * PLANE + j * RS + 8 * kq;
        const unsigned short* wrow = whp + (j & 7) * RS + 8 * kq;
        auto hp = [&](int pc, int s) { return *reinterpret_cast<const f16x8*>(hrow + pc * PLANE + 32 * s); };
        auto wp = [&](int pc, int s) { return *reinterpret_cast<const f16x8*>(wrow + pc * HPL + 32 * s); };
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            hb = __builtin_amdgcn_mfma_f32_16x16x32_f16(wp(1, s), hp(0, s), hb, 0, 0, 0);
            hb = __builtin_amdgcn_mfma_f32_16x16x32_f16(wp(0, s), hp(1, s), hb, 0, 0, 0);
            ha = __builtin_amdgcn_mfma_f32_16x16x32_f16(wp(0, s), hp(0, s), ha, 0, 0, 0);
            asm volatile("" ::: "memory");               // one slab's fragments at a time
        }
        if (live && kq < 2) {
            float* dst = heads + ((size_t)n * T + t) * NHD;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * kq + r < NHD) dst[4 * kq + r] = (ha[r] + H3_LO * hb[r]) + bl[4 * H + 4 * kq + r];
        }
    };
    auto put_h = [&](unsigned short* plane0, const float (&hv)[4]) {     // split and park h[env j][uo .. uo+3]
        unsigned short b[2][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            _Float16 p0, p1;
            split2h(hv[r], p0, p1);
            b[0][r] = h_bits(p0); b[1][r] = h_bits(p1);
        }
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            uint2 v;
            v.x = (unsigned)b[pc][0] | ((unsigned)b[pc][1] << 16);
            v.y = (unsigned)b[pc][2] | ((unsigned)b[pc][3] << 16);
            *reinterpret_cast<uint2*>(plane0 + pc * PLANE + j * RS + uo) = v;
        }
    };

    // chunk staging through registers: chunk c + 1 is loaded while chunk c runs, committed to the other buffer
    float xr[XPT], kr = 1.f;
    auto stage_load = [&](int t0) {
        if (FUSE_X) {
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                const int idx = threadIdx.x + i * NT;                    // (e, tt, f) with f fastest
                const int e = min(idx / (TC * 8), MT - 1), tt = (idx >> 3) % TC, f = idx & 7;
                const int ne = min(n0 + e, N - 1), t = min(t0 + tt, T - 1);
                xr[i] = (f < I) ? x[((size_t)ne * T + t) * I + f] : 0.f;
            }
        }
        if (threadIdx.x < TC * MT) {
            const int e = threadIdx.x / TC, tt = threadIdx.x % TC;
            const int ne = min(n0 + e, N - 1), t = t0 + tt + 1;
            kr = (keep && t < T) ? keep[(size_t)ne * T + t] : 1.f;
        }
    };
    auto stage_commit = [&](int buf) {
        if (FUSE_X) {
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                const int idx = threadIdx.x + i * NT;
                const int e = idx / (TC * 8), tt = (idx >> 3) % TC, f = idx & 7;
                if (e < MT) xbuf[((buf * TC + tt) * MT + e) * 8 + f] = xr[i];
            }
        }
        if (threadIdx.x < TC * MT) kbuf[(buf * TC + threadIdx.x % TC) * MT + threadIdx.x / TC] = kr;
    };

    float c_reg[4];
    {
        const float k0 = keep ? keep[(size_t)n * T] : 1.f;
        const float4 cv = *reinterpret_cast<const float4*>(c0 + (size_t)n * H + uo);
        const float4 hv4 = *reinterpret_cast<const float4*>(h0 + (size_t)n * H + uo);
        c_reg[0] = cv.x * k0; c_reg[1] = cv.y * k0; c_reg[2] = cv.z * k0; c_reg[3] = cv.w * k0;
        const float hv[4] = {hv4.x * k0, hv4.y * k0, hv4.z * k0, hv4.w * k0};
        put_h(hpl, hv);
        if (stash && I > 6 && live)                                      // h_prev of step 0 (generic wgrad path)
            *reinterpret_cast<float4*>(stash + ((size_t)n * T) * (6 * H) + 5 * H + uo) = float4{hv[0], hv[1], hv[2], hv[3]};
    }
    stage_load(0);
    stage_commit(0);
    int cur = 0;
    float kcur = 1.f;            // keep of the step about to run (the mask on the incoming h); h0 is masked above
    X6_PROF_DECL;
    lds_barrier();

    const int nchunk = (T + TC - 1) / TC;
    for (int ch = 0; ch < nchunk; ++ch) {
        const int t0 = ch * TC, tc = min(TC, T - t0), xb = ch & 1;
        if (ch + 1 < nchunk) stage_load(t0 + TC);
        for (int tt = 0; tt < tc; ++tt) {
            const int t = t0 + tt;
            const size_t row = (size_t)n * T + t;
            X6_PROF_MARK(0);
            if (heads && w == NW - 1 && t > 0) emit_heads(cur, t - 1);   // heads of h_{t-1}, while no accumulator is live
            // the planes hold h_{t-1} UNMASKED (the heads need it so); the episode mask k_t is a per-env scalar, so it
            // is applied to the finished h-part of the accumulator: acc = k_t (W_hh h_{t-1}) + bias + W_ih x_t
            f32x4 acc[4], acl[4];                                         // main products | cross products (x 2^-11)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = acl[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            const unsigned short* hrow = hpl + cur * 2 * PLANE + j * RS + 8 * kq;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const f16x8 a0 = *reinterpret_cast<const f16x8*>(hrow + 32 * s);
                const f16x8 a1 = *reinterpret_cast<const f16x8*>(hrow + PLANE + 32 * s);
                // eight independent accumulators between dependent MFMAs
#pragma unroll
                for (int q = 0; q < 4; ++q) acl[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[q][s][1], a0, acl[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[q][s][0], a0, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acl[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[q][s][0], a1, acl[q], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = FUSE_X ? *reinterpret_cast<const float4*>(bl + q * H + uo)
                                        : *reinterpret_cast<const float4*>(stash + row * (6 * H) + q * H + uo);
                acc[q] = (acc[q] + acl[q] * H3_LO) * kcur + f32x4{v.x, v.y, v.z, v.w};
            }
            if (FUSE_X) {       // K = I <= 8 input projection: two exact-f32 k-steps
                const float2 ax = *reinterpret_cast<const float2*>(&xbuf[((xb * TC + tt) * MT + j) * 8 + 2 * kq]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wxw[(2 * q) * 64], ax.x, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wxw[(2 * q + 1) * 64], ax.y, acc[q], 0, 0, 0);
                }
            }
            X6_PROF_DEP(acc[0][0]); X6_PROF_DEP(acc[1][1]); X6_PROF_DEP(acc[2][2]); X6_PROF_DEP(acc[3][3]);
            X6_PROF_MARK(1);
            const float kn = kbuf[(xb * TC + tt) * MT + j];              // keep of step t+1 (1 past the end)
            float gi[4], gf[4], gg[4], go[4], cp[4], hh[4], hm[4], cc[4];
            float* sp = stash + row * (6 * H) + uo;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                gi[r] = sigmoidf_(acc[0][r]); gf[r] = sigmoidf_(acc[1][r]); gg[r] = tanhf_(acc[2][r]);
                cp[r] = c_reg[r];
                cc[r] = gf[r] * cp[r] + gi[r] * gg[r];
            }
            if (live && stash) {
                *reinterpret_cast<float4*>(sp) = float4{gi[0], gi[1], gi[2], gi[3]};
                *reinterpret_cast<float4*>(sp + H) = float4{gf[0], gf[1], gf[2], gf[3]};
                *reinterpret_cast<float4*>(sp + 2 * H) = float4{gg[0], gg[1], gg[2], gg[3]};
                *reinterpret_cast<float4*>(sp + 4 * H) = float4{cp[0], cp[1], cp[2], cp[3]};
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                go[r] = sigmoidf_(acc[3][r]);
                hh[r] = go[r] * tanhf_(cc[r]);
                hm[r] = hh[r] * kn;
                c_reg[r] = (t == T - 1) ? cc[r] : cc[r] * kn;            // cn is the unmasked final cell state
            }
            put_h(hpl + (cur ^ 1) * 2 * PLANE, hh);
            kcur = kn;
            if (live) {
                *reinterpret_cast<float4*>(y + row * H + uo) = float4{hh[0], hh[1], hh[2], hh[3]};
                if (stash) {
                    *reinterpret_cast<float4*>(sp + 3 * H) = float4{go[0], go[1], go[2], go[3]};
                    if (I > 6 && t + 1 < T)                              // h_prev of step t+1 (generic wgrad path)
                        *reinterpret_cast<float4*>(sp + 6 * H + 5 * H) = float4{hm[0], hm[1], hm[2], hm[3]};
                }
                if (t == T - 1) {
                    *reinterpret_cast<float4*>(hn + (size_t)n * H + uo) = float4{hh[0], hh[1], hh[2], hh[3]};
                    *reinterpret_cast<float4*>(cn + (size_t)n * H + uo) = float4{c_reg[0], c_reg[1], c_reg[2], c_reg[3]};
                }
            }
            cur ^= 1;
            X6_PROF_MARK(2);
            lds_barrier();
            X6_PROF_MARK(3);
        }
        if (ch + 1 < nchunk) {
            stage_commit(xb ^ 1);
            lds_barrier();
        }
    }
    if (heads && w == NW - 1) emit_heads(cur, T - 1);                     // planes `cur` hold h_{T-1}
    X6_PROF_FLUSH();
}

// ---------------------------------------------------------------------------------------- backward
template <int H>
struct BwdGeom {
    static constexpr int S = 4 * H + 4;        // padded row of the dgates tile (conflict-free b128)
    static constexpr size_t LDS = (2 * MT * S + TC * MT * 8 + (TC + 1) * MT) * sizeof(float);
};

// dy source: either dy[N][T][H] or (dheads[N][T][NH], w_head[NH][H]) with dy = dheads . w_head
template <int H>
__global__ __launch_bounds__(H * 4) void lstm_bwd_kernel(
    const float* __restrict__ keep, const float* __restrict__ stash, const float* __restrict__ w_hh,
    const float* __restrict__ dy, const float* __restrict__ dheads, const float* __restrict__ w_head, int NH,
    const float* __restrict__ dhn, const float* __restrict__ dcn, int N, int T, float* __restrict__ dgates,
    float* __restrict__ dh0, float* __restrict__ dc0) {
    using G = BwdGeom<H>;
    constexpr int S = G::S;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dgbuf = smem;                       // [2][MT][S]
    float* dhbuf = dgbuf + 2 * MT * S;         // [TC][MT][8]   staged dheads
    float* kbuf = dhbuf + TC * MT * 8;         // [TC+1][MT]    keep[t0 .. t0+tc]

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int u = 16 * w + j;
    const int n0 = blockIdx.x * MT;

    // W_hh^T slice: B[k][col j] = W_hh[k][u]; lane group kq covers gate kq's rows (k = kq*H + s)
    float wt[H];
#pragma unroll
    for (int s = 0; s < H; ++s) wt[s] = w_hh[(size_t)(kq * H + s) * H + u];
    // head weights as MFMA B-fragments (k = head index): dy = dheads . W_head is two MFMA k-steps whose
    // result lands in the accumulator layout the pointwise needs (row = env, col = unit)
    float whb[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) whb[a] = (dheads && 4 * a + kq < NH) ? w_head[(size_t)(4 * a + kq) * H + u] : 0.f;

    float dh_rec[4], dc_next[4];
    size_t srow[4];                            // stash row base of this lane's 4 env rows at t = 0
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = min(n0 + 4 * kq + r, N - 1);
        dh_rec[r] = dhn ? dhn[(size_t)n * H + u] : 0.f;
        dc_next[r] = dcn ? dcn[(size_t)n * H + u] : 0.f;
        srow[r] = (size_t)n * T;
    }
    // software prefetch: the stash values (and dy) of step t-1 are loaded while step t's MFMAs run,
    // so the HBM latency never sits between two steps of the recurrence
    float pf[4][5], pdy[4];
    auto prefetch = [&](int t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* sp = stash + (srow[r] + t) * (6 * H);
            pf[r][0] = sp[u]; pf[r][1] = sp[H + u]; pf[r][2] = sp[2 * H + u]; pf[r][3] = sp[3 * H + u];
            pf[r][4] = sp[4 * H + u];
            pdy[r] = dy ? dy[(srow[r] + t) * H + u] : 0.f;
        }
    };
    prefetch(T - 1);
    int cur = 0;
    const int nchunk = (T + TC - 1) / TC;
    for (int ch = nchunk - 1; ch >= 0; --ch) {
        const int t0 = ch * TC, tc = min(TC, T - t0);
        if (dheads) {
            for (int idx = threadIdx.x; idx < MT * tc * 8; idx += blockDim.x) {
                const int e = idx / (tc * 8), rem = idx % (tc * 8), tt = rem >> 3, f = rem & 7;
                const int n = min(n0 + e, N - 1);
                dhbuf[(tt * MT + e) * 8 + f] = (f < NH) ? dheads[((size_t)n * T + t0 + tt) * NH + f] : 0.f;
            }
        }
        for (int idx = threadIdx.x; idx < MT * tc; idx += blockDim.x) {
            const int e = idx / tc, tt = idx % tc;
            const int n = min(n0 + e, N - 1);
            kbuf[tt * MT + e] = keep ? keep[(size_t)n * T + t0 + tt] : 1.f;
        }
        lds_barrier();
        for (int tt = tc - 1; tt >= 0; --tt) {
            const int t = t0 + tt;
            float* dgw = dgbuf + cur * MT * S;
            f32x4 dyacc = {0.f, 0.f, 0.f, 0.f};
            if (dheads) {
                const float* dr = &dhbuf[(tt * MT + j) * 8];       // A-fragment: dheads[env j][head kq / 4+kq]
                dyacc = __builtin_amdgcn_mfma_f32_16x16x4f32(dr[kq], whb[0], dyacc, 0, 0, 0);
                dyacc = __builtin_amdgcn_mfma_f32_16x16x4f32(dr[4 + kq], whb[1], dyacc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = 4 * kq + r;
                const float gi = pf[r][0], gf = pf[r][1], gg = pf[r][2], go = pf[r][3], cp = pf[r][4];
                const float dyv = dheads ? dyacc[r] : pdy[r];
                const float dh = dyv + dh_rec[r];
                const float c = gf * cp + gi * gg;
                const float tch = tanhf_(c);
                const float dc = dh * go * (1.0f - tch * tch) + dc_next[r];
                const float dgi = dc * gg * gi * (1.0f - gi);
                const float dgf = dc * cp * gf * (1.0f - gf);
                const float dgg = dc * gi * (1.0f - gg * gg);
                const float dgo = dh * tch * go * (1.0f - go);
                const float kp = kbuf[tt * MT + e];            // keep[t]: the incoming state was masked by it
                dc_next[r] = dc * gf * kp;
                dgw[e * S + 0 * H + u] = dgi;
                dgw[e * S + 1 * H + u] = dgf;
                dgw[e * S + 2 * H + u] = dgg;
                dgw[e * S + 3 * H + u] = dgo;
                if (n0 + e < N) {
                    float* gp = dgates + (srow[r] + t) * (4 * H);
                    gp[u] = dgi; gp[H + u] = dgf; gp[2 * H + u] = dgg; gp[3 * H + u] = dgo;
                }
            }
            if (t > 0) prefetch(t - 1);
            lds_barrier();
            // dh_{t-1}[env][u] = sum_k dgates[env][k] W_hh[k][u]; four independent accumulators
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
            const float* drow = dgw + j * S + kq * H;
#pragma unroll
            for (int s = 0; s < H; s += 4) {
                const float4 a = *reinterpret_cast<const float4*>(drow + s);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wt[s], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wt[s + 1], a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wt[s + 2], a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wt[s + 3], a3, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float kp = kbuf[tt * MT + 4 * kq + r];
                dh_rec[r] = ((a0[r] + a1[r]) + (a2[r] + a3[r])) * kp;
            }
            cur ^= 1;
        }
        lds_barrier();     // kbuf / dhbuf are restaged by the next chunk
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + 4 * kq + r;
        if (n < N) {
            if (dh0) dh0[(size_t)n * H + u] = dh_rec[r];
            if (dc0) dc0[(size_t)n * H + u] = dc_next[r];
        }
    }
}

// ------------------------------------------------------- backward, split-bf16 MFMA, K split over the waves
// dh_{t-1}^T = W_hh^T dG_t^T on the bf16 matrix pipe at f32 accuracy (3-way split, six products; see
// lstm_fwd_x6_kernel), weights as the A operand: lane (j, kq) owns env j and the four consecutive units uo..uo+3.
// Giving every wave 16 OUTPUT units and the whole K = 4H of dG (the first form built) makes each wave read all three
// piece planes of the step's gate gradients back from LDS (48 b128 fragments, 384 KB per step and CU -- as many cycles
// of LDS bandwidth as the MFMAs take) behind a barrier: 0.52 of 0.62 ms with HBM taken out of the picture.
// Here the K dimension is split instead: wave w multiplies ONLY the 64 gate gradients it produced itself
// (gates x its 16 units -- they already sit in the right lanes: the MFMA's k order is free, so lane (j, kq)
// supplies k = (gate, unit 16w + 4kq + r) of its own env j) against W_hh[those 64 rows][all H units], and writes
// H/16 partial 16 x 16 tiles (f32) to LDS; after ONE barrier each wave sums the NW partials of its own tile in a
// fixed order.  LDS traffic per step drops from 384 + 48 KB to 64 + 64 KB, the gate gradients never go to LDS,
// and the MFMAs start right after the pointwise without waiting for anybody.  The stash ring needs a single
// slot: the refill for step t-1 is issued at the TOP of step t, right after the slot is read, a full step ahead.
// Stash by LDS-DMA (global_load_lds: no VGPR destination): each wave gathers exactly what its own lanes read back, so
// the only ordering needed is the wave's own counted s_waitcnt vmcnt(k) = "everything but the k ops issued last has
// landed" (loads, stores and LDS-DMA retire in issue order; the DMA ops of a step are issued unconditionally).
template <int H>
struct BwdKGeom {
    static constexpr int NW = H / 16;
    static constexpr int SLOT = 5 * MT * 16;                         // floats per wave in the stash ring (one slot)
    static constexpr int SMALL = 128;                                // floats per small slot: dheads[16][6] | keep[16]
    static constexpr int PK = (H >= 128) ? 6 : 0;                    // (tile, slab) items whose smallest weight piece is LDS-parked
    static constexpr int WPARK = PK * 64 * 8;                        // bf16 elements per wave
    static constexpr size_t LDS = (NW * NW * 64 * 4 /*partials*/ + NW * SLOT + 2 * SMALL) * sizeof(float) +
                                  NW * WPARK * sizeof(unsigned short);
};

template <int H>
__global__ __launch_bounds__(H * 4) void lstm_bwd_x6k_kernel(
    const float* __restrict__ keep, const float* __restrict__ stash, const float* __restrict__ w_hh,
    const float* __restrict__ dheads, const float* __restrict__ w_head, int NH, const float* __restrict__ dhn,
    const float* __restrict__ dcn, int N, int T, float* __restrict__ dgates, float* __restrict__ dh0,
    float* __restrict__ dc0) {
    using G = BwdKGeom<H>;
    constexpr int SLOT = G::SLOT, NW = G::NW, SMALL = G::SMALL, PK = G::PK, WPARK = G::WPARK;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* part = reinterpret_cast<f32x4*>(smem);                           // [tile m][wave w][64 lanes]
    float* ring = smem + NW * NW * 64 * 4;                                  // [NW][SLOT]
    float* small = ring + NW * SLOT;                                        // [2 slots][SMALL]
    unsigned short* wpark = reinterpret_cast<unsigned short*>(small + 2 * SMALL);   // [NW][PK][64 lanes][8]

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, kq = lane >> 4;
    const int uw = 16 * w + j;                 // A row of the dy product (head weights of unit uw)
    const int uo = 16 * w + 4 * kq;            // first of this lane's four units; its env is j
    const int n0 = blockIdx.x * MT;
    const int n = min(n0 + j, N - 1);
    const bool live = n0 + j < N;

    // A fragments: tile m (units 16m..16m+15), slab sb (gates 2sb, 2sb+1): lane (i = j, kq), element e holds
    // W_hh[(2sb + e/4) H + 16w + 4kq + e%4][16m + j] -- the k order of this wave's own gate gradients
    bf16x8 wa[NW][2][2], wa2[2 * NW - PK];
    bf16x8* const wpk = reinterpret_cast<bf16x8*>(wpark + w * WPARK) + lane;           // + item * 64
#pragma unroll
    for (int m = 0; m < NW; ++m)
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            bf16x8 p2v;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = (2 * sb + e / 4) * H + 16 * w + 4 * kq + (e % 4);
                __bf16 p0, p1, p2;
                split3(w_hh[(size_t)k * H + 16 * m + j], p0, p1, p2);
                wa[m][sb][0][e] = p0; wa[m][sb][1][e] = p1; p2v[e] = p2;
            }
            const int item = 2 * m + sb;
            if (item < PK) wpk[item * 64] = p2v;
            else wa2[item < PK ? 0 : item - PK] = p2v;
        }
    float whb[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) whb[a] = (4 * a + kq < NH) ? w_head[(size_t)(4 * a + kq) * H + uw] : 0.f;

    float dh_rec[4], dc_next[4];
    {
        const float4 a4 = dhn ? *reinterpret_cast<const float4*>(dhn + (size_t)n * H + uo) : float4{0.f, 0.f, 0.f, 0.f};
        const float4 c4 = dcn ? *reinterpret_cast<const float4*>(dcn + (size_t)n * H + uo) : float4{0.f, 0.f, 0.f, 0.f};
        dh_rec[0] = a4.x; dh_rec[1] = a4.y; dh_rec[2] = a4.z; dh_rec[3] = a4.w;
        dc_next[0] = c4.x; dc_next[1] = c4.y; dc_next[2] = c4.z; dc_next[3] = c4.w;
    }
    const size_t srow = (size_t)n * T;

    // ---- LDS-DMA (inline asm; ordering: comment above BwdKGeom): stash gather [q][env lane/4][units 4 (lane%4) ..]
    typedef __attribute__((address_space(3))) float lds_f;
    const int e_d = lane >> 2, g4 = lane & 3;
    const size_t drow = (size_t)min(n0 + e_d, N - 1) * T;
    const unsigned ring_base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((lds_f*)(ring + w * SLOT)));
    const unsigned small_base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((lds_f*)small));
    auto issue = [&](int t) {
        const float* src = stash + (drow + t) * (6 * H) + 16 * w + 4 * g4;
        unsigned m0save;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
            "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
            "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
            "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
            "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, off\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(m0save)
            : "v"(src), "v"(src + H), "v"(src + 2 * H), "v"(src + 3 * H), "v"(src + 4 * H), "s"(ring_base)
            : "memory", "scc");
    };
    // waves 0 / 1: one 256-B piece each of the step's small image  [dheads(16 envs x NH, env-major) | keep(16)].
    // The lane's element of step 0 and its per-step stride are fixed: only `+ t * stride` is left in the loop
    // (the division by NH used to run every step on the two waves everybody waits for).
    const float* small_at0;
    unsigned small_stride;
    {
        const int e = w * 64 + lane;
        if (e < 16 * NH) { small_at0 = dheads + (size_t)min(n0 + e / NH, N - 1) * T * NH + e % NH; small_stride = (unsigned)NH; }
        else if (e >= 112 && keep) { small_at0 = keep + (size_t)min(n0 + e - 112, N - 1) * T; small_stride = 1u; }
        else { small_at0 = w_hh + (lane & 15); small_stride = 0u; }         // padding: any readable dwords
    }
    auto issue_small = [&](int t, int slot) {
        const float* src = small_at0 + (size_t)((unsigned)t * small_stride);
        const unsigned dst = small_base + (unsigned)((slot * SMALL + w * 64) * 4);
        unsigned m0save;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0save) : "v"(src), "s"(dst) : "memory");
    };
    const bool small_wave = w < 2;
    // vector-memory operations of this wave, in issue order (all retire in order):
    //   prologue  ring(T-1) [small(T-1)] [small(T-2)]
    //   step t    wait . ring(t-1) . 4 dG stores . [small(t-2)]
    // so at the top of step t everything but the 4 stores (and one small piece) of step t+1 must have landed
    issue(T - 1);
    if (small_wave) {
        issue_small(T - 1, (T - 1) & 1);
        issue_small(T >= 2 ? T - 2 : 0, (T - 2) & 1);
    }
    bool first = true;
    X6_PROF_DECL;
    for (int t = T - 1; t >= 0; --t) {
        X6_PROF_MARK(0);
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (small_wave) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        first = false;
        const float* sl = ring + w * SLOT + j * 16 + 4 * kq;
        float4 pf[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) pf[q] = *reinterpret_cast<const float4*>(sl + q * 256);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the slot is in registers: refill it with step t-1
        issue(t >= 1 ? t - 1 : 0);
        X6_PROF_MARK(1);
        lds_barrier();           // b2: the partials of step t+1 have been summed by everyone; small image of step t landed
        X6_PROF_MARK(2);
        const float* sm = small + (t & 1) * SMALL;
        f32x4 dyacc = {0.f, 0.f, 0.f, 0.f};
        {
            const float d0 = (kq < NH) ? sm[j * NH + kq] : 0.f;
            const float d1 = (4 + kq < NH) ? sm[j * NH + 4 + kq] : 0.f;
            dyacc = __builtin_amdgcn_mfma_f32_16x16x4f32(whb[0], d0, dyacc, 0, 0, 0);
            dyacc = __builtin_amdgcn_mfma_f32_16x16x4f32(whb[1], d1, dyacc, 0, 0, 0);
        }
        const float kp = keep ? sm[112 + j] : 1.f;                           // keep[env j][t]
        const float gi[4] = {pf[0].x, pf[0].y, pf[0].z, pf[0].w}, gf[4] = {pf[1].x, pf[1].y, pf[1].z, pf[1].w};
        const float gg[4] = {pf[2].x, pf[2].y, pf[2].z, pf[2].w}, go[4] = {pf[3].x, pf[3].y, pf[3].z, pf[3].w};
        const float cp[4] = {pf[4].x, pf[4].y, pf[4].z, pf[4].w};
        float dg[4][4];                                                      // [gate][unit r]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dh = dyacc[r] + dh_rec[r];
            const float c = gf[r] * cp[r] + gi[r] * gg[r];
            const float tch = tanhf_(c);
            const float dc = dh * go[r] * (1.0f - tch * tch) + dc_next[r];
            dg[0][r] = dc * gg[r] * gi[r] * (1.0f - gi[r]);
            dg[1][r] = dc * cp[r] * gf[r] * (1.0f - gf[r]);
            dg[2][r] = dc * gi[r] * (1.0f - gg[r] * gg[r]);
            dg[3][r] = dh * tch * go[r] * (1.0f - go[r]);
            dc_next[r] = dc * gf[r] * kp;
        }
        // this wave's 64 gate gradients as B fragments: slab sb = gates 2sb, 2sb+1; element e = (gate 2sb + e/4, unit e%4)
        bf16x8 bp[2][3];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                __bf16 p0, p1, p2;
                split3(dg[q][r], p0, p1, p2);
                bp[q >> 1][0][4 * (q & 1) + r] = p0; bp[q >> 1][1][4 * (q & 1) + r] = p1; bp[q >> 1][2][4 * (q & 1) + r] = p2;
            }
            if (live)
                *reinterpret_cast<float4*>(dgates + (srow + t) * (4 * H) + q * H + uo) =
                    float4{dg[q][0], dg[q][1], dg[q][2], dg[q][3]};
        }
        X6_PROF_DEP(bp[1][2]); X6_PROF_DEP(bp[0][0]);
        X6_PROF_MARK(3);
        // partial dh^T tiles: D_m[unit 16m + 4kq + r][env j] over this wave's K range
#pragma unroll
        for (int m = 0; m < NW; ++m) {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                const int item = 2 * m + sb;
                const bf16x8 w2 = (item < PK) ? wpk[item * 64] : wa2[item < PK ? 0 : item - PK];
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[m][sb][0], bp[sb][2], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[m][sb][1], bp[sb][1], a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, bp[sb][0], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[m][sb][0], bp[sb][1], a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[m][sb][1], bp[sb][0], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[m][sb][0], bp[sb][0], a1, 0, 0, 0);
            }
            part[(m * NW + w) * 64 + lane] = a0 + a1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        X6_PROF_MARK(4);
        lds_barrier();           // b1: all partials written; nobody reads the small image of step t any more
        X6_PROF_MARK(5);
        if (small_wave) issue_small(t >= 2 ? t - 2 : 0, t & 1);
        f32x4 sum = part[(w * NW) * 64 + lane];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) sum += part[(w * NW + ww) * 64 + lane];      // fixed order: deterministic
#pragma unroll
        for (int r = 0; r < 4; ++r) dh_rec[r] = sum[r] * kp;
        X6_PROF_DEP(dh_rec[3]);
        X6_PROF_MARK(6);
    }
    X6_PROF_FLUSH8();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // retire the clamped tail DMAs before the LDS is released
    if (live) {
        if (dh0) *reinterpret_cast<float4*>(dh0 + (size_t)n * H + uo) = float4{dh_rec[0], dh_rec[1], dh_rec[2], dh_rec[3]};
        if (dc0) *reinterpret_cast<float4*>(dc0 + (size_t)n * H + uo) = float4{dc_next[0], dc_next[1], dc_next[2], dc_next[3]};
    }
}

// lstm_bwd_h3k_kernel: the K-split backward with the recurrent product as three fp16 MFMA products (common.h, split2h)
// instead of the bf16 split's six: W_hh as two fp16 pieces in exactly the f32 weights' 128 VGPRs (no LDS slab), and this
// step's gate gradients block-scaled per env by a power of two before the split (below).
template <int H>
struct BwdH3Geom {
    static constexpr int NW = H / 16;
    static constexpr int SLOT = 5 * MT * 16;                         // floats per wave in the stash ring (one slot)
    static constexpr int SMALL = 128;                                // floats per small slot: dheads[16][6] | keep[16]
    static constexpr size_t LDS = (NW * NW * 64 * 4 /*partials*/ + NW * SLOT + 2 * SMALL) * sizeof(float);
};

template <int H>
__global__ __launch_bounds__(H * 4) void lstm_bwd_h3k_kernel(
    const float* __restrict__ keep, const float* __restrict__ stash, const float* __restrict__ w_hh,
    const float* __restrict__ dheads, const float* __restrict__ w_head, int NH, const float* __restrict__ dhn,
    const float* __restrict__ dcn, int N, int T, float* __restrict__ dgates, float* __restrict__ dh0,
    float* __restrict__ dc0) {
    using G = BwdH3Geom<H>;
    constexpr int SLOT = G::SLOT, NW = G::NW, SMALL = G::SMALL;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* part = reinterpret_cast<f32x4*>(smem);                           // [tile m][wave w][64 lanes]
    float* ring = smem + NW * NW * 64 * 4;                                  // [NW][SLOT]
    float* small = ring + NW * SLOT;                                        // [2 slots][SMALL]

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, kq = lane >> 4;
    const int uw = 16 * w + j;                 // A row of the dy product (head weights of unit uw)
    const int uo = 16 * w + 4 * kq;            // first of this lane's four units; its env is j
    const int n0 = blockIdx.x * MT;
    const int n = min(n0 + j, N - 1);
    const bool live = n0 + j < N;

    // A fragments: tile m (units 16m..16m+15), slab sb (gates 2sb, 2sb+1): lane (i = j, kq), element e holds
    // W_hh[(2sb + e/4) H + 16w + 4kq + e%4][16m + j] -- the k order of this wave's own gate gradients
    f16x8 wa[NW][2][2];                                                      // two fp16 pieces: the f32 weights' 128 VGPRs
#pragma unroll
    for (int m = 0; m < NW; ++m)
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = (2 * sb + e / 4) * H + 16 * w + 4 * kq + (e % 4);
                _Float16 p0, p1;
                split2h(w_hh[(size_t)k * H + 16 * m + j], p0, p1);
                wa[m][sb][0][e] = p0; wa[m][sb][1][e] = p1;
            }
        }
    float whb[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) whb[a] = (4 * a + kq < NH) ? w_head[(size_t)(4 * a + kq) * H + uw] : 0.f;

    float dh_rec[4], dc_next[4];
    {
        const float4 a4 = dhn ? *reinterpret_cast<const float4*>(dhn + (size_t)n * H + uo) : float4{0.f, 0.f, 0.f, 0.f};
        const float4 c4 = dcn ? *reinterpret_cast<const float4*>(dcn + (size_t)n * H + uo) : float4{0.f, 0.f, 0.f, 0.f};
        dh_rec[0] = a4.x; dh_rec[1] = a4.y; dh_rec[2] = a4.z; dh_rec[3] = a4.w;
        dc_next[0] = c4.x; dc_next[1] = c4.y; dc_next[2] = c4.z; dc_next[3] = c4.w;
    }
    const size_t srow = (size_t)n * T;

    // ---- LDS-DMA (inline asm; ordering: comment above BwdKGeom): stash gather [q][env lane/4][units 4 (lane%4) ..]
    typedef __attribute__((address_space(3))) float lds_f;
    const int e_d = lane >> 2, g4 = lane & 3;
    const size_t drow = (size_t)min(n0 + e_d, N - 1) * T;
    const unsigned ring_base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((lds_f*)(ring + w * SLOT)));
    const unsigned small_base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((lds_f*)small));
    auto issue = [&](int t) {
        const float* src = stash + (drow + t) * (6 * H) + 16 * w + 4 * g4;
        unsigned m0save;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
            "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
            "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
            "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
            "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, off\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(m0save)
            : "v"(src), "v"(src + H), "v"(src + 2 * H), "v"(src + 3 * H), "v"(src + 4 * H), "s"(ring_base)
            : "memory", "scc");
    };
    // waves 0 / 1: one 256-B piece each of the step's small image  [dheads(16 envs x NH, env-major) | keep(16)].
    // The lane's element of step 0 and its per-step stride are fixed: only `+ t * stride` is left in the loop
    // (the division by NH used to run every step on the two waves everybody waits for).
    const float* small_at0;
    unsigned small_stride;
    {
        const int e = w * 64 + lane;
        if (e < 16 * NH) { small_at0 = dheads + (size_t)min(n0 + e / NH, N - 1) * T * NH + e % NH; small_stride = (unsigned)NH; }
        else if (e >= 112 && keep) { small_at0 = keep + (size_t)min(n0 + e - 112, N - 1) * T; small_stride = 1u; }
        else { small_at0 = w_hh + (lane & 15); small_stride = 0u; }         // padding: any readable dwords
    }
    auto issue_small = [&](int t, int slot) {
        const float* src = small_at0 + (size_t)((unsigned)t * small_stride);
        const unsigned dst = small_base + (unsigned)((slot * SMALL + w * 64) * 4);
        unsigned m0save;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0save) : "v"(src), "s"(dst) : "memory");
    };
    const bool small_wave = w < 2;
    // vector-memory operations of this wave, in issue order (all retire in order):
    //   prologue  ring(T-1) [small(T-1)] [small(T-2)]
    //   step t    wait . ring(t-1) . 4 dG stores . [small(t-2)]
    // so at the top of step t everything but the 4 stores (and one small piece) of step t+1 must have landed
    issue(T - 1);
    if (small_wave) {
        issue_small(T - 1, (T - 1) & 1);
        issue_small(T >= 2 ? T - 2 : 0, (T - 2) & 1);
    }
    bool first = true;
    X6_PROF_DECL;
    for (int t = T - 1; t >= 0; --t) {
        X6_PROF_MARK(0);
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (small_wave) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        first = false;
        const float* sl = ring + w * SLOT + j * 16 + 4 * kq;
        float4 pf[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) pf[q] = *reinterpret_cast<const float4*>(sl + q * 256);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the slot is in registers: refill it with step t-1
        issue(t >= 1 ? t - 1 : 0);
        X6_PROF_MARK(1);
        lds_barrier();           // b2: the partials of step t+1 have been summed by everyone; small image of step t landed
        X6_PROF_MARK(2);
        const float* sm = small + (t & 1) * SMALL;
        f32x4 dyacc = {0.f, 0.f, 0.f, 0.f};
        {
            const float d0 = (kq < NH) ? sm[j * NH + kq] : 0.f;
            const float d1 = (4 + kq < NH) ? sm[j * NH + 4 + kq] : 0.f;
            dyacc = __builtin_amdgcn_mfma_f32_16x16x4f32(whb[0], d0, dyacc, 0, 0, 0);
            dyacc = __builtin_amdgcn_mfma_f32_16x16x4f32(whb[1], d1, dyacc, 0, 0, 0);
        }
        const float kp = keep ? sm[112 + j] : 1.f;                           // keep[env j][t]
        const float gi[4] = {pf[0].x, pf[0].y, pf[0].z, pf[0].w}, gf[4] = {pf[1].x, pf[1].y, pf[1].z, pf[1].w};
        const float gg[4] = {pf[2].x, pf[2].y, pf[2].z, pf[2].w}, go[4] = {pf[3].x, pf[3].y, pf[3].z, pf[3].w};
        const float cp[4] = {pf[4].x, pf[4].y, pf[4].z, pf[4].w};
        float dg[4][4];                                                      // [gate][unit r]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dh = dyacc[r] + dh_rec[r];
            const float c = gf[r] * cp[r] + gi[r] * gg[r];
            const float tch = tanhf_(c);
            const float dc = dh * go[r] * (1.0f - tch * tch) + dc_next[r];
            dg[0][r] = dc * gg[r] * gi[r] * (1.0f - gi[r]);
            dg[1][r] = dc * cp[r] * gf[r] * (1.0f - gf[r]);
            dg[2][r] = dc * gi[r] * (1.0f - gg[r] * gg[r]);
            dg[3][r] = dh * tch * go[r] * (1.0f - go[r]);
            dc_next[r] = dc * gf[r] * kp;
        }
        // Gate gradients span many binades, fp16 does not: each env's 64 values of this wave are scaled by a power of two
        // that puts their largest magnitude in [2^13, 2^14) (exact), and the env's column of the partial tiles -- it sits
        // in these same lanes -- is scaled back.  max over the lane's 16 values, then over the four kq rows of env j with
        // the two gfx950 row-swap instructions.
        float mx = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, fabsf(dg[q][r]));
        {
            const unsigned u = __builtin_bit_cast(unsigned, mx);
            const auto s16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
            mx = fmaxf(__builtin_bit_cast(float, (unsigned)s16[0]), __builtin_bit_cast(float, (unsigned)s16[1]));
            const unsigned v = __builtin_bit_cast(unsigned, mx);
            const auto s32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
            mx = fmaxf(__builtin_bit_cast(float, (unsigned)s32[0]), __builtin_bit_cast(float, (unsigned)s32[1]));
        }
        int ex = 14 - __builtin_amdgcn_frexp_expf(mx);                      // mx = f 2^e, f in [0.5, 1)  ->  mx 2^ex in [2^13, 2^14)
        ex = mx > 0.f ? min(max(ex, -100), 100) : 0;
        const float unscale = __builtin_amdgcn_ldexpf(1.0f, -ex);
        // this wave's 64 gate gradients as B fragments: slab sb = gates 2sb, 2sb+1; element e = (gate 2sb + e/4, unit e%4)
        f16x8 bp[2][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                _Float16 p0, p1;
                split2h(__builtin_amdgcn_ldexpf(dg[q][r], ex), p0, p1);
                bp[q >> 1][0][4 * (q & 1) + r] = p0; bp[q >> 1][1][4 * (q & 1) + r] = p1;
            }
        }
        X6_PROF_DEP(bp[1][1]); X6_PROF_DEP(bp[0][0]);
        X6_PROF_MARK(3);
        // partial dh^T tiles: D_m[unit 16m + 4kq + r][env j] over this wave's K range
#pragma unroll
        for (int m = 0; m < NW; ++m) {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[m][sb][1], bp[sb][0], a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[m][sb][0], bp[sb][0], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[m][sb][0], bp[sb][1], a1, 0, 0, 0);
            }
            part[(m * NW + w) * 64 + lane] = (a0 + a1 * H3_LO) * unscale;
        }
        // the step's four dG stores behind the products (same place in the wave's issue order as before: between the ring DMA
        // and the small piece, so the counted waits are unchanged): they queued in front of the products for the slowest
        // wave, which everybody then waited for at b1; here they drain under the partial-sum reduce
        if (live) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(dgates + (srow + t) * (4 * H) + q * H + uo) = float4{dg[q][0], dg[q][1], dg[q][2], dg[q][3]};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        X6_PROF_MARK(4);
        lds_barrier();           // b1: all partials written; nobody reads the small image of step t any more
        X6_PROF_MARK(5);
        if (small_wave) issue_small(t >= 2 ? t - 2 : 0, t & 1);
        f32x4 sum = part[(w * NW) * 64 + lane];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) sum += part[(w * NW + ww) * 64 + lane];      // fixed order: deterministic
#pragma unroll
        for (int r = 0; r < 4; ++r) dh_rec[r] = sum[r] * kp;
        X6_PROF_DEP(dh_rec[3]);
        X6_PROF_MARK(6);
    }
    X6_PROF_FLUSH8();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // retire the clamped tail DMAs before the LDS is released
    if (live) {
        if (dh0) *reinterpret_cast<float4*>(dh0 + (size_t)n * H + uo) = float4{dh_rec[0], dh_rec[1], dh_rec[2], dh_rec[3]};
        if (dc0) *reinterpret_cast<float4*>(dc0 + (size_t)n * H + uo) = float4{dc_next[0], dc_next[1], dc_next[2], dc_next[3]};
    }
}

template <int H>
static int launch_bwd_h3k(const float* keep, const float* stash, const float* w_hh, const float* dheads,
                          const float* w_head, int NH, const float* dhn, const float* dcn, int N, int T, float* dgates,
                          float* dh0, float* dc0, hipStream_t st) {
    const dim3 grid((N + MT - 1) / MT), block(H * 4);
    UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(&lstm_bwd_h3k_kernel<H>), (int)BwdH3Geom<H>::LDS));
    hipLaunchKernelGGL((lstm_bwd_h3k_kernel<H>), grid, block, BwdH3Geom<H>::LDS, st, keep, stash, w_hh, dheads, w_head,
                       NH, dhn, dcn, N, T, dgates, dh0, dc0);
    UAV_LAUNCH_CHECK();
    return 0;
}

template <int H>
static int launch_bwd_x6k(const float* keep, const float* stash, const float* w_hh, const float* dheads,
                          const float* w_head, int NH, const float* dhn, const float* dcn, int N, int T, float* dgates,
                          float* dh0, float* dc0, hipStream_t st) {
    const dim3 grid((N + MT - 1) / MT), block(H * 4);
    UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(&lstm_bwd_x6k_kernel<H>), (int)BwdKGeom<H>::LDS));
    hipLaunchKernelGGL((lstm_bwd_x6k_kernel<H>), grid, block, BwdKGeom<H>::LDS, st, keep, stash, w_hh, dheads, w_head,
                       NH, dhn, dcn, N, T, dgates, dh0, dc0);
    UAV_LAUNCH_CHECK();
    return 0;
}

template <int H>
static int launch_fwd(bool fuse, const float* x, const float* keep, const float* h0, const float* c0,
                      const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, int N, int T, int I,
                      float* y, float* hn, float* cn, float* stash, const float* w_head, const float* b_head, int NHD,
                      float* heads, bool* heads_done, hipStream_t st) {
    const dim3 grid((N + MT - 1) / MT), block(H * 4);
    if (!f32_mfma_requested() && !bf16x6_requested()) {   // default: three fp16 piece products on the matrix pipe (f32 accuracy)
        const size_t lx = FwdH3Geom<H>::LDS;
        UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(&lstm_fwd_h3_kernel<H, true>), (int)lx));
        UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(&lstm_fwd_h3_kernel<H, false>), (int)lx));
        if (fuse)
            hipLaunchKernelGGL((lstm_fwd_h3_kernel<H, true>), grid, block, lx, st, x, keep, h0, c0, w_ih, w_hh, b_ih,
                               b_hh, N, T, I, y, hn, cn, stash, w_head, b_head, NHD, heads);
        else
            hipLaunchKernelGGL((lstm_fwd_h3_kernel<H, false>), grid, block, lx, st, x, keep, h0, c0, w_ih, w_hh, b_ih,
                               b_hh, N, T, I, y, hn, cn, stash, w_head, b_head, NHD, heads);
        *heads_done = heads != nullptr;
        UAV_LAUNCH_CHECK();
        return 0;
    }
    if (!f32_mfma_requested()) {         // UAV_LSTM_BF16X6=1: the six-product bf16 split
        const size_t lx = FwdX6Geom<H>::LDS;
        UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(&lstm_fwd_x6_kernel<H, true>), (int)lx));
        UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(&lstm_fwd_x6_kernel<H, false>), (int)lx));
        if (fuse)
            hipLaunchKernelGGL((lstm_fwd_x6_kernel<H, true>), grid, block, lx, st, x, keep, h0, c0, w_ih, w_hh, b_ih,
                               b_hh, N, T, I, y, hn, cn, stash, w_head, b_head, NHD, heads);
        else
            hipLaunchKernelGGL((lstm_fwd_x6_kernel<H, false>), grid, block, lx, st, x, keep, h0, c0, w_ih, w_hh, b_ih,
                               b_hh, N, T, I, y, hn, cn, stash, w_head, b_head, NHD, heads);
        *heads_done = heads != nullptr;
        UAV_LAUNCH_CHECK();
        return 0;
    }
    const size_t lds = FwdGeom<H>::LDS;
    if (fuse)
        hipLaunchKernelGGL((lstm_fwd_kernel<H, true>), grid, block, lds, st, x, keep, h0, c0, w_ih, w_hh, b_ih, b_hh, N,
                           T, I, y, hn, cn, stash);
    else
        hipLaunchKernelGGL((lstm_fwd_kernel<H, false>), grid, block, lds, st, x, keep, h0, c0, w_ih, w_hh, b_ih, b_hh,
                           N, T, I, y, hn, cn, stash);
    UAV_LAUNCH_CHECK();
    return 0;
}

template <int H>
static int launch_bwd(const float* keep, const float* stash, const float* w_hh, const float* dy, const float* dheads,
                      const float* w_head, int NH, const float* dhn, const float* dcn, int N, int T, float* dgates,
                      float* dh0, float* dc0, hipStream_t st) {
    const dim3 grid((N + MT - 1) / MT), block(H * 4);
    UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(&lstm_bwd_kernel<H>), (int)BwdGeom<H>::LDS));
    hipLaunchKernelGGL((lstm_bwd_kernel<H>), grid, block, BwdGeom<H>::LDS, st, keep, stash, w_hh, dy, dheads, w_head, NH,
                       dhn, dcn, N, T, dgates, dh0, dc0);
    UAV_LAUNCH_CHECK();
    return 0;
}

// b[i] = a0[i] + a1[i]
__global__ void add2_kernel(const float* a0, const float* a1, float* b, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) b[i] = a0[i] + a1[i];
}

static int lstm_bwd_seq(const float* keep, const float* stash, const float* w_hh, const float* dy,
                        const float* dheads, const float* w_head, int NH, const float* dhn, const float* dcn, int N,
                        int T, int H, float* dgates, float* dh0, float* dc0, hipStream_t st) {
    if (dheads && NH <= 7 && !f32_mfma_requested() && !bf16x6_requested()) {   // the PPO path: split-fp16, K split over waves
        switch (H) {
            case 64: return launch_bwd_h3k<64>(keep, stash, w_hh, dheads, w_head, NH, dhn, dcn, N, T, dgates, dh0, dc0, st);
            case 128: return launch_bwd_h3k<128>(keep, stash, w_hh, dheads, w_head, NH, dhn, dcn, N, T, dgates, dh0, dc0, st);
        }
    }
    if (dheads && NH <= 7 && !f32_mfma_requested()) {   // UAV_ARITH_BF16X6: split-bf16 (f32's exponent range), K split over waves
        switch (H) {
            case 64: return launch_bwd_x6k<64>(keep, stash, w_hh, dheads, w_head, NH, dhn, dcn, N, T, dgates, dh0, dc0, st);
            case 128: return launch_bwd_x6k<128>(keep, stash, w_hh, dheads, w_head, NH, dhn, dcn, N, T, dgates, dh0, dc0, st);
        }
    }
    // exact-f32 MFMA (UAV_ARITH_F32_MFMA), plain dy (stacked layers), more than 7 heads
    switch (H) {
        case 64: return launch_bwd<64>(keep, stash, w_hh, dy, dheads, w_head, NH, dhn, dcn, N, T, dgates, dh0, dc0, st);
        case 128: return launch_bwd<128>(keep, stash, w_hh, dy, dheads, w_head, NH, dhn, dcn, N, T, dgates, dh0, dc0, st);
    }
    uav_set_error("uav_lstm_bwd: H=%d unsupported (64, 128)", H);
    return 2;
}

extern "C" {

int uav_lstm_fwd(uav_ctx* ctx, const float* x, const float* keep, const float* h0, const float* c0, const float* w_ih,
                 const float* w_hh, const float* b_ih, const float* b_hh, int N, int T, int I, int H, float* y,
                 float* hn, float* cn, float* stash, const float* w_head, const float* b_head, int n_heads, float* heads,
                 uav_stream stream) {
    UAV_REQUIRE(ctx && x && h0 && c0 && w_ih && w_hh && b_ih && b_hh && y && hn && cn, "uav_lstm_fwd: NULL argument");
    uav_enter(ctx);
    UAV_REQUIRE(N > 0 && T > 0 && I > 0, "uav_lstm_fwd: N=%d T=%d I=%d", N, T, I);
    UAV_REQUIRE(!heads || (w_head && b_head && n_heads > 0 && n_heads <= 8), "uav_lstm_fwd: heads needs w_head, b_head, 1..8 heads");
    hipStream_t st = as_stream(stream);
    const bool persistent = (H == 64 || H == 128);
    const bool fuse = I <= 8 && persistent;
    const bool step256 = lstm_h3_step_path(H);      // h = 256: one fused launch per step, input projection inside
    if (step256) UAV_REQUIRE(stash, "uav_lstm_fwd: stash is required when H is not 64/128");
    if (!fuse && !step256) {
        // time-batched input projection into the gates slot of the stash: pre = x W_ih^T + (b_ih + b_hh)
        UAV_REQUIRE(stash, "uav_lstm_fwd: stash is required when I > 8 or H is not 64/128");
        float* bsum = (float*)ctx->ws;   // 4H floats at the head of the workspace... kept clear of GEMM slabs below
        uav_ctx sub = *ctx;
        sub.ws = (char*)ctx->ws + 65536;
        sub.ws_bytes = ctx->ws_bytes - 65536;
        hipLaunchKernelGGL(add2_kernel, dim3((4 * H + 255) / 256), dim3(256), 0, st, b_ih, b_hh, bsum, 4 * H);
        int rc = gemm_f32(&sub, (int64_t)N * T, 4 * H, I, x, I, 1, w_ih, 1, I, stash, 6 * H, bsum, 0, st);
        if (rc) return rc;
    }
    bool heads_done = false;
    int rc;
    switch (H) {
        case 64: rc = launch_fwd<64>(fuse, x, keep, h0, c0, w_ih, w_hh, b_ih, b_hh, N, T, I, y, hn, cn, stash, w_head, b_head, n_heads, heads, &heads_done, st); break;
        case 128: rc = launch_fwd<128>(fuse, x, keep, h0, c0, w_ih, w_hh, b_ih, b_hh, N, T, I, y, hn, cn, stash, w_head, b_head, n_heads, heads, &heads_done, st); break;
        default:  // any other hidden size: one launch per time step (lstm_generic.hip)
            rc = step256 ? lstm_h3_fwd(ctx, x, I, w_ih, b_ih, b_hh, keep, h0, c0, w_hh, N, T, y, hn, cn, stash, st)
                         : lstm_generic_fwd(ctx, keep, h0, c0, w_hh, N, T, H, y, hn, cn, stash, st);
    }
    if (rc) return rc;
    // kernels without the fused head product: heads = y W_head^T + b_head as one GEMM over the rows of y
    if (heads && !heads_done)
        return gemm_f32(ctx, (int64_t)N * T, n_heads, H, y, H, 1, w_head, 1, H, heads, n_heads, b_head, 0, st);
    return 0;
}

int uav_lstm_bwd_caps(uav_ctx* ctx, int I, int H) {
    if (!ctx) return 0;
    if (H == 64 || H == 128) return UAV_BWD_TAKES_DHEADS;            // the persistent sequence kernels
    uav_enter(ctx);
    return lstm_generic_bwd_caps(I, H);
}

size_t uav_lstm_dgates_bytes(uav_ctx* ctx, int N, int T, int H) {
    if (!ctx || N <= 0 || T <= 0 || H <= 0) return 0;
    uav_enter(ctx);
    if (H == DgPack::H && lstm_h3_dg_packed(H)) return DgPack(N, T).bytes();
    return (size_t)N * T * 4 * H * sizeof(float);
}

int uav_lstm_dgates_f32(uav_ctx* ctx, const float* dgates, int N, int T, int H, float* out, uav_stream stream) {
    UAV_REQUIRE(ctx && dgates && out && N > 0 && T > 0 && H > 0, "uav_lstm_dgates_f32: bad argument");
    uav_enter(ctx);
    const int form = H == DgPack::H ? uav_dg_form(ctx, dgates) : 0;          // as recorded by the call that wrote it, else by the mode
    if (form > 0 || (form < 0 && H == DgPack::H && lstm_h3_dg_packed(H))) return lstm_pc_unpack(dgates, N, T, out, as_stream(stream));
    UAV_CHECK_HIP(hipMemcpyAsync(out, dgates, (size_t)N * T * 4 * H * sizeof(float), hipMemcpyDeviceToDevice, as_stream(stream)));
    return 0;
}

int uav_lstm_bwd_stack(uav_ctx* ctx, int n_layers, const uav_lstm_bwd_layer* layers, const float* dy, const float* dheads,
                       const float* w_head, int n_heads, int N, int T, int H, uav_stream stream) {
    UAV_REQUIRE(ctx && layers, "uav_lstm_bwd_stack: NULL argument");
    uav_enter(ctx);
    UAV_REQUIRE((dy != nullptr) != (dheads != nullptr), "uav_lstm_bwd_stack: give exactly one of dy / dheads");
    UAV_REQUIRE(!dheads || (w_head && n_heads > 0 && n_heads <= 8), "uav_lstm_bwd_stack: dheads needs w_head and 1..8 heads");
    UAV_REQUIRE(N > 0 && T > 0, "uav_lstm_bwd_stack: N=%d T=%d", N, T);
    UAV_REQUIRE(H == 256 && lstm_h3_stack_ok(H), "uav_lstm_bwd_stack: H = 256 on the fp16-split arithmetic only (uav_lstm_bwd_caps)");
    return lstm_h3_bwd_stack(ctx, n_layers, layers, dy, dheads, w_head, n_heads, N, T, as_stream(stream));
}

int uav_lstm_bwd(uav_ctx* ctx, const float* keep, const float* stash, const float* w_hh, const float* dy,
                 const float* dheads, const float* w_head, int n_heads, const float* dhn, const float* dcn, int N,
                 int T, int H, float* dgates, float* dh0, float* dc0, const float* w_ih, int I, float* dx,
                 uav_stream stream) {
    UAV_REQUIRE(ctx && stash && w_hh && dgates, "uav_lstm_bwd: NULL argument");
    uav_enter(ctx);
    UAV_REQUIRE((dy != nullptr) != (dheads != nullptr), "uav_lstm_bwd: give exactly one of dy / dheads");
    UAV_REQUIRE(!dheads || (w_head && n_heads > 0 && n_heads <= 8), "uav_lstm_bwd: dheads needs w_head and 1..8 heads");
    UAV_REQUIRE(N > 0 && T > 0, "uav_lstm_bwd: N=%d T=%d", N, T);
    if (H != 64 && H != 128) {
        return lstm_generic_bwd(ctx, keep, stash, w_hh, dy, dheads, w_head, n_heads, dhn, dcn, N, T, H, dgates, dh0, dc0, w_ih, I, dx,
                                as_stream(stream));
    }
    UAV_REQUIRE(!dx, "uav_lstm_bwd: dx is not formed by the H = 64 / 128 sequence kernels (uav_lstm_wgrad does it)");
    return lstm_bwd_seq(keep, stash, w_hh, dy, dheads, w_head, n_heads, dhn, dcn, N, T, H, dgates, dh0, dc0,
                        as_stream(stream));
}

int uav_lstm_wgrad(uav_ctx* ctx, const float* x, const float* keep, const float* h0, const float* y,
                   const float* stash, const float* dgates, const float* w_ih, const float* dheads, int n_heads, int N,
                   int T, int I, int H, float* dw_ih, float* dw_hh, float* db, float* db_hh, float* dw_head, float* dx,
                   uav_stream stream) {
    UAV_REQUIRE(ctx && x && h0 && y && dgates && w_ih && dw_ih && dw_hh && db, "uav_lstm_wgrad: NULL argument");
    uav_enter(ctx);
    UAV_REQUIRE(N > 0 && T > 0 && I > 0 && H > 0, "uav_lstm_wgrad: N=%d T=%d I=%d H=%d", N, T, I, H);
    UAV_REQUIRE(!dheads || (dw_head && n_heads > 0 && n_heads <= 8), "uav_lstm_wgrad: dheads needs dw_head, 1..8 heads");
    hipStream_t st = as_stream(stream);
    const int64_t NT = (int64_t)N * T;
    int rc;
    if (I <= 6 && (H == 64 || H == 128)) {
        if ((rc = lstm_wgrad_fused(ctx, dgates, y, keep, h0, x, I, dheads ? y : nullptr, dheads, n_heads, N, T, H, dw_ih,
                                   dw_hh, db, db_hh, dw_head, st))) return rc;
    } else {
        // generic path: column sums use the tail of the workspace, the split-K slabs everything in front of it
        UAV_REQUIRE(stash, "uav_lstm_wgrad: stash is required when I > 6");
        const size_t red_floats = (size_t)1024 * 9 * 4 * H + 64;      // colsum (+ dG^T x) partials + the word for max |dG|
        UAV_REQUIRE(ctx->ws_bytes >= red_floats * sizeof(float) * 2, "uav_lstm_wgrad: workspace too small");
        float* red = (float*)((char*)ctx->ws + ctx->ws_bytes) - red_floats;
        uav_ctx sub = *ctx;
        sub.ws_bytes = ctx->ws_bytes - red_floats * sizeof(float);
        // h = 256 on the fp16-split step path: `dgates` is the BPTT's piece chunks (common.h: DgPack), read as they are by
        // wgrad_pc.hip -- db, dw_hh and dw_ih for a narrow (I <= 8) or hidden-wide (I = 256) input.  Any other request (another
        // width, dx) first unpacks to f32 rows in the workspace and takes the products below.
        bool done = false;
        const float* hprev = stash + 5 * H;                 // rows of h_prev: the stash's slot, stride 6H
        int64_t ld_hprev = 6 * H;
        {
            const int form = uav_dg_form(ctx, dgates), want = lstm_h3_dg_packed(H) ? 1 : 0;
            UAV_REQUIRE(form < 0 || H != DgPack::H || form == want, "uav_lstm_wgrad: this dgates buffer was written as %s but the handle's "
                        "arithmetic mode / debug flags now read %s (they must not change between uav_lstm_bwd and uav_lstm_wgrad)",
                        form ? "fp16 piece chunks" : "f32 rows", want ? "fp16 piece chunks" : "f32 rows");
        }
        if (lstm_h3_dg_packed(H)) {
            if ((I <= 8 || I == H) && !dx) {
                if ((rc = lstm_pc_wgrad(ctx, x, y, h0, dgates, N, T, I, dw_ih, dw_hh, db, db_hh, st))) return rc;
                done = true;
            } else {
                const size_t f32_bytes = (size_t)NT * 5 * H * sizeof(float);       // gate gradients 4H + h_prev H (this mode's forward
                UAV_REQUIRE(sub.ws_bytes >= f32_bytes + (64u << 20), "uav_lstm_wgrad (h=256, I=%d%s): needs %zu bytes of workspace for "
                            "the gate gradients and h_prev as f32 rows", I, dx ? ", dx" : "", f32_bytes + (64u << 20));   // writes no slot)
                float* rows = (float*)((char*)sub.ws + sub.ws_bytes - f32_bytes);
                sub.ws_bytes -= f32_bytes;
                if ((rc = lstm_pc_unpack(dgates, N, T, rows, st))) return rc;
                float* hp = rows + (size_t)NT * 4 * H;
                if ((rc = lstm_pc_hprev_rows(y, keep, h0, N, T, hp, st))) return rc;
                dgates = rows;
                hprev = hp;
                ld_hprev = H;
            }
        }
        // The large products go to the 16-bit matrix pipe as three fp16 piece products (gemm_h3.hip) unless exact f32 or
        // the bf16 split was asked for: dG is block-scaled by one power of two from its absolute maximum, which the bias
        // gradient's column-sum pass (it reads all of dG anyway) delivers; h_prev, x and W_ih are inside fp16's range
        // under the same preconditions as the sequence kernels' (include/uavppo.h, uav_set_lstm_arith).
        const bool h3 = !uav_want_f32_mfma() && !uav_want_bf16x6() && (reinterpret_cast<uintptr_t>(dgates) & 15) == 0;
        unsigned* amax = h3 ? reinterpret_cast<unsigned*>(red + (size_t)1024 * 9 * 4 * H) : nullptr;
        auto product = [&](int64_t M, int64_t Nn, int64_t K, const float* A, int64_t sa_m, int64_t sa_k, const float* B,
                           int64_t sb_k, int64_t sb_n, float* C, int64_t ldc) {
            if (h3 && gemm_h3_ok(M, Nn, K, A, sa_m, sa_k, B, sb_k, sb_n))
                return gemm_h3(&sub, M, Nn, K, A, sa_m, sa_k, B, sb_k, sb_n, C, ldc, nullptr, 0, amax, st);
            return gemm_f32(&sub, M, Nn, K, A, sa_m, sa_k, B, sb_k, sb_n, C, ldc, nullptr, 0, st);
        };
        if (!done) {
            // a narrow input (layer 1: obs + trend, I <= 8): dW_ih = dG^T x rides on the bias gradient's pass over dG
            const bool narrow = I <= 8 && (reinterpret_cast<uintptr_t>(dgates) & 15) == 0;
            if (narrow) rc = colsum_xw(&sub, dgates, NT, 4 * H, x, I, db, dw_ih, red, amax, st);
            else rc = colsum_absmax(&sub, dgates, NT, 4 * H, db, red, amax, st);
            if (rc) return rc;
            if (db_hh) UAV_CHECK_HIP(hipMemcpyAsync(db_hh, db, (size_t)4 * H * sizeof(float), hipMemcpyDeviceToDevice, st));
            if ((rc = product(4 * H, H, NT, dgates, 1, 4 * H, hprev, ld_hprev, 1, dw_hh, H))) return rc;
            if (!narrow && (rc = product(4 * H, I, NT, dgates, 1, 4 * H, x, I, 1, dw_ih, I))) return rc;
        }
        if (dheads) {        // dW_head = dheads^T y [n_heads][H]: a stream over y with the few dheads columns riding along
            if (H % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0)
                rc = colsum_xw(&sub, y, NT, H, dheads, n_heads, nullptr, dw_head, red, nullptr, st, 1);
            else
                rc = gemm_f32(&sub, n_heads, H, NT, dheads, 1, n_heads, y, H, 1, dw_head, H, nullptr, 0, st);
            if (rc) return rc;
        }
        if (dx) return product(NT, I, 4 * H, dgates, 4 * H, 1, w_ih, I, 1, dx, I);
        return 0;
    }
    if (dx) return gemm_f32(ctx, NT, I, 4 * H, dgates, 4 * H, 1, w_ih, I, 1, dx, I, nullptr, 0, st);
    return 0;
}

}  // extern "C"

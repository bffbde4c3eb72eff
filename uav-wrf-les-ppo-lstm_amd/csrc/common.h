// common.h -- shared device/host helpers for libuavppo (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/uavppo.h"

struct uav_ctx {
    int device;
    int num_cu;
    void* ws;          // scratch for two-stage reductions / split-K slabs
    size_t ws_bytes;
    double* pow075;    // device table pow(i, 0.75), i = 0..5000 (env_core.h)
    double* wave;      // device tables sin(0.05 x) | cos(0.07 y), x, y = 0..499 (env_core.h field_at)
    double* ftab;      // device tables of the table-driven f64 log / cos-sin / exp (env_core.h ft_*)
    int lstm_arith;    // UAV_ARITH_*: how the LSTM sequence kernels evaluate their f32 matrix products (uav_set_lstm_arith)
    unsigned debug;    // UAV_DEBUG_*: A/B switches of the h = 256 step path (uav_set_debug_flags)
    hipStream_t side[3];      // uav_lstm_bwd_stack: one stream per layer below the top (created on first use)
    hipEvent_t side_ev[8];    // fork / join + a small ring of per-step hand-off events per side stream
    void* comm;               // ncclComm_t of this process's rank (comm.hip: uav_comm_init), or NULL
    int comm_rank, comm_world;
    // form of the last few gate-gradient buffers uav_lstm_bwd / _bwd_stack wrote (1 = fp16 piece chunks, 0 = f32 rows): uav_lstm_wgrad
    // refuses a buffer whose recorded form is not the one the handle's CURRENT mode would read (mode changed in between)
    struct { const void* p; int packed; } dg_form[8];
    int dg_next;
};
static inline void uav_dg_record(uav_ctx* ctx, const void* p, int packed) {
    for (auto& e : ctx->dg_form)
        if (e.p == p) { e.packed = packed; return; }
    ctx->dg_form[ctx->dg_next] = {p, packed};
    ctx->dg_next = (ctx->dg_next + 1) % 8;
}
static inline int uav_dg_form(const uav_ctx* ctx, const void* p) {       // -1 = not recorded (a buffer filled by the caller)
    for (auto& e : ctx->dg_form)
        if (e.p == p) return e.packed;
    return -1;
}

// arithmetic / debug switches of the call in flight on this thread (set from the handle by the LSTM entry points; the
// launch helpers below them have no ctx argument).  No getenv on any call path: the UAV_LSTM_BF16X6 / UAV_LSTM_F32_MFMA
// environment variables are read ONCE, by uav_create, as the handle's initial mode.
extern thread_local int g_uav_arith;
extern thread_local unsigned g_uav_debug;
static inline bool uav_want_f32_mfma() { return g_uav_arith == UAV_ARITH_F32_MFMA; }
static inline bool uav_want_bf16x6() { return g_uav_arith == UAV_ARITH_BF16X6; }
static inline bool uav_debug(unsigned bit) { return (g_uav_debug & bit) != 0; }
static inline void uav_enter(const struct uav_ctx* ctx);

static inline void uav_enter(const uav_ctx* ctx) {
    g_uav_arith = ctx->lstm_arith;
    g_uav_debug = ctx->debug;
}

void uav_set_error(const char* fmt, ...);

// Opt a kernel in to `bytes` of dynamic LDS.  Function attributes are PER DEVICE, so the "already done" memo is keyed by
// (current device, kernel): a process that drives several devices through several handles gets the attribute on each.
hipError_t uav_dyn_lds(const void* kernel, int bytes);

#define UAV_CHECK_HIP(expr)                                                          \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) {                                                      \
            uav_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return 1;                                                                \
        }                                                                            \
    } while (0)

#define UAV_REQUIRE(cond, ...)                                                       \
    do {                                                                             \
        if (!(cond)) {                                                               \
            uav_set_error(__VA_ARGS__);                                              \
            return 2;                                                                \
        }                                                                            \
    } while (0)

#define UAV_LAUNCH_CHECK() UAV_CHECK_HIP(hipGetLastError())

static inline hipStream_t as_stream(uav_stream s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int WAVE = 64;

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() also waits vmcnt(0): in the
// persistent kernels that drained the software-prefetched global loads and every outstanding
// store at each time step (lstm_bwd: 0.84 -> 0.6 ms).  Waves of one workgroup never exchange data
// through global memory inside these kernels, so LDS ordering is all the barrier has to give.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// LSTM gate activations: v_exp_f32 + v_rcp_f32 (1 ulp each).  A plain `1.0f / x` compiles to the
// ~10-instruction IEEE division sequence; with 20 activations per lane per step that sequence,
// not the MFMAs, paced the persistent kernels (tools/mfma_probe.hip).
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) {
    // |x| >= 1/4: tanh(x) = 1 - 2 / (exp(2x) + 1): exact limits at +-inf, absolute error <= 1.9e-7, i.e. relative <= 8e-7 there.
    // |x| <  1/4: that form cancels (its RELATIVE error is 1.9e-7 / |x|: 1e-3 at |x| = 1e-4), so the odd Taylor polynomial
    // x (1 - x^2/3 + 2 x^4/15 - 17 x^6/315 + 62 x^8/2835) takes over: truncation 8.9e-3 x^10 <= 8.5e-9 relative, rounding
    // ~2e-7 relative.  Both are evaluated and one selected (no branch): +8 VALU per call.  UAV_TANH_LEGACY: the exp form alone
    // (A/B builds only: tools/ab_tanh.sh).
    const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f);
#ifdef UAV_TANH_LEGACY
    return big;
#else
    const float x2 = x * x;
    float p = __builtin_fmaf(x2, 62.0f / 2835.0f, -17.0f / 315.0f);
    p = __builtin_fmaf(p, x2, 2.0f / 15.0f);
    p = __builtin_fmaf(p, x2, -1.0f / 3.0f);
    p = __builtin_fmaf(p, x2, 1.0f);
    return __builtin_fabsf(x) < 0.25f ? x * p : big;
#endif
}

// ---- split-bf16 ("x6") arithmetic: an f32 value as three bf16 pieces a = p0 + p1 + p2 (8 significand bits each, so
// the split is exact); products of two split operands keep the six piece products with i + j <= 2 and accumulate them
// in f32 (v_mfma_f32_16x16x32_bf16).  Error and rate: tools/bf16x6_probe.hip; the argument: lstm.hip, lstm_fwd_x6_kernel.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split3(float a, __bf16& p0, __bf16& p1, __bf16& p2) {
    p0 = (__bf16)a;
    const float r1 = a - (float)p0;
    p1 = (__bf16)r1;
    p2 = (__bf16)(r1 - (float)p1);
}
__device__ __forceinline__ unsigned short bf_bits(__bf16 v) { return __builtin_bit_cast(unsigned short, v); }

// ---- split-fp16 ("h3") arithmetic: an f32 value as TWO fp16 pieces, a = p0 + 2^-11 p1 with p0 = fp16(a) and
// p1 = fp16((a - p0) * 2^11): the residual is scaled back into fp16's normal range, so a is carried to 2^-24 |a| (its last
// f32 bit).  a b = p0 q0 + 2^-11 (p0 q1 + p1 q0) + O(2^-24 |a b|): the main product and the two cross products go to two
// f32 accumulators (v_mfma_f32_16x16x32_f16), combined once per dot product as acc0 + 2^-11 acc1.  Three products
// instead of the bf16 split's six, two pieces instead of three (the weights keep their f32 register footprint), and on a
// K=128 dot product a SMALLER error against f64 than either the bf16 split or the exact-f32 MFMA chain (rms 1.06e-7 vs
// 1.50e-7 vs 2.05e-7 of the terms' rms sum: fewer accumulator roundings; tools/f16x3_probe.hip).  fp16's range is the
// caller's business: |a| < 65504, and operands with a wide dynamic range (gradients) are block-scaled by a power of two.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float H3_LO = 1.0f / 2048.0f;          // weight of the cross-product accumulator
__device__ __forceinline__ void split2h(float a, _Float16& p0, _Float16& p1) {
    p0 = (_Float16)a;
    p1 = (_Float16)((a - (float)p0) * 2048.0f);
}
__device__ __forceinline__ unsigned short h_bits(_Float16 v) { return __builtin_bit_cast(unsigned short, v); }

// ---- gate gradients of the h = 256 fp16-split step path, stored ONCE (round 5): the `dgates` buffer that travels from
// uav_lstm_bwd / _bwd_stack to uav_lstm_wgrad holds what the BPTT's recurrent product consumes -- the two fp16 pieces of
// every row, scaled per (env, step) by the power of two that puts the row's largest magnitude in [2^13, 2^14), in MFMA
// fragment order (lstm_generic.hip: frag_index) -- and the weight-gradient pass reads THAT (wgrad_pc.hip) instead of a second,
// f32 copy (8 of the 17 KB the gate-gradient kernel moved per env-step; -10 % of a C5 iteration by ablation,
// profiles/r05_c5_no_f32_dgates_ablation.log).  Layout, time-major so that a step's block and the GEMM's K walk are linear:
//   halves  pieces[t][rt][s][piece][512]      rt < RT = NP / 16 (16-env row tiles, NP = N rounded up to 64; rows >= N zero),
//                                             s < NS = 4H / 32 (32 gate rows), 512 = (kq * 16 + env) * 8 + i, row 32 s + 8 kq + i
//   floats  isc[t][NP]                        2^-e of (env, step): dG = (p0 + 2^-11 p1) * isc; 0 for rows >= N
//   floats  iscm[t][NP]                       isc * keep[env][step]: the scale under which dW_hh = dG^T h_prev takes h_prev[n][t] = y[n][t-1]
//                                             * keep[n][t] straight from the layer's output y (h0 at t = 0) -- the forward pass then
//                                             need not write an h_prev slot into the stash at all
// The same byte count as f32 [N][T][4H] plus 8 bytes per (env, step): uav_lstm_dgates_bytes.
struct DgPack {
    static constexpr int H = 256, NS = 4 * H / 32;
    int N, T, NP, RT;
    __host__ __device__ DgPack(int n, int t) : N(n), T(t), NP((n + 63) / 64 * 64), RT(NP / 16) {}
    __host__ __device__ size_t step_halves() const { return (size_t)RT * NS * 1024; }
    __host__ __device__ size_t piece_bytes() const { return (size_t)T * step_halves() * 2; }
    __host__ __device__ size_t bytes() const { return piece_bytes() + 2 * (size_t)T * NP * sizeof(float); }
    __host__ __device__ unsigned short* pieces(void* base, int t) const { return (unsigned short*)base + (size_t)t * step_halves(); }
    __host__ __device__ const unsigned short* pieces(const void* base, int t) const { return (const unsigned short*)base + (size_t)t * step_halves(); }
    __host__ __device__ float* isc(void* base, int t) const { return (float*)((char*)base + piece_bytes()) + (size_t)t * NP; }
    __host__ __device__ const float* isc(const void* base, int t) const { return (const float*)((const char*)base + piece_bytes()) + (size_t)t * NP; }
    __host__ __device__ float* iscm(void* base, int t) const { return isc(base, T) + (size_t)t * NP; }
    __host__ __device__ const float* iscm(const void* base, int t) const { return isc(base, T) + (size_t)t * NP; }
};

// ---- wave / block reductions (deterministic: fixed tree, no atomics) -------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;   // valid in lane 0
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_allsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over a 256-thread block; result valid in thread 0. `sm` has >= 4 slots.
template <typename T>
__device__ __forceinline__ T block256_sum(T v, T* sm) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sm[w] = v;
    __syncthreads();
    T r = T(0);
    if (threadIdx.x == 0) r = ((sm[0] + sm[1]) + sm[2]) + sm[3];
    __syncthreads();
    return r;
}

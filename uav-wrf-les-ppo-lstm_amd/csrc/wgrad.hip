// wgrad.hip -- fused LSTM weight-gradient kernel (the time-batched "dW = dG^T [Hprev | X | 1]" GEMM).
//
// One pass over the per-step gate gradients dgates[N*T][4H] (written by lstm_bwd_kernel) produces
//   dW_hh[4H][H] = dG^T Hprev,  dW_ih[4H][I] = dG^T X,  db[4H] = dG^T 1        (I <= 6, ones column)
// and, for the top layer, dW_head[NH][H] = dheads^T Y -- instead of four generic split-K GEMMs and a
// column-sum that each re-read dgates.  MFMA-bound (exact f32, v_mfma_f32_16x16x4_f32):
// 2*4H*(H+16) + 2*16*H flop per (n,t) row; HBM: 4H + 2H + 8 floats read per row, once.
//
// Decomposition: K (= the N*T rows) is split over the grid, one contiguous row range per
// workgroup; wave w owns gate rows [64w, 64w+64) x all H+16 columns (4 x (H/16+1) accumulator
// tiles) plus one 16 x 16 tile of dW_head.  Row chunks of 16 are register-staged and written to a
// double-buffered LDS image whose row strides are = 16 (mod 32) floats so both "transposed"
// fragment reads (lane (i, kq) reads element i of row 4s+kq) are conflict-free ds_read_b32.
// Each workgroup writes its partial slab to the context workspace; a second kernel sums the slabs
// in a fixed order (deterministic, no float atomics).
#include <stdlib.h>
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 16;          // rows per staged chunk (4 MFMA k-steps)

template <int H>
struct WG {
    static constexpr int NW = H / 16;                 // waves (each owns 64 gate rows)
    static constexpr int NT_ = H / 16 + 1;            // 16-column tiles of [Hprev | X,1,pad]
    static constexpr int NC = H + 16;                 // columns of the B image
    static constexpr int SG = 4 * H + 16;             // LDS row strides, all = 16 (mod 32)
    static constexpr int SB = H + 16;
    static constexpr int SY = H + 16;
    static constexpr int SD = 48;
    static constexpr int BUF = KC * (SG + SB + SY + SD);          // floats per LDS buffer
    static constexpr size_t LDS = 2 * BUF * sizeof(float);
    static constexpr size_t SLAB = (size_t)4 * H * NC + 16 * H;   // floats per workgroup partial
};

template <int H>
__global__ __launch_bounds__(H * 4) void lstm_wgrad_kernel(
    const float* __restrict__ dgates, const float* __restrict__ y_prev_src, const float* __restrict__ keep,
    const float* __restrict__ h0, const float* __restrict__ x, int I, const float* __restrict__ ytop,
    const float* __restrict__ dheads, int NH, int N, int T, int64_t rows_per_block, float* __restrict__ slabs) {
    using G = WG<H>;
    constexpr int NT_ = G::NT_, SG = G::SG, SB = G::SB, SY = G::SY, SD = G::SD, BUF = G::BUF;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int64_t NTr = (int64_t)N * T;
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r_end = (r_begin + rows_per_block < NTr) ? r_begin + rows_per_block : NTr;
    const int nchunk = (r_end > r_begin) ? (int)((r_end - r_begin + KC - 1) / KC) : 0;

    f32x4 acc[4][NT_];
    f32x4 acch = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT_; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- register staging of one chunk --------------------------------------------------------
    constexpr int GV = KC * H / (H * 4);          // float4 of dgates per thread per chunk: KC*4H/4 / (4H threads) = KC/4
    float4 sg[GV];
    float4 sh, sy;
    float sx, sd;
    // per-thread fixed coordinates inside a chunk
    const int g_col4 = tid % H;                   // float4 column of dgates (4H/4 = H float4 per row)
    const int g_row0 = tid / H;                   // 0..3 ; rows g_row0 + 4*i
    const int h_row = tid / (H / 4), h_col4 = tid % (H / 4);     // KC*H/4 = 4H float4 -> exactly one per thread
    const int s_row = tid >> 3, s_f = tid & 7;    // first 128 threads: x / dheads element (row, feature)

    // branch-free chunk load: rows past the end are clamped to a valid row and multiplied by 0
    auto load_chunk = [&](int c) {
        const int64_t base = r_begin + (int64_t)c * KC;
#pragma unroll
        for (int i = 0; i < GV; ++i) {
            const int64_t r = base + g_row0 + 4 * i;
            const float m = (r < r_end) ? 1.f : 0.f;
            const float4 v = *reinterpret_cast<const float4*>(dgates + (r < NTr ? r : NTr - 1) * (4 * H) + 4 * g_col4);
            sg[i] = make_float4(v.x * m, v.y * m, v.z * m, v.w * m);
        }
        {
            const int64_t r = base + h_row, rc = (r < NTr ? r : NTr - 1);
            // h_prev of row (n,t): y[n][t-1] * keep[n][t], or h0[n] * keep[n][0] at t = 0
            const int64_t n = rc / T;
            const int t = (int)(rc - n * T);
            const float m = (r < r_end) ? 1.f : 0.f;
            const float kp = (keep ? keep[rc] : 1.f) * m;
            const float* src = (t == 0) ? (h0 + n * H) : (y_prev_src + (rc - 1) * H);
            const float4 v = *reinterpret_cast<const float4*>(src + 4 * h_col4);
            sh = make_float4(v.x * kp, v.y * kp, v.z * kp, v.w * kp);
            sy = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ytop) {
                const float4 u4 = *reinterpret_cast<const float4*>(ytop + rc * H + 4 * h_col4);
                sy = make_float4(u4.x * m, u4.y * m, u4.z * m, u4.w * m);
            }
        }
        sx = 0.f;
        sd = 0.f;
        if (tid < KC * 8) {
            const int64_t r = base + s_row;
            if (r < r_end) {
                sx = (s_f < I) ? x[r * I + s_f] : (s_f == 6 ? 1.f : 0.f);      // column 6 = ones -> db
                if (dheads && s_f < NH) sd = dheads[r * NH + s_f];
            }
        }
    };
    auto store_chunk = [&](int buf) {
        float* lg = smem + buf * BUF;
        float* lb = lg + KC * SG;
        float* ly = lb + KC * SB;
        float* ld = ly + KC * SY;
#pragma unroll
        for (int i = 0; i < GV; ++i)
            *reinterpret_cast<float4*>(lg + (g_row0 + 4 * i) * SG + 4 * g_col4) = sg[i];
        *reinterpret_cast<float4*>(lb + h_row * SB + 4 * h_col4) = sh;
        *reinterpret_cast<float4*>(ly + h_row * SY + 4 * h_col4) = sy;
        if (tid < KC * 8) {
            lb[s_row * SB + H + s_f] = sx;
            lb[s_row * SB + H + 8 + s_f] = 0.f;
            ld[s_row * SD + s_f] = sd;
            ld[s_row * SD + 8 + s_f] = 0.f;
        }
    };

    if (nchunk > 0) load_chunk(0);
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        store_chunk(buf);
        lds_barrier();
        if (c + 1 < nchunk) load_chunk(c + 1);          // global loads fly under the MFMAs below
        const float* lg = smem + buf * BUF;
        const float* lb = lg + KC * SG;
        const float* ly = lb + KC * SB;
        const float* ld = ly + KC * SY;
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) {
            const int kr = 4 * s + kq;                   // chunk row this lane supplies
            float a[4], b[NT_];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) a[mi] = lg[kr * SG + 64 * w + 16 * mi + j];
#pragma unroll
            for (int ni = 0; ni < NT_; ++ni) b[ni] = lb[kr * SB + 16 * ni + j];
            const float ah = ld[kr * SD + j];
            const float bh = ly[kr * SY + 16 * w + j];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < NT_; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            acch = __builtin_amdgcn_mfma_f32_16x16x4f32(ah, bh, acch, 0, 0, 0);
        }
    }
    // ---- partial slab: [4H][NC] then [16][H]; C/D map row = 4*kq + r, col = j
    float* slab = slabs + (size_t)blockIdx.x * G::SLAB;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT_; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                slab[(size_t)(64 * w + 16 * mi + 4 * kq + r) * G::NC + 16 * ni + j] = acc[mi][ni][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) slab[(size_t)4 * H * G::NC + (size_t)(4 * kq + r) * H + 16 * w + j] = acch[r];
}

// ------------------------------------------------------------------------------ split-bf16 variant
// The same products on the bf16 matrix pipe at f32 accuracy (see lstm.hip, lstm_fwd_x6_kernel): both operands
// are split exactly into three bf16 pieces and the six piece products with i + j <= 2 are accumulated in f32
// by v_mfma_f32_16x16x32_bf16 -- 6 x 16 cycles per 32 rows against 8 x 32 for the exact-f32 chain.
// K slab = 32 rows.  Operands:
//   A = dG^T   each dG element is used by exactly one lane (wave w owns gate rows [64w, 64w+64)), so it never
//              touches LDS: lane (i, kq) loads rows 8kq..8kq+7 straight from HBM, one dwordx2 per tile pair p
//              (gate columns 64w + 32p + 2i, +1: each wave load covers whole 128-B lines; tile 2p+m holds the
//              gate rows 64w + 32p + 2i + m, a row permutation undone when the slab is written), one slab
//              ahead, and splits them in registers;
//   B = [Hprev | x, 1]  and  Y   are needed by every wave: split ONCE by the thread that loaded them (column c,
//              8 consecutive rows: coalesced wave loads, one slab ahead) and parked k-contiguous in LDS piece
//              planes [piece][column][32 rows + pad], read back as conflict-free ds_read_b128 fragments.
// The split is VALU work (~550 instructions per wave and slab) and the two waves of a SIMD would do it at the
// same time, leaving the matrix pipe idle (measured: 7000 of 13700 cycles per slab).  So the wave groups run
// the slab body in rotated order -- EARLY waves: split0 mfma0 commit split1 mfma1, LATE waves: mfma0 commit
// split1 mfma1 split0(next) -- and one group's MFMAs cover the other group's VALU phases.
// keep[] and the t == 0 test are wave-uniform (scalar).  Needs full 32-row slabs, T >= 8, I <= 6.

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& p0, bf16x8& p1, bf16x8& p2) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        __bf16 a, b, c;
        split3(v[i], a, b, c);
        p0[i] = a; p1[i] = b; p2[i] = c;
    }
}

constexpr int KS6 = 32;         // rows per slab

template <int H>
struct WGX {
    static constexpr int NW = H / 16, NT = H * 4;
    static constexpr int NT_ = H / 16 + 1, NC = H + 16;
    static constexpr int KP = KS6 + 8;                               // padded k stride (bf16): conflict-free b128
    static constexpr int BPL = NC * KP, YPL = H * KP, DPL = 16 * KP; // elements per piece plane
    static constexpr int BUF = 3 * (BPL + YPL + DPL);                // bf16 elements per buffer
    static constexpr size_t LDS = 2 * BUF * sizeof(unsigned short);
    static constexpr int XV = 16 * KS6 / NT;                         // x|1 elements per thread per slab
    static constexpr int DV = (8 * KS6 + NT - 1) / NT;               // dheads elements per thread per slab
};

#ifdef UAV_X6_PROFILE
__device__ unsigned long long g_wx6_prof[2][8];
#define WX_PROF_DECL unsigned long long pm_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pl_ = __builtin_readcyclecounter()
#define WX_PROF_MARK(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); pm_[i] += n_ - pl_; pl_ = n_; } while (0)
#define WX_PROF_DEP(v) asm volatile("" ::"v"(v))
#define WX_PROF_FLUSH() do { if (blockIdx.x == 0 && lane == 0 && (w == 0 || w == H / 16 - 1)) \
        for (int i_ = 0; i_ < 8; ++i_) g_wx6_prof[w ? 1 : 0][i_] = pm_[i_]; } while (0)
extern "C" int uav_wx6_prof_read(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wx6_prof), sizeof(g_wx6_prof)) == hipSuccess ? 0 : 1;
}
#else
#define WX_PROF_DECL
#define WX_PROF_MARK(i)
#define WX_PROF_DEP(v)
#define WX_PROF_FLUSH()
#endif

template <int H, bool HEADS>
__global__ __launch_bounds__(H * 4) void lstm_wgrad_x6_kernel(
    const float* __restrict__ dgates, const float* __restrict__ y, const float* __restrict__ keep,
    const float* __restrict__ h0, const float* __restrict__ x, int I, const float* __restrict__ dheads, int NH,
    int N, int T, int rows_per_block, float* __restrict__ slabs) {
    using G = WGX<H>;
    constexpr int NT_ = G::NT_, KP = G::KP, BPL = G::BPL, YPL = G::YPL, DPL = G::DPL, BUF = G::BUF, NT = G::NT;
    constexpr int XV = G::XV, DV = G::DV, NW = G::NW;
    extern __shared__ __attribute__((aligned(16))) unsigned short sm16[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool late = w >= NW / 2;                 // waves w and w + NW/2 share a SIMD
    const int j = lane & 15, kq = lane >> 4;
    const int r_begin = blockIdx.x * rows_per_block;
    const int nslab = rows_per_block / KS6;

    // staging coordinates: column c, row group rg (wave-uniform) -> rows 8 rg .. 8 rg + 7 of the slab
    const int c = tid % H;
    const int rg = __builtin_amdgcn_readfirstlane(tid / H);

    f32x4 acc[4][NT_];
    f32x4 acch = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT_; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // heads 8..15 of the dheads^T image are never written: zero both buffers once
    for (int idx = tid; idx < 2 * 3 * DPL; idx += NT) {
        const int b = idx / (3 * DPL), rem = idx % (3 * DPL);
        sm16[b * BUF + 3 * (BPL + YPL) + rem] = 0;
    }

    // ---- A operand: raw dG values of this lane, [tile mi][row 8 kq + e]; tile 2p+m <-> gate rows 64w + 32p + 2i + m
    float raw[4][8];
    auto load_a = [&](int slab, int pair) {
        slab = slab < nslab ? slab : nslab - 1;                       // clamped: the tail issues harmless reloads
        // 32-bit element offsets from the (scalar) base pointer: the launch checks N*T*4H < 2^30
        const unsigned off = (unsigned)(r_begin + slab * KS6 + 8 * kq) * (4 * H) + 64 * w + 32 * pair + 2 * j;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float2 t2 = *reinterpret_cast<const float2*>(dgates + (off + (unsigned)e * (4 * H)));
            raw[2 * pair][e] = t2.x;
            raw[2 * pair + 1][e] = t2.y;
        }
    };
    // ---- B / Y / x / dheads staging registers (one slab ahead)
    constexpr int NV = HEADS ? 9 : 8;                                // Y = the same rows of y, shifted by one
    float v[NV], hz = 0.f, xv[XV], dv[DV];
    float kv = 1.f;                                                  // keep[q0 + (lane & 7)]: read back by v_readlane
    int i_start = -1;                                                // row of this thread's group with t == 0
    auto load_b = [&](int slab) {
        slab = slab < nslab ? slab : nslab - 1;
        const int q0 = r_begin + slab * KS6 + 8 * rg;                 // first row of this thread's group (uniform)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int rr = q0 - 1 + i;
            v[i] = y[(unsigned)(rr < 0 ? 0 : rr) * H + c];
        }
        const int tq = q0 % T;                                        // T >= 8: at most one sequence start in 8 rows
        i_start = (tq == 0) ? 0 : (tq + 7 >= T ? T - tq : -1);
        if (i_start >= 0) hz = h0[(unsigned)((q0 + i_start) / T) * H + c];
        // a vector load, NOT a scalar one: s_load returns through lgkmcnt, so every LDS fragment wait behind it
        // would also wait out its HBM latency
        kv = keep ? keep[q0 + (lane & 7)] : 1.f;
#pragma unroll
        for (int k = 0; k < XV; ++k) {
            const int idx = tid + k * NT, q = idx >> 4, f = idx & 15;
            const unsigned r = (unsigned)(r_begin + slab * KS6 + q);
            xv[k] = (f < I) ? x[r * I + f] : (f == 6 ? 1.f : 0.f);
        }
#pragma unroll
        for (int k = 0; k < DV; ++k) {
            const int idx = tid + k * NT, q = idx >> 3, a = idx & 7;
            const unsigned r = (unsigned)(r_begin + slab * KS6 + (q < KS6 ? q : 0));
            dv[k] = (HEADS && a < NH && q < KS6) ? dheads[r * NH + a] : 0.f;
        }
    };
    auto commit_b = [&](int buf) {
        unsigned short* bp = sm16 + buf * BUF;
        unsigned short* yp = bp + 3 * BPL;
        unsigned short* dp = yp + 3 * YPL;
        float hp[8], yy[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float kpi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, kv), i));
            hp[i] = ((i == i_start) ? hz : v[i]) * kpi;
            yy[i] = HEADS ? v[i + 1 < NV ? i + 1 : 0] : 0.f;
        }
        bf16x8 p0, p1, p2;
        split8(hp, p0, p1, p2);
        *reinterpret_cast<bf16x8*>(bp + c * KP + 8 * rg) = p0;
        *reinterpret_cast<bf16x8*>(bp + BPL + c * KP + 8 * rg) = p1;
        *reinterpret_cast<bf16x8*>(bp + 2 * BPL + c * KP + 8 * rg) = p2;
        if (HEADS) {
            split8(yy, p0, p1, p2);
            *reinterpret_cast<bf16x8*>(yp + c * KP + 8 * rg) = p0;
            *reinterpret_cast<bf16x8*>(yp + YPL + c * KP + 8 * rg) = p1;
            *reinterpret_cast<bf16x8*>(yp + 2 * YPL + c * KP + 8 * rg) = p2;
        }
#pragma unroll
        for (int k = 0; k < XV; ++k) {
            const int idx = tid + k * NT, q = idx >> 4, f = idx & 15;
            __bf16 a, b, cc;
            split3(xv[k], a, b, cc);
            unsigned short* d = bp + (H + f) * KP + q;
            d[0] = __builtin_bit_cast(unsigned short, a);
            d[BPL] = __builtin_bit_cast(unsigned short, b);
            d[2 * BPL] = __builtin_bit_cast(unsigned short, cc);
        }
        if (HEADS) {
#pragma unroll
            for (int k = 0; k < DV; ++k) {
                const int idx = tid + k * NT, q = idx >> 3, a = idx & 7;
                if (q < KS6) {
                    __bf16 pa, pb, pc;
                    split3(dv[k], pa, pb, pc);
                    unsigned short* d = dp + a * KP + q;
                    d[0] = __builtin_bit_cast(unsigned short, pa);
                    d[DPL] = __builtin_bit_cast(unsigned short, pb);
                    d[2 * DPL] = __builtin_bit_cast(unsigned short, pc);
                }
            }
        }
    };
    // six piece products, smallest first
    auto mac6 = [&](f32x4& d, const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], d, 0, 0, 0);
    };
    bf16x8 ap[2][3];
    // split the raw values of a tile pair (slab `sl`), then refill the freed registers from slab sl + 1
    auto split_pair = [&](int sl, int pair) {
#pragma unroll
        for (int m = 0; m < 2; ++m) split8(raw[2 * pair + m], ap[m][0], ap[m][1], ap[m][2]);
        load_a(sl + 1, pair);
    };
    auto mfma_pair = [&](int buf, int pair) {
        const unsigned short* bp = sm16 + buf * BUF;
#pragma unroll
        for (int ni = 0; ni < NT_; ++ni) {
            bf16x8 bb[3];
            const unsigned short* src = bp + (16 * ni + j) * KP + 8 * kq;
            bb[0] = *reinterpret_cast<const bf16x8*>(src);
            bb[1] = *reinterpret_cast<const bf16x8*>(src + BPL);
            bb[2] = *reinterpret_cast<const bf16x8*>(src + 2 * BPL);
            mac6(acc[2 * pair][ni], ap[0], bb);
            mac6(acc[2 * pair + 1][ni], ap[1], bb);
            if (HEADS && (ni & 1)) asm volatile("" ::: "memory");   // at most two tiles of B fragments in flight: room for the head tile
        }
        if (HEADS && pair == 1) {                         // dW_head tile of this wave: dheads^T Y[:, 16w .. 16w+16)
            const unsigned short* yp = bp + 3 * BPL;
            const unsigned short* dp = yp + 3 * YPL;
            bf16x8 da[3], yb[3];
            const unsigned short* sa = dp + j * KP + 8 * kq;
            const unsigned short* sb = yp + (16 * w + j) * KP + 8 * kq;
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                da[pc] = *reinterpret_cast<const bf16x8*>(sa + pc * DPL);
                yb[pc] = *reinterpret_cast<const bf16x8*>(sb + pc * YPL);
            }
            mac6(acch, da, yb);
        }
    };

    load_b(0);
    load_a(0, 0);
    load_a(0, 1);
    commit_b(0);
    load_b(1);
    split_pair(0, 0);
    lds_barrier();
    WX_PROF_DECL;
    // one body for both groups -- mfma0 commit split1 mfma1 split0(next) -- and the group decides where in it the
    // slab barrier sits: EARLY waves wait before split0(next), LATE waves after it, so after every barrier the
    // LATE wave of a SIMD streams MFMAs while the EARLY one still splits, and so on round the slab
    for (int sl = 0; sl < nslab; ++sl) {
        const int buf = sl & 1;
        WX_PROF_MARK(0);
        mfma_pair(buf, 0);
        WX_PROF_DEP(acc[0][NT_ - 1]); WX_PROF_DEP(acc[1][NT_ - 1]); WX_PROF_MARK(2);
        commit_b(buf ^ 1);                                // planes of slab sl + 1 (a clamped copy on the last slab)
        load_b(sl + 2);
        WX_PROF_MARK(5);
        split_pair(sl, 1);
        WX_PROF_DEP(ap[0][0]); WX_PROF_DEP(ap[1][2]); WX_PROF_MARK(3);
        mfma_pair(buf, 1);
        WX_PROF_DEP(acc[2][NT_ - 1]); WX_PROF_DEP(acc[3][NT_ - 1]); WX_PROF_MARK(4);
        if (!late) { lds_barrier(); WX_PROF_MARK(7); }
        split_pair(sl + 1, 0);                            // pair 0 of the NEXT slab (raw loaded one slab ago)
        WX_PROF_DEP(ap[0][0]); WX_PROF_DEP(ap[1][2]); WX_PROF_MARK(1);
        if (late) { lds_barrier(); WX_PROF_MARK(7); }
    }
    WX_PROF_FLUSH();
    float* slab = slabs + (size_t)blockIdx.x * WG<H>::SLAB;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT_; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                slab[(size_t)(64 * w + 32 * (mi >> 1) + 2 * (4 * kq + r) + (mi & 1)) * G::NC + 16 * ni + j] = acc[mi][ni][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) slab[(size_t)4 * H * G::NC + (size_t)(4 * kq + r) * H + 16 * w + j] = acch[r];
}

// ------------------------------------------------------------------------------ split-fp16 variant
// lstm_wgrad_x6_kernel with the gate-gradient products as THREE fp16 MFMA products (common.h, split2h) instead of six bf16
// ones.  Differences from the forward / backward kernels' use of the split, forced by the 37 accumulator tiles per wave:
// the residual pieces are kept UNscaled (a = p0 + p1, p1 = fp16(a - p0)) so that main and cross products can share one
// accumulator, and both operands are block-scaled up by powers of two so that those residuals stay in fp16's normal
// range -- the Hprev columns by 2^10 (|h| < 1), the x | 1 columns by 2^4 (|x| < 4096 is assumed; a column scale is undone
// per output column), dG by a running scale per gate row (below).  The head tile (dheads^T Y, one of 37) stays on the bf16
// split.
constexpr float WGH_BSCALE = 1024.0f;      // Hprev columns
constexpr float WGH_XSCALE = 16.0f;        // x | 1 columns (the last 16-column tile)

template <int H>
struct WGH {
    static constexpr int NW = H / 16, NT = H * 4;
    static constexpr int NT_ = H / 16 + 1, NC = H + 16;
    static constexpr int KP = KS6 + 8;
    static constexpr int BPL = NC * KP, YPL = H * KP, DPL = 16 * KP;
    static constexpr int BUF = 2 * BPL + 3 * (YPL + DPL);            // 16-bit elements per buffer
    static constexpr size_t LDS = 2 * BUF * sizeof(unsigned short);
    static constexpr int XV = 16 * KS6 / NT;
    static constexpr int DV = (8 * KS6 + NT - 1) / NT;
};

__device__ __forceinline__ void split2u(float a, _Float16& p0, _Float16& p1) {
    p0 = (_Float16)a;
    p1 = (_Float16)(a - (float)p0);
}
__device__ __forceinline__ void split8u(const float (&v)[8], f16x8& p0, f16x8& p1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        _Float16 a, b;
        split2u(v[i], a, b);
        p0[i] = a; p1[i] = b;
    }
}

template <int H, bool HEADS>
__global__ __launch_bounds__(H * 4) void lstm_wgrad_h3_kernel(
    const float* __restrict__ dgates, const float* __restrict__ y, const float* __restrict__ keep,
    const float* __restrict__ h0, const float* __restrict__ x, int I, const float* __restrict__ dheads, int NH,
    int N, int T, int rows_per_block, float* __restrict__ slabs) {
    using G = WGH<H>;
    constexpr int NT_ = G::NT_, KP = G::KP, BPL = G::BPL, YPL = G::YPL, DPL = G::DPL, BUF = G::BUF, NT = G::NT;
    constexpr int XV = G::XV, DV = G::DV, NW = G::NW;
    extern __shared__ __attribute__((aligned(16))) unsigned short sm16[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool late = w >= NW / 2;                 // waves w and w + NW/2 share a SIMD
    const int j = lane & 15, kq = lane >> 4;
    const int r_begin = blockIdx.x * rows_per_block;
    const int nslab = rows_per_block / KS6;

    // staging coordinates: column c, row group rg (wave-uniform) -> rows 8 rg .. 8 rg + 7 of the slab
    const int c = tid % H;
    const int rg = __builtin_amdgcn_readfirstlane(tid / H);

    f32x4 acc[4][NT_];
    f32x4 acch = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT_; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // heads 8..15 of the dheads^T image are never written: zero both buffers once
    for (int idx = tid; idx < 2 * 3 * DPL; idx += NT) {
        const int b = idx / (3 * DPL), rem = idx % (3 * DPL);
        sm16[b * BUF + 2 * BPL + 3 * YPL + rem] = 0;
    }

    // ---- A operand: raw dG values of this lane, [tile mi][row 8 kq + e]; tile 2p+m <-> gate rows 64w + 32p + 2i + m
    float raw[4][8];
    auto load_a_one = [&](int slab, int pair, int e) {
        slab = slab < nslab ? slab : nslab - 1;                       // clamped: the tail issues harmless reloads
        // 32-bit element offsets from the (scalar) base pointer: the launch checks N*T*4H < 2^30
        const unsigned off = (unsigned)(r_begin + slab * KS6 + 8 * kq) * (4 * H) + 64 * w + 32 * pair + 2 * j;
        const float2 t2 = *reinterpret_cast<const float2*>(dgates + (off + (unsigned)e * (4 * H)));
        raw[2 * pair][e] = t2.x;
        raw[2 * pair + 1][e] = t2.y;
    };
    auto load_a = [&](int slab, int pair) {
#pragma unroll
        for (int e = 0; e < 8; ++e) load_a_one(slab, pair, e);
    };
    // ---- B / Y / x / dheads staging registers (one slab ahead)
    constexpr int NV = HEADS ? 9 : 8;                                // Y = the same rows of y, shifted by one
    float v[NV], hz = 0.f, xv[XV], dv[DV];
    float kv = 1.f;                                                  // keep[q0 + (lane & 7)]: read back by v_readlane
    int i_start = -1;                                                // row of this thread's group with t == 0
    auto load_b_y = [&](int slab, int i) {
        slab = slab < nslab ? slab : nslab - 1;
        const int rr = r_begin + slab * KS6 + 8 * rg - 1 + i;
        v[i] = y[(unsigned)(rr < 0 ? 0 : rr) * H + c];
    };
    auto load_b_rest = [&](int slab) {
        slab = slab < nslab ? slab : nslab - 1;
        const int q0 = r_begin + slab * KS6 + 8 * rg;                 // first row of this thread's group (uniform)
        const int tq = q0 % T;                                        // T >= 8: at most one sequence start in 8 rows
        i_start = (tq == 0) ? 0 : (tq + 7 >= T ? T - tq : -1);
        if (i_start >= 0) hz = h0[(unsigned)((q0 + i_start) / T) * H + c];
        // a vector load, NOT a scalar one: s_load returns through lgkmcnt, so every LDS fragment wait behind it
        // would also wait out its HBM latency
        kv = keep ? keep[q0 + (lane & 7)] : 1.f;
#pragma unroll
        for (int k = 0; k < XV; ++k) {
            const int idx = tid + k * NT, q = idx >> 4, f = idx & 15;
            const unsigned r = (unsigned)(r_begin + slab * KS6 + q);
            xv[k] = (f < I) ? x[r * I + f] : (f == 6 ? 1.f : 0.f);
        }
#pragma unroll
        for (int k = 0; k < DV; ++k) {
            const int idx = tid + k * NT, q = idx >> 3, a = idx & 7;
            const unsigned r = (unsigned)(r_begin + slab * KS6 + (q < KS6 ? q : 0));
            dv[k] = (HEADS && a < NH && q < KS6) ? dheads[r * NH + a] : 0.f;
        }
    };
    auto load_b = [&](int slab) {
#pragma unroll
        for (int i = 0; i < NV; ++i) load_b_y(slab, i);
        load_b_rest(slab);
    };
    auto commit_b = [&](int buf) {
        unsigned short* bp = sm16 + buf * BUF;
        unsigned short* yp = bp + 2 * BPL;
        unsigned short* dp = yp + 3 * YPL;
        float hp[8], yy[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float kpi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, kv), i));
            hp[i] = ((i == i_start) ? hz : v[i]) * (kpi * WGH_BSCALE);
            yy[i] = HEADS ? v[i + 1 < NV ? i + 1 : 0] : 0.f;
        }
        {
            f16x8 q0, q1;
            split8u(hp, q0, q1);
            *reinterpret_cast<f16x8*>(bp + c * KP + 8 * rg) = q0;
            *reinterpret_cast<f16x8*>(bp + BPL + c * KP + 8 * rg) = q1;
        }
        if (HEADS) {
            bf16x8 p0, p1, p2;
            split8(yy, p0, p1, p2);
            *reinterpret_cast<bf16x8*>(yp + c * KP + 8 * rg) = p0;
            *reinterpret_cast<bf16x8*>(yp + YPL + c * KP + 8 * rg) = p1;
            *reinterpret_cast<bf16x8*>(yp + 2 * YPL + c * KP + 8 * rg) = p2;
        }
#pragma unroll
        for (int k = 0; k < XV; ++k) {
            const int idx = tid + k * NT, q = idx >> 4, f = idx & 15;
            _Float16 a, b;
            split2u(xv[k] * WGH_XSCALE, a, b);
            unsigned short* d = bp + (H + f) * KP + q;
            d[0] = h_bits(a);
            d[BPL] = h_bits(b);
        }
        if (HEADS) {
#pragma unroll
            for (int k = 0; k < DV; ++k) {
                const int idx = tid + k * NT, q = idx >> 3, a = idx & 7;
                if (q < KS6) {
                    __bf16 pa, pb, pc;
                    split3(dv[k], pa, pb, pc);
                    unsigned short* d = dp + a * KP + q;
                    d[0] = __builtin_bit_cast(unsigned short, pa);
                    d[DPL] = __builtin_bit_cast(unsigned short, pb);
                    d[2 * DPL] = __builtin_bit_cast(unsigned short, pc);
                }
            }
        }
    };
    // three fp16 piece products into one accumulator (unscaled residuals), smallest first
    auto mac3 = [&](f32x4& d, const f16x8 (&a)[2], const f16x8 (&b)[2]) {
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], d, 0, 0, 0);
    };
    // six bf16 piece products (the head tile: dheads^T Y), smallest first
    auto mac6 = [&](f32x4& d, const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], d, 0, 0, 0);
    };
    // A pieces of the current tile pair.  dG spans many binades, fp16 does not: every gate ROW keeps its own power-of-two
    // scale (lane (i, kq) holds row i of tiles 2p, 2p+1; the four kq lanes agree on it through the two gfx950 row swaps),
    // lowered -- and that row of the tile's NT_ accumulators rescaled, exactly -- whenever a slab's largest magnitude in
    // the row would leave [.., 2^14).  The residual is kept unscaled so that main and cross products share ONE
    // accumulator (there is no room for a second set beside 37 tiles): full relative precision for elements within 2^16
    // of their row's running maximum, an absolute error below 2^-39 of it for smaller ones.  Accumulator register r of
    // lane (j, kq) belongs to row 4 kq + r, whose scale lives in lanes i = 4 kq + r: fetched by ds_bpermute when (rarely,
    // after the first slabs) some row's scale changes.
    f16x8 ap[2][2];
    unsigned exr4 = 0xe4e4e4e4u;                                      // the four tiles' row exponents, one biased byte each (100 + 128)
    auto exr_get = [&](int mi) { return (int)((exr4 >> (8 * mi)) & 255u) - 128; };
    auto exr_set = [&](int mi, int v) { exr4 = (exr4 & ~(255u << (8 * mi))) | ((unsigned)(v + 128) << (8 * mi)); };
    auto row_max4 = [&](float m) {                                   // max over the four kq lanes of the same i
        const unsigned u = __builtin_bit_cast(unsigned, m);
        const auto s16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        m = fmaxf(__builtin_bit_cast(float, (unsigned)s16[0]), __builtin_bit_cast(float, (unsigned)s16[1]));
        const unsigned v2 = __builtin_bit_cast(unsigned, m);
        const auto s32 = __builtin_amdgcn_permlane32_swap(v2, v2, false, false);
        return fmaxf(__builtin_bit_cast(float, (unsigned)s32[0]), __builtin_bit_cast(float, (unsigned)s32[1]));
    };
    auto split_pair = [&](int sl, int pair) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int mi = 2 * pair + m;
            float mx = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(raw[mi][e]));
            mx = row_max4(mx);
            const int eo = exr_get(mi);
            const int ne = mx > 0.f ? min(min(14 - __builtin_amdgcn_frexp_expf(mx), 100), eo) : eo;
            const int delta = eo - ne;                                 // >= 0, the same in the four kq lanes of row i
            if (__builtin_amdgcn_ballot_w64(delta != 0)) {             // wave-uniform
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int dr = __builtin_amdgcn_ds_bpermute((4 * kq + r) * 4, delta);
                    const float f = __builtin_amdgcn_ldexpf(1.0f, -dr);
#pragma unroll
                    for (int ni = 0; ni < NT_; ++ni) acc[mi][ni][r] *= f;
                }
                exr_set(mi, ne);
            }
            float sc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) sc[e] = __builtin_amdgcn_ldexpf(raw[mi][e], ne);
            split8u(sc, ap[m][0], ap[m][1]);
        }
    };
    // The loads ride BETWEEN the column tiles of the MFMA blocks, one or two per tile, instead of in bursts of 8 / 8 / 13 behind
    // the splits: while the memory pipe is taking a burst a wave's issue stalls for ~95 cycles per load instruction (a lone
    // load issues in ~15: tools/vmem_issue_probe.hip, profiles/r04_wgrad_phases.log) and the pipe then idles through the
    // matrix phases.  Pair 0's block carries the dG rows of slab sl + 1 / pair 0 (their registers were split just before it),
    // pair 1's block those of pair 1 and the B-side rows of slab sl + 2 (their registers were committed before it).
    auto mfma_pair = [&](int buf, int pair, int sl) {
        const unsigned short* bp = sm16 + buf * BUF;
#pragma unroll
        for (int ni = 0; ni < NT_; ++ni) {
            f16x8 bb[2];
            const unsigned short* src = bp + (16 * ni + j) * KP + 8 * kq;
            bb[0] = *reinterpret_cast<const f16x8*>(src);
            bb[1] = *reinterpret_cast<const f16x8*>(src + BPL);
            mac3(acc[2 * pair][ni], ap[0], bb);
            mac3(acc[2 * pair + 1][ni], ap[1], bb);
#pragma unroll
            for (int e = ni; e < 8; e += NT_) load_a_one(sl + 1, pair, e);          // (H = 64 has five column tiles for eight rows)
            if (pair == 1) {
#pragma unroll
                for (int i = ni; i < NV; i += NT_) load_b_y(sl + 2, i);
                if (ni == NT_ - 1) load_b_rest(sl + 2);
            }
            asm volatile("" ::: "memory");                         // one tile of B fragments in flight; the loads stay where they are
        }
        if (HEADS && pair == 1) {                         // dW_head tile of this wave: dheads^T Y[:, 16w .. 16w+16)
            const unsigned short* yp = bp + 2 * BPL;
            const unsigned short* dp = yp + 3 * YPL;
            bf16x8 da[3], yb[3];
            const unsigned short* sa = dp + j * KP + 8 * kq;
            const unsigned short* sb = yp + (16 * w + j) * KP + 8 * kq;
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                da[pc] = *reinterpret_cast<const bf16x8*>(sa + pc * DPL);
                yb[pc] = *reinterpret_cast<const bf16x8*>(sb + pc * YPL);
            }
            mac6(acch, da, yb);
        }
    };

    load_b(0);
    load_a(0, 0);
    load_a(0, 1);
    commit_b(0);
    load_b(1);
    split_pair(0, 0);
    lds_barrier();
    WX_PROF_DECL;
    // one body for both groups -- mfma0 commit split1 mfma1 split0(next) -- and the group decides where in it the
    // slab barrier sits: EARLY waves wait before split0(next), LATE waves after it, so after every barrier the
    // LATE wave of a SIMD streams MFMAs while the EARLY one still splits, and so on round the slab
    for (int sl = 0; sl < nslab; ++sl) {
        const int buf = sl & 1;
        WX_PROF_MARK(0);
        mfma_pair(buf, 0, sl);
        WX_PROF_DEP(acc[0][NT_ - 1]); WX_PROF_DEP(acc[1][NT_ - 1]); WX_PROF_MARK(2);
        commit_b(buf ^ 1);                                // planes of slab sl + 1 (a clamped copy on the last slab)
        WX_PROF_MARK(5);
        split_pair(sl, 1);
        WX_PROF_DEP(ap[0][0]); WX_PROF_DEP(ap[1][1]); WX_PROF_MARK(3);
        mfma_pair(buf, 1, sl);
        WX_PROF_DEP(acc[2][NT_ - 1]); WX_PROF_DEP(acc[3][NT_ - 1]); WX_PROF_MARK(4);
        if (!late) { lds_barrier(); WX_PROF_MARK(7); }
        split_pair(sl + 1, 0);                            // pair 0 of the NEXT slab (raw loaded one slab ago)
        WX_PROF_DEP(ap[0][0]); WX_PROF_DEP(ap[1][1]); WX_PROF_MARK(1);
        if (late) { lds_barrier(); WX_PROF_MARK(7); }
    }
    WX_PROF_FLUSH();
    float* slab = slabs + (size_t)blockIdx.x * WG<H>::SLAB;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int er = __builtin_amdgcn_ds_bpermute((4 * kq + r) * 4, exr_get(mi));   // scale of row 4 kq + r of tile mi
            const float usc = __builtin_amdgcn_ldexpf(1.0f / WGH_BSCALE, -er);
#pragma unroll
            for (int ni = 0; ni < NT_; ++ni)
                slab[(size_t)(64 * w + 32 * (mi >> 1) + 2 * (4 * kq + r) + (mi & 1)) * G::NC + 16 * ni + j] =
                    acc[mi][ni][r] * (ni == NT_ - 1 ? usc * (WGH_BSCALE / WGH_XSCALE) : usc);
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) slab[(size_t)4 * H * G::NC + (size_t)(4 * kq + r) * H + 16 * w + j] = acch[r];
}

// sum the slabs in block order and scatter into dW_hh / dW_ih / db / dW_head
template <int H>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, int nb, int I, int NH,
                                                           float* __restrict__ dw_ih, float* __restrict__ dw_hh,
                                                           float* __restrict__ db, float* __restrict__ db_hh,
                                                           float* __restrict__ dw_head) {
    using G = WG<H>;
    const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= G::SLAB) return;
    float p8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = 0;
    for (; b + 8 <= nb; b += 8)
#pragma unroll
        for (int k = 0; k < 8; ++k) p8[k] += slabs[(size_t)(b + k) * G::SLAB + o];
    for (; b < nb; ++b) p8[0] += slabs[(size_t)b * G::SLAB + o];
    const float s = ((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]));
    const size_t gsz = (size_t)4 * H * G::NC;
    if (o < gsz) {
        const int m = (int)(o / G::NC), c = (int)(o % G::NC);
        if (c < H) dw_hh[(size_t)m * H + c] = s;
        else if (c - H < I) dw_ih[(size_t)m * I + (c - H)] = s;
        else if (c - H == 6) {
            db[m] = s;
            if (db_hh) db_hh[m] = s;          // nn.LSTM's two bias vectors always enter as a sum: the same gradient
        }
    } else if (dw_head) {
        const int a = (int)((o - gsz) / H), uu = (int)((o - gsz) % H);
        if (a < NH) dw_head[(size_t)a * H + uu] = s;
    }
}

template <int H>
static int launch_wgrad(uav_ctx* ctx, const float* dgates, const float* y_prev_src, const float* keep, const float* h0,
                        const float* x, int I, const float* ytop, const float* dheads, int NH, int N, int T,
                        float* dw_ih, float* dw_hh, float* db, float* db_hh, float* dw_head, hipStream_t st) {
    using G = WG<H>;
    const int64_t NTr = (int64_t)N * T;
    int nb = ctx->num_cu;
    const int64_t max_nb = (int64_t)(ctx->ws_bytes / (G::SLAB * sizeof(float)));
    if (nb > max_nb) nb = (int)max_nb;
    UAV_REQUIRE(nb >= 1, "uav_lstm_wgrad: workspace too small for one slab");
    int64_t rpb = (NTr + nb - 1) / nb;
    rpb = (rpb + KC - 1) / KC * KC;
    nb = (int)((NTr + rpb - 1) / rpb);
    UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(&lstm_wgrad_kernel<H>), (int)G::LDS));
    float* slabs = (float*)ctx->ws;
    // split-bf16 variant: full 32-row slabs everywhere, T >= 8 (at most one sequence start per 8 rows)
    {
        int64_t rpx = (NTr + nb - 1) / nb;
        rpx = (rpx + KS6 - 1) / KS6 * KS6;
        const int nbx = (int)((NTr + rpx - 1) / rpx);
        const bool x6_ok = (NTr % rpx == 0) && T >= 8 && I <= 6 && (!dheads || NH <= 8) && NTr * 4 * H < (1ll << 30) &&
                           (y_prev_src == ytop || ytop == nullptr) && !uav_want_f32_mfma();
        if (x6_ok) {
            using GX = WGX<H>;
            UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(&lstm_wgrad_x6_kernel<H, true>), (int)GX::LDS));
            UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(&lstm_wgrad_x6_kernel<H, false>), (int)GX::LDS));
            const bool h3 = !uav_want_bf16x6();
            if (h3) {
                using GH = WGH<H>;
                UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(&lstm_wgrad_h3_kernel<H, true>), (int)GH::LDS));
                UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(&lstm_wgrad_h3_kernel<H, false>), (int)GH::LDS));
                if (dheads)
                    hipLaunchKernelGGL((lstm_wgrad_h3_kernel<H, true>), dim3(nbx), dim3(H * 4), GH::LDS, st, dgates, y_prev_src,
                                       keep, h0, x, I, dheads, NH, N, T, (int)rpx, slabs);
                else
                    hipLaunchKernelGGL((lstm_wgrad_h3_kernel<H, false>), dim3(nbx), dim3(H * 4), GH::LDS, st, dgates, y_prev_src,
                                       keep, h0, x, I, dheads, NH, N, T, (int)rpx, slabs);
            } else if (dheads)
                hipLaunchKernelGGL((lstm_wgrad_x6_kernel<H, true>), dim3(nbx), dim3(H * 4), GX::LDS, st, dgates, y_prev_src,
                                   keep, h0, x, I, dheads, NH, N, T, (int)rpx, slabs);
            else
                hipLaunchKernelGGL((lstm_wgrad_x6_kernel<H, false>), dim3(nbx), dim3(H * 4), GX::LDS, st, dgates, y_prev_src,
                                   keep, h0, x, I, dheads, NH, N, T, (int)rpx, slabs);
            hipLaunchKernelGGL((wgrad_reduce_kernel<H>), dim3((unsigned)((G::SLAB + 255) / 256)), dim3(256), 0, st, slabs,
                               nbx, I, NH, dw_ih, dw_hh, db, db_hh, dw_head);
            UAV_LAUNCH_CHECK();
            return 0;
        }
    }
    // exact-f32 MFMA (UAV_ARITH_F32_MFMA), ragged shapes the split kernels do not take
    hipLaunchKernelGGL((lstm_wgrad_kernel<H>), dim3(nb), dim3(H * 4), G::LDS, st, dgates, y_prev_src, keep, h0, x, I, ytop,
                       dheads, NH, N, T, rpb, slabs);
    hipLaunchKernelGGL((wgrad_reduce_kernel<H>), dim3((unsigned)((G::SLAB + 255) / 256)), dim3(256), 0, st, slabs, nb, I,
                       NH, dw_ih, dw_hh, db, db_hh, dw_head);
    UAV_LAUNCH_CHECK();
    return 0;
}

// fast path entry used by uav_lstm_wgrad (lstm.hip) when I <= 6 and H in {64,128}
int lstm_wgrad_fused(uav_ctx* ctx, const float* dgates, const float* y_prev_src, const float* keep, const float* h0,
                     const float* x, int I, const float* ytop, const float* dheads, int NH, int N, int T, int H,
                     float* dw_ih, float* dw_hh, float* db, float* db_hh, float* dw_head, hipStream_t st) {
    switch (H) {
        case 64: return launch_wgrad<64>(ctx, dgates, y_prev_src, keep, h0, x, I, ytop, dheads, NH, N, T, dw_ih, dw_hh, db, db_hh, dw_head, st);
        case 128: return launch_wgrad<128>(ctx, dgates, y_prev_src, keep, h0, x, I, ytop, dheads, NH, N, T, dw_ih, dw_hh, db, db_hh, dw_head, st);
    }
    uav_set_error("lstm_wgrad_fused: H=%d unsupported", H);
    return 2;
}

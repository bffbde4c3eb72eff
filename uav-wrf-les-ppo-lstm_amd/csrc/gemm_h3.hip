// gemm_h3.hip -- the dense building block on the 16-bit matrix pipe at f32 accuracy: C[M][N] (+)= op(A) op(B) + bias with
// every f32 product evaluated as three fp16 piece products (common.h, split2h: a = p0 + 2^-11 p1, main and cross
// accumulators in f32, v_mfma_f32_16x16x32_f16).  Same operand addressing as gemm.hip (element strides, so NN / NT / TN
// are one entry), for the LARGE products of the per-step LSTM path (BASELINE C5, h = 256 stacked): the weight gradients
// dW = dG^T [h_prev | x] (K = N*T rows) and the input gradient dx = dG W_ih, which the exact-f32 kernel runs at its
// matrix peak (4.4 ms each at 1 M rows x 1024 x 256) and this one at the fp16 pipe's rate.
//
// Workgroup tile 128 (M) x BN (N = 128 or 256) x 32 (K), four waves as 2 x 2, wave tile 64 x BN/2: 4 x BN/32 accumulator
// tiles twice (main, cross) = up to 256 accumulator registers of the 512 a 256-thread workgroup may use per lane.
// Operands travel global -> registers (f32, one slab ahead) -> split -> LDS piece planes [piece][row][32 k + 8 pad]
// (16-bit, k-contiguous: every fragment is one conflict-free ds_read_b128) -> MFMA; two LDS buffers, one barrier per slab.
// Where the time goes at [1 M x 1024]^T [1 M x 256] (profiles/r02_gemm_h3_ablation.log): 2.33 ms whole; 1.27 ms without
// the loop's global loads; 1.84 ms without the MFMAs; 2.04 ms with the split arithmetic replaced by bit moves -- one slab
// of loads in flight per CU does not cover the loaded memory latency.  Tried and dropped: eight waves with 64 x 64 wave
// tiles and the loads two slabs ahead (256 registers per wave do not hold 128 accumulators + two stages without
// spilling, both waves of a SIMD run the same phase: 5.2 ms); touching the slab after next with one dword per 128-byte
// line so that the stage's loads hit L2 (vmcnt retires in order, so the stage's loads wait behind the touches: 2.94 ms).
//   * an operand whose k index is contiguous in memory is staged by float4 loads along k (thread = row, 4 k);
//   * an operand whose row index is contiguous (the "transposed" side of TN) by eight coalesced dword loads down k per
//     thread (thread = column, 8 k) -> one ds_write_b128 per piece: the transposition costs nothing in LDS.
// fp16's range: |a| < 65504 is the caller's business (weights, activations, observations under the trainer's range guard);
// an operand with a wide dynamic range (gradients, ~1e-6) is block-scaled by ONE power of two derived on the device from
// its absolute maximum (a_absmax, written by colsum's fused max; the product is exact under the scale, undone in the
// epilogue).  Values more than 2^28 below that maximum lose relative -- not absolute -- accuracy.
//
// XCD-aware order: workgroups are dispatched round-robin over the 8 XCDs (own L2 each), so the tiles that share operand
// slabs -- all tiles of one K range under split-K, the column tiles of one row block otherwise -- are placed on
// consecutive slots of ONE XCD: the operands then cross HBM once per group instead of once per tile.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#ifndef H3_ABL
#define H3_ABL 0        // ablation builds only (tools/ab_gemm_h3.sh): 1 no split arithmetic, 2 no loads, 3 no MFMAs, 4 no commit
#endif

namespace {

constexpr int HM = 128, HK = 32, HKP = HK + 8;       // tile rows of A, slab depth, padded k extent of a plane row (halves)

template <int BN>
struct H3Tile {
    static constexpr int A_PLANE = HM * HKP, B_PLANE = BN * HKP;              // halves per piece plane
    static constexpr int BUF = 2 * A_PLANE + 2 * B_PLANE;                     // halves per buffer
    static constexpr size_t LDS = 2 * (size_t)BUF * sizeof(unsigned short);   // two buffers
};

// ---- staging: ROWS x 32 slab of an operand, element (r, k) = base[r * s_r + k * s_k] -----------------------------------
// KC = true: s_k == 1 (k contiguous): thread -> (row, 4 consecutive k), ROWS * 8 float4 per slab
// KC = false: s_r == 1 (row contiguous): thread -> (row, 8 consecutive k), ROWS * 4 items of 8 dwords per slab
template <int ROWS, bool KC>
struct Stage {
    static constexpr int ITEMS = KC ? ROWS * 8 / 256 : ROWS * 4 / 256;
    static constexpr int NF = KC ? 4 * ITEMS : 8 * ITEMS;
    float v[NF];

    // Addresses are a wave-uniform 64-bit base (scalar registers) plus a 32-bit per-lane offset, so a load costs no vector
    // address arithmetic; full slabs take the branch-free path, only a K tail pays for per-element guards.
    __device__ __forceinline__ void load(const float* __restrict__ base, int64_t ld, int64_t r0, int64_t k0, int64_t ke) {
        const int t = threadIdx.x;
        if (k0 + HK <= ke) {                                                      // uniform: a full slab, no guards
            if (KC) {
                const float* ub = base + r0 * ld + k0;                            // uniform
                const unsigned kc = (t & 7) * 4, ldu = (unsigned)ld;
#pragma unroll
                for (int i = 0; i < ITEMS; ++i) {
                    const float4 q = *reinterpret_cast<const float4*>(ub + ((unsigned)((t >> 3) + 32 * i) * ldu + kc));
                    v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < ITEMS; ++i) {
                    const int idx = t + 256 * i;
                    const unsigned r = idx % ROWS;
                    const int g = __builtin_amdgcn_readfirstlane(idx / ROWS);     // 0..3, the same for a whole wave (ROWS >= 64)
                    const float* ub = base + (k0 + 8 * g) * ld + r0;              // uniform
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[8 * i + j] = (ub + j * ld)[r];
                }
            }
        } else if (KC) {                                                          // the K tail: per-element guards
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const int r = (t >> 3) + 32 * i, kc = (t & 7) * 4;
                const float* p = base + (r0 + r) * ld + k0 + kc;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[4 * i + j] = (k0 + kc + j < ke) ? p[j] : 0.f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const int idx = t + 256 * i, r = idx % ROWS, g = idx / ROWS;
                const float* p = base + (k0 + 8 * g) * ld + r0 + r;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[8 * i + j] = (k0 + 8 * g + j < ke) ? p[j * ld] : 0.f;
            }
        }
    }
    // split into the two fp16 pieces (after the block scale) and park them k-contiguous in the piece planes
    __device__ __forceinline__ void commit(unsigned short* __restrict__ plane0, unsigned short* __restrict__ plane1,
                                           float scale) const {
        const int t = threadIdx.x;
        constexpr int PL = ROWS * HKP;
        (void)PL;
        if (KC) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const int r = (t >> 3) + 32 * i, kc = (t & 7) * 4;
                unsigned short a[4], b[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#if H3_ABL == 1
                    const unsigned u = __builtin_bit_cast(unsigned, v[4 * i + j]);
                    a[j] = (unsigned short)u; b[j] = (unsigned short)(u >> 16);
#else
                    _Float16 p0, p1;
                    split2h(v[4 * i + j] * scale, p0, p1);
                    a[j] = h_bits(p0); b[j] = h_bits(p1);
#endif
                }
                *reinterpret_cast<uint2*>(plane0 + r * HKP + kc) = make_uint2(a[0] | (unsigned)a[1] << 16, a[2] | (unsigned)a[3] << 16);
                *reinterpret_cast<uint2*>(plane1 + r * HKP + kc) = make_uint2(b[0] | (unsigned)b[1] << 16, b[2] | (unsigned)b[3] << 16);
            }
        } else {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const int idx = t + 256 * i, r = idx % ROWS, g = idx / ROWS;
                f16x8 a, b;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
#if H3_ABL == 1
                    const unsigned u = __builtin_bit_cast(unsigned, v[8 * i + j]);
                    a[j] = __builtin_bit_cast(_Float16, (unsigned short)u); b[j] = __builtin_bit_cast(_Float16, (unsigned short)(u >> 16));
#else
                    _Float16 p0, p1;
                    split2h(v[8 * i + j] * scale, p0, p1);
                    a[j] = p0; b[j] = p1;
#endif
                }
                *reinterpret_cast<f16x8*>(plane0 + r * HKP + 8 * g) = a;
                *reinterpret_cast<f16x8*>(plane1 + r * HKP + 8 * g) = b;
            }
        }
    }
};

// power-of-two block scale from the bits of max |a| (0, inf, NaN -> 1): max * scale lands in [2^13, 2^14)
__device__ __forceinline__ float block_scale(const unsigned* __restrict__ absmax_bits) {
    if (!absmax_bits) return 1.f;
    const unsigned b = *absmax_bits & 0x7fffffffu;
    const int e = (int)(b >> 23);                         // biased exponent of the maximum
    if (e == 0 || e == 255) return 1.f;
    int se = 127 + 13 - (e - 127);                        // biased exponent of 2^(13 - floor(log2 max))
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    return __builtin_bit_cast(float, (unsigned)se << 23);
}

template <int BN, bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_h3_kernel(int64_t M, int64_t N, int64_t K, const float* __restrict__ A, int64_t lda,
                                                      const float* __restrict__ B, int64_t ldb, float* __restrict__ C,
                                                      int64_t ldc, const float* __restrict__ bias, int accumulate,
                                                      int64_t k_per_split, float* __restrict__ slabs,
                                                      const unsigned* __restrict__ a_absmax, int tm, int tn, int S) {
    using TL = H3Tile<BN>;
    constexpr int NTW = BN / 32;                          // 16-column tiles per wave
    extern __shared__ __attribute__((aligned(16))) unsigned short sm16[];
    // ---- XCD-aware placement (see the header)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int G = (S > 1) ? tm * tn : tn, ngroups = (S > 1) ? S : tm;
    const int g = (slot / G) * 8 + xcd, within = slot % G;
    if (g >= ngroups) return;
    const int z = (S > 1) ? g : 0;
    const int mi = (S > 1) ? within % tm : g, ni = (S > 1) ? within / tm : within;
    const int64_t m0 = (int64_t)mi * HM, n0 = (int64_t)ni * BN;
    const int64_t kb = (int64_t)z * k_per_split;
    const int64_t ke = (kb + k_per_split < K) ? kb + k_per_split : K;

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int fi = lane & 15, kq = lane >> 4;
    const float sa = block_scale(a_absmax);

    f32x4 acc0[4][NTW], acc1[4][NTW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc0[i][j] = acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Stage<HM, AKC> ra;
    Stage<BN, BKC> rb;
    ra.load(A, lda, m0, kb, ke);
    rb.load(B, ldb, n0, kb, ke);
    ra.commit(sm16, sm16 + TL::A_PLANE, sa);
    rb.commit(sm16 + 2 * TL::A_PLANE, sm16 + 2 * TL::A_PLANE + TL::B_PLANE, 1.f);
    __syncthreads();
    // one slab of products out of LDS buffer `buf`
    auto multiply = [&](int buf) {
        const unsigned short* bp = sm16 + buf * TL::BUF;
        const unsigned short* a0p = bp + (wm * 64 + fi) * HKP + 8 * kq;
        const unsigned short* b0p = bp + 2 * TL::A_PLANE + (wn * (BN / 2) + fi) * HKP + 8 * kq;
        f16x8 a0[4], a1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a0[i] = *reinterpret_cast<const f16x8*>(a0p + i * 16 * HKP);
            a1[i] = *reinterpret_cast<const f16x8*>(a0p + TL::A_PLANE + i * 16 * HKP);
        }
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const f16x8 b0 = *reinterpret_cast<const f16x8*>(b0p + j * 16 * HKP);
            const f16x8 b1 = *reinterpret_cast<const f16x8*>(b0p + TL::B_PLANE + j * 16 * HKP);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc0[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[i], b0, acc0[i][j], 0, 0, 0);
                acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[i], b1, acc1[i][j], 0, 0, 0);
                acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[i], b0, acc1[i][j], 0, 0, 0);
            }
        }
    };
    int buf = 0;
    // all slabs but the last: the next slab's loads fly under this slab's MFMAs, its split + LDS writes follow them
    for (int64_t k0 = kb; k0 + HK < ke; k0 += HK) {
#if H3_ABL != 2
        ra.load(A, lda, m0, k0 + HK, ke);
        rb.load(B, ldb, n0, k0 + HK, ke);
#endif
#if H3_ABL != 3
        multiply(buf);
#endif
        unsigned short* np = sm16 + (buf ^ 1) * TL::BUF;
#if H3_ABL == 4
#pragma unroll
        for (int q = 0; q < decltype(ra)::NF; ++q) asm volatile("" ::"v"(ra.v[q]));
#pragma unroll
        for (int q = 0; q < decltype(rb)::NF; ++q) asm volatile("" ::"v"(rb.v[q]));
        (void)np;
#else
        ra.commit(np, np + TL::A_PLANE, sa);
        rb.commit(np + 2 * TL::A_PLANE, np + 2 * TL::A_PLANE + TL::B_PLANE, 1.f);
#endif
        __syncthreads();
        buf ^= 1;
    }
    multiply(buf);
    // ---- epilogue: C = (main + 2^-11 cross) / scale (+ bias), or this split's slab
    const float inv = 1.0f / sa;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int64_t col = n0 + wn * (BN / 2) + j * 16 + fi;
            const float bv = (bias && !slabs) ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = m0 + wm * 64 + i * 16 + 4 * kq + r;
                const float val = (acc0[i][j][r] + H3_LO * acc1[i][j][r]) * inv;
                if (slabs) {
                    slabs[((int64_t)z * M + row) * N + col] = val;
                } else {
                    float* cp = C + row * ldc + col;
                    *cp = (accumulate ? *cp : 0.f) + val + bv;
                }
            }
        }
}

// ------------------------------------------------------------------------------ TN kernel (the dW shape)
// dW = dG^T [h_prev | y]: both operands contiguous along their ROW index, K = envs x steps down the columns.  Same arithmetic,
// tile and K order as gemm_h3_kernel -- the results are bit-identical -- with the instruction streams rebuilt
// (profiles/r04_gemm_tn_ablation.log: 2.31 -> 1.86 ms at 1 M x 1024 x 256):
//   * a thread's share of a slab is a 4-row x 2-k A item and a 4-row x 4-k B item read as dwordx4 along the row index (6 loads
//     per slab instead of 24 dword loads; whole 128-byte lines per 32 lanes), addressed as a wave-uniform row pointer + a
//     32-bit lane byte offset; two slabs are in flight while a third is multiplied;
//   * an item's four rows go to LDS rows e * ROWS/4 + rg (row = 4 rg + e), which keeps the piece planes k-contiguous for the
//     b128 fragment reads; tile rows / columns are therefore PERMUTED inside the workgroup and un-permuted by the epilogue;
//   * the MFMAs are inline asm with the accumulator TIED to the destination ("+a"): left to the register allocator, the
//     accumulator chains of the unrolled loop bodies are renamed into one another and copied back by ~125 v_accvgpr moves
//     per slab; one value of the next slab is split after each of the first MFMAs of a column tile, the loads follow the
//     units that free their registers (a burst of loads stops the wave's issue for the bytes' transfer time), and a
//     scheduling barrier after every MFMA pins that order (without it every fragment read and split is hoisted and spills);
//   * EIGHT waves, two per SIMD, 64 x 64 wave tiles: 128 accumulator registers + two slabs of staging (2 x 24) + one column
//     tile of B fragments ahead fit the 256 registers a wave has at this occupancy.  What binds now is LDS: per slab 128 KB
//     of fragment reads + 48 KB of piece writes per CU at 128 B per clock take as long as the MFMAs.
#ifndef TN8_GROUP
#define TN8_GROUP 3        // MFMAs (with the pair split behind them) per scheduling region; 1 .. 12 measure the same (r04_gemm_tn_ablation.log)
#endif
struct Tn8Regs {
    f32x4 a[2];            // A item: rows 4 arg .. + 3 in the components, k = 2 akp + q
    f32x4 b[4];            // B item: rows 4 brg .. + 3, k = 4 bkg + q
};

__global__ __launch_bounds__(512) void gemm_h3_tn8_kernel(int64_t M, int64_t N, int64_t K, const float* __restrict__ A, int64_t lda,
                                                          const float* __restrict__ B, int64_t ldb, float* __restrict__ C,
                                                          int64_t ldc, int accumulate, int64_t k_per_split,
                                                          float* __restrict__ slabs, const unsigned* __restrict__ a_absmax,
                                                          int tm, int tn, int S) {
    constexpr int BN = 256;
    using TL = H3Tile<BN>;
    constexpr int NTW = 4;                                // 16-column tiles per wave (wave tile 64 x 64)
    extern __shared__ __attribute__((aligned(16))) unsigned short sm16[];
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int G = (S > 1) ? tm * tn : tn, ngroups = (S > 1) ? S : tm;
    const int g = (slot / G) * 8 + xcd, within = slot % G;
    if (g >= ngroups) return;
    const int z = (S > 1) ? g : 0;
    const int mi = (S > 1) ? within % tm : g, ni = (S > 1) ? within / tm : within;
    const int64_t m0 = (int64_t)mi * HM, n0 = (int64_t)ni * BN;
    const int64_t kb = (int64_t)z * k_per_split;
    const int64_t ke = (kb + k_per_split < K) ? kb + k_per_split : K;
    const int nslab = (int)((ke - kb) / HK);

    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = w >> 2, wn = w & 3;
    const int fi = lane & 15, kq = lane >> 4;
    const float sa = block_scale(a_absmax);

    // Which thread stages which (row group, k group) decides the LDS bank pattern of the piece writes (plane rows are 80 bytes =
    // 20 banks apart: eight consecutive rows start on eight different multiples of 4 banks and then repeat).  A b32 write is
    // conflict-free when 32 consecutive lanes are 8 row groups x 4 k pairs, a b64 write when 16 consecutive lanes are 8 row
    // groups x 2 k groups; a load instruction then reads 128-byte segments (8 lanes x 16 bytes) of several k rows.
    const int arg = (lane & 7) | ((lane >> 5) & 1) << 3 | (w & 1) << 4;       // A item: rows 4 arg .. (32 groups)
    const int akl = (lane >> 3) & 3, akp = akl | (w >> 1) << 2;              //         k 2 akp, 2 akp + 1 (16 pairs)
    const int brg = (lane & 7) | ((lane >> 4) & 3) << 3 | (w & 1) << 5;       // B item: rows 4 brg .. (64 groups)
    const int bkl = (lane >> 3) & 1, bkg = bkl | (w >> 1) << 1;              //         k 4 bkg .. (8 groups)
    const float* Bp = B + n0;
    const float* Ap = A + m0;
    const unsigned ldbu = (unsigned)ldb, ldau = (unsigned)lda;
    const unsigned oab = ((unsigned)(2 * akl) * ldau + 4u * (unsigned)arg) * 4u;          // lane part; the wave part is uniform
    const unsigned obb = ((unsigned)(4 * bkl) * ldbu + 4u * (unsigned)brg) * 4u;
    auto load_one = [&](Tn8Regs& r, int slab, int item, int q) {
        slab = slab < nslab ? slab : nslab - 1;
        const int64_t k0 = kb + (int64_t)slab * HK;
        if (item == 0) {
            const char* ua = reinterpret_cast<const char*>(Ap + (k0 + 8 * (w >> 1) + q) * lda);
            r.a[q] = *reinterpret_cast<const f32x4*>(ua + oab);
        } else {
            const char* ub = reinterpret_cast<const char*>(Bp + (k0 + 8 * (w >> 1) + q) * (int64_t)ldbu);
            r.b[q] = *reinterpret_cast<const f32x4*>(ub + obb);
        }
    };
    // The split of TWO values in six mixed-precision FMAs, the halves landing packed in place (common.h's split2h to the bit:
    // every step is exact or the same single rounding): p0 = f16(x s) into the low / high half of P, r = x s - p0 (the fp16
    // operand read from P), p1 = f16(2048 r) into Q.  Against cvt_pk / cvt / pk_add / pk_mul / cvt_pk + register moves: 3
    // instructions per value instead of 4.5, none of them packed-f32 (two passes).
    auto split_pair = [&](float x, float y, float s, unsigned& P, unsigned& Q) {
        float rx, ry;
        const float c2048 = 2048.0f;
        asm("v_fma_mixlo_f16 %0, %4, %6, 0\n\t"
            "v_fma_mixhi_f16 %0, %5, %6, 0\n\t"
            "v_fma_mix_f32 %2, %4, %6, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mix_f32 %3, %5, %6, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixlo_f16 %1, %2, %7, 0\n\t"
            "v_fma_mixhi_f16 %1, %3, %7, 0"
            : "=&v"(P), "=&v"(Q), "=&v"(rx), "=&v"(ry) : "v"(x), "v"(y), "s"(s), "s"(c2048));
    };
    // pairs in commit order: B rows first (8 pairs: row e = pp >> 1, k 2 (pp & 1), + 1), then the A rows (4 pairs: row e = pp - 8)
    uint2 blo, bhi;
    auto commit_pair = [&](const Tn8Regs& r, unsigned short* np, int pp, float sca) {
        if (pp < 8) {
            const int e = pp >> 1, h = pp & 1;
            unsigned P, Q;
            split_pair(r.b[2 * h][e], r.b[2 * h + 1][e], 1.0f, P, Q);
            if (h == 0) { blo.x = P; bhi.x = Q; }
            else {
                blo.y = P; bhi.y = Q;
                unsigned short* d = np + 2 * TL::A_PLANE + (e * (BN / 4) + brg) * HKP + 4 * bkg;
                *reinterpret_cast<uint2*>(d) = blo;
                *reinterpret_cast<uint2*>(d + TL::B_PLANE) = bhi;
            }
        } else {
            const int e = pp - 8;
            unsigned P, Q;
            split_pair(r.a[0][e], r.a[1][e], sca, P, Q);
            unsigned short* d = np + (e * (HM / 4) + arg) * HKP + 2 * akp;
            *reinterpret_cast<unsigned*>(d) = P;
            *reinterpret_cast<unsigned*>(d + TL::A_PLANE) = Q;
        }
    };

    f32x4 acc0[4][NTW], acc1[4][NTW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            acc0[i][j] = acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+a"(acc0[i][j]), "+a"(acc1[i][j]));
        }
#define TN_MFMA(ACC, FA, FB) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(FA), "v"(FB))

    // one slab: 4 column tiles x 12 MFMAs; a value of slab sl + 1 is split after each of the first six MFMAs of a tile (B's 16
    // values, then A's 8); B's four loads of slab sl + 3 follow in tile 2 once its rows are committed, A's two in tile 3
    auto slab_body = [&](Tn8Regs& r, int buf, int sl) {
        const unsigned short* bp = sm16 + buf * TL::BUF;
        unsigned short* np = sm16 + (buf ^ 1) * TL::BUF;
        const float sca = (sl + 1) < nslab ? sa : 0.f;
        const unsigned short* a0p = bp + (wm * 64 + fi) * HKP + 8 * kq;
        const unsigned short* b0p = bp + 2 * TL::A_PLANE + (wn * 64 + fi) * HKP + 8 * kq;
        f16x8 a0[4], a1[4], bq[2][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a0[i] = *reinterpret_cast<const f16x8*>(a0p + i * 16 * HKP);
            a1[i] = *reinterpret_cast<const f16x8*>(a0p + TL::A_PLANE + i * 16 * HKP);
        }
        bq[0][0] = *reinterpret_cast<const f16x8*>(b0p);
        bq[0][1] = *reinterpret_cast<const f16x8*>(b0p + TL::B_PLANE);
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            if (j + 1 < NTW) {
                bq[(j + 1) & 1][0] = *reinterpret_cast<const f16x8*>(b0p + (j + 1) * 16 * HKP);
                bq[(j + 1) & 1][1] = *reinterpret_cast<const f16x8*>(b0p + TL::B_PLANE + (j + 1) * 16 * HKP);
            }
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                const int i = m & 3;
                if (m < 4) TN_MFMA(acc0[i][j], a0[i], bq[j & 1][0]);
                else if (m < 8) TN_MFMA(acc1[i][j], a0[i], bq[j & 1][1]);
                else TN_MFMA(acc1[i][j], a1[i], bq[j & 1][0]);
                if (m < 6 && (m & 1)) commit_pair(r, np, 3 * j + (m >> 1), sca);
                // B's pairs are 0..7: the last one is committed in tile 2 at m = 3; A's (8..11) in tile 3 at m = 5
                if (j == 2 && m >= 4 && (m & 1) == 0) load_one(r, sl + 3, 1, (m - 4) >> 1);          // m = 4, 6, 8, 10
                if (j == 2 && m == 11) load_one(r, sl + 3, 1, 3) ;
                if (j == 3 && (m == 8 || m == 10)) load_one(r, sl + 3, 0, (m - 8) >> 1);
                if (m % TN8_GROUP == TN8_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    Tn8Regs r0, r1;
#pragma unroll
    for (int q = 0; q < 4; ++q) { load_one(r0, 0, 1, q); load_one(r1, 1, 1, q); }
#pragma unroll
    for (int q = 0; q < 2; ++q) { load_one(r0, 0, 0, q); load_one(r1, 1, 0, q); }
#pragma unroll
    for (int pp = 0; pp < 12; ++pp) commit_pair(r0, sm16, pp, sa);
#pragma unroll
    for (int q = 0; q < 4; ++q) load_one(r0, 2, 1, q);
#pragma unroll
    for (int q = 0; q < 2; ++q) load_one(r0, 2, 0, q);
    __syncthreads();
    // iteration s: slab s out of buffer s & 1, slab s + 1 into the other (from the register set that then takes slab s + 3);
    // one rolled loop of two bodies, padded to an even count with zero slabs
    for (int s = 0; s < nslab; s += 2) {
        slab_body(r1, 0, s);
        __syncthreads();
        slab_body(r0, 1, s + 1);
        __syncthreads();
    }
#undef TN_MFMA
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const float inv = 1.0f / sa;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int pc = wn * 64 + j * 16 + fi;
            const int64_t col = n0 + 4 * (pc & (BN / 4 - 1)) + pc / (BN / 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pr = wm * 64 + i * 16 + 4 * kq + r;
                const int64_t row = m0 + 4 * (pr & (HM / 4 - 1)) + pr / (HM / 4);
                const float val = (acc0[i][j][r] + H3_LO * acc1[i][j][r]) * inv;
                if (slabs) {
                    slabs[((int64_t)z * M + row) * N + col] = val;
                } else {
                    float* cp = C + row * ldc + col;
                    *cp = (accumulate ? *cp : 0.f) + val;
                }
            }
        }
}

__global__ __launch_bounds__(256) void h3_splitk_reduce_kernel(const float* __restrict__ slabs, int S, int64_t MN, int64_t N,
                                                               float* __restrict__ C, int64_t ldc,
                                                               const float* __restrict__ bias, int accumulate) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= MN) return;
    float4 s = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < S; ++z) {
        const float4 v = *reinterpret_cast<const float4*>(slabs + (int64_t)z * MN + i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int64_t row = i / N, col = i % N;               // N % 4 == 0: the four values share a row
    float* c = C + row * ldc + col;
    const float o[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = (accumulate ? c[j] : 0.f) + o[j] + (bias ? bias[col + j] : 0.f);
}

// split-K plan shared by the two kernels: whole groups of 8 splits (one per XCD) when the tile grid cannot fill the chip
static void h3_plan(uav_ctx* ctx, int64_t M, int64_t N, int64_t K, int64_t tiles, int64_t& S, int64_t& kps) {
    S = 1;
    if (tiles < ctx->num_cu && K >= 64 * HK) {
        S = (ctx->num_cu + tiles - 1) / tiles;
        S = (S + 7) / 8 * 8;
        const int64_t max_k = K / (16 * HK), max_ws = (int64_t)(ctx->ws_bytes / sizeof(float)) / (M * N);
        if (S > max_k) S = max_k;
        if (S > max_ws) S = max_ws;
        if (S < 1) S = 1;
    }
    kps = ((K + S - 1) / S + HK - 1) / HK * HK;
    S = (K + kps - 1) / kps;
}

// dW-shaped products (see gemm_h3_tn8_kernel): A [K][M] and B [K][N] contiguous along their rows
int launch_h3_tn(uav_ctx* ctx, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                 float* C, int64_t ldc, int accumulate, const unsigned* a_absmax, hipStream_t st) {
    constexpr int BN = 256;
    const int tm = (int)(M / HM), tn = (int)(N / BN);
    const int64_t tiles = (int64_t)tm * tn;
    int64_t S, kps;
    h3_plan(ctx, M, N, K, tiles, S, kps);
    float* slabs = (S > 1) ? (float*)ctx->ws : nullptr;
    const int64_t G = (S > 1) ? tiles : tn, ngroups = (S > 1) ? S : tm;
    const int64_t grid = 8 * G * ((ngroups + 7) / 8);
    UAV_REQUIRE(grid < (1ll << 31), "gemm_h3: grid too large");
    auto kern = gemm_h3_tn8_kernel;
    UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(kern), (int)H3Tile<BN>::LDS));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), H3Tile<BN>::LDS, st, M, N, K, A, lda, B, ldb, C, ldc, accumulate, kps,
                       slabs, a_absmax, tm, tn, (int)S);
    if (S > 1) {
        const int64_t MN = M * N, nb = (MN / 4 + 255) / 256;
        hipLaunchKernelGGL(h3_splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, slabs, (int)S, MN, N, C, ldc,
                           (const float*)nullptr, accumulate);
    }
    UAV_LAUNCH_CHECK();
    return 0;
}

template <int BN, bool AKC, bool BKC>
int launch_h3(uav_ctx* ctx, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
              int64_t ldc, const float* bias, int accumulate, const unsigned* a_absmax, hipStream_t st) {
    const int tm = (int)(M / HM), tn = (int)(N / BN);
    const int64_t tiles = (int64_t)tm * tn;
    int64_t S, kps;
    h3_plan(ctx, M, N, K, tiles, S, kps);
    float* slabs = (S > 1) ? (float*)ctx->ws : nullptr;
    const int64_t G = (S > 1) ? tiles : tn, ngroups = (S > 1) ? S : tm;
    const int64_t grid = 8 * G * ((ngroups + 7) / 8);
    UAV_REQUIRE(grid < (1ll << 31), "gemm_h3: grid too large");
    auto kern = gemm_h3_kernel<BN, AKC, BKC>;
    UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(kern), (int)H3Tile<BN>::LDS));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), H3Tile<BN>::LDS, st, M, N, K, A, lda, B, ldb, C, ldc, bias,
                       accumulate, kps, slabs, a_absmax, tm, tn, (int)S);
    if (S > 1) {
        const int64_t MN = M * N, nb = (MN / 4 + 255) / 256;
        hipLaunchKernelGGL(h3_splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, slabs, (int)S, MN, N, C, ldc, bias,
                           accumulate);
    }
    UAV_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// Shapes the split kernel takes: M a multiple of 128, N of 128, each operand contiguous along k or along its row index,
// 16-byte aligned where it is read by float4.  Everything else stays on the exact-f32 kernels (gemm.hip).
bool gemm_h3_ok(int64_t M, int64_t N, int64_t K, const float* A, int64_t sa_m, int64_t sa_k, const float* B, int64_t sb_k,
                int64_t sb_n) {
    if (M < HM || M % HM || N < 128 || N % 128 || K < HK) return false;
    if (sa_k != 1 && sa_m != 1) return false;
    if (sa_m >= (1 << 23) || sa_k >= (1 << 23) || sb_n >= (1 << 23) || sb_k >= (1 << 23)) return false;   // 32-bit lane offsets
    if (sb_k != 1 && sb_n != 1) return false;
    if (sa_k == 1 && ((sa_m & 3) || (reinterpret_cast<uintptr_t>(A) & 15))) return false;
    if (sb_k == 1 && ((sb_n & 3) || (reinterpret_cast<uintptr_t>(B) & 15))) return false;
    return true;
}

// The dW shape (both operands contiguous along their row index, whole 32-row slabs, 256-column tiles, dwordx4 loads) goes to
// gemm_h3_tn8_kernel unless UAV_DEBUG_GEMM_TN_OFF asks for the older kernel (same results bit for bit: an A/B switch).
bool gemm_h3_tn_ok(int64_t M, int64_t N, int64_t K, const float* A, int64_t sa_m, int64_t sa_k, const float* B, int64_t sb_k,
                   int64_t sb_n) {
    if (uav_debug(UAV_DEBUG_GEMM_TN_OFF)) return false;
    if (sa_m != 1 || sb_n != 1 || sa_k == 1 || sb_k == 1) return false;
    if (M < HM || M % HM || N < 256 || N % 256 || K < HK || K % HK) return false;
    if ((sa_k & 3) || (sb_k & 3) || sa_k >= (1 << 23) || sb_k >= (1 << 23)) return false;
    return ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0;
}

int gemm_h3(uav_ctx* ctx, int64_t M, int64_t N, int64_t K, const float* A, int64_t sa_m, int64_t sa_k, const float* B,
            int64_t sb_k, int64_t sb_n, float* C, int64_t ldc, const float* bias, int accumulate, const unsigned* a_absmax,
            hipStream_t st) {
    UAV_REQUIRE(ctx && A && B && C, "gemm_h3: NULL argument");
    UAV_REQUIRE(gemm_h3_ok(M, N, K, A, sa_m, sa_k, B, sb_k, sb_n), "gemm_h3: shape / strides not supported (M %% 128, N %% 128, "
                "unit stride along k or along the row index of each operand)");
    const bool akc = sa_k == 1, bkc = sb_k == 1;
    const int64_t lda = akc ? sa_m : sa_k, ldb = bkc ? sb_n : sb_k;
    const bool wide = N % 256 == 0;
    if (!bias && gemm_h3_tn_ok(M, N, K, A, sa_m, sa_k, B, sb_k, sb_n))
        return launch_h3_tn(ctx, M, N, K, A, lda, B, ldb, C, ldc, accumulate, a_absmax, st);
#define H3_GO(BN, AK, BK_) return launch_h3<BN, AK, BK_>(ctx, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, a_absmax, st)
    if (wide) {
        if (akc && bkc) H3_GO(256, true, true);
        if (akc) H3_GO(256, true, false);
        if (bkc) H3_GO(256, false, true);
        H3_GO(256, false, false);
    }
    if (akc && bkc) H3_GO(128, true, true);
    if (akc) H3_GO(128, true, false);
    if (bkc) H3_GO(128, false, true);
    H3_GO(128, false, false);
#undef H3_GO
}

extern "C" int uav_gemm_f16x3(uav_ctx* ctx, int64_t M, int64_t N, int64_t K, const float* A, int64_t sa_m, int64_t sa_k,
                              const float* B, int64_t sb_k, int64_t sb_n, float* C, int64_t ldc, const float* bias,
                              int accumulate, const float* a_absmax, uav_stream stream) {
    UAV_REQUIRE(ctx, "uav_gemm_f16x3: NULL handle");
    uav_enter(ctx);
    return gemm_h3(ctx, M, N, K, A, sa_m, sa_k, B, sb_k, sb_n, C, ldc, bias, accumulate,
                   reinterpret_cast<const unsigned*>(a_absmax), as_stream(stream));
}

// wgrad_pc.hip -- weight gradients of the h = 256 fp16-split step path straight from the BPTT's piece chunks (common.h:
// DgPack): the gate gradients exist ONCE, as the two fp16 pieces per row the recurrent product consumed, scaled per
// (env, step) by a power of two, in MFMA fragment order -- no f32 copy is written or read.
//
//   colsum_pc_kernel    db = sum_k dG[k][:]  (+ dW_ih = dG^T x for a narrow input, I <= 8, exact f32 FMAs) and max isc: one
//                       HBM-bound pass, 1 KB chunk per wave load.
//   gemm_pc_kernel      dW = dG^T [h_prev | x]  as three fp16 piece products (gemm_h3.hip's arithmetic, tile and wave layout):
//                       * K = (env, step) walks 32-env slabs (two 16-env row tiles) with the step index innermost, so a workgroup's
//                         B rows advance by one row per slab and A comes in two 8 KB runs; the h_prev operand is read as rows of the
//                         layer's OUTPUT y one step back (h0 at step 0) with the restart mask folded into its scales (iscm), so the
//                         forward pass writes no h_prev slot into the stash;
//                       * A = the piece chunks AS STORED: LDS-DMA (global_load_lds_dwordx4, no registers, no VALU), one 1 KB chunk
//                         per wave instruction, the 16-byte granules permuted through the per-lane GLOBAL address so that the
//                         transposed fragment reads (ds_read_b64_tr_b16: the chunks hold [env][gate row], the product sums over
//                         envs) are conflict-free; three stages, the DMA two slabs ahead;
//                       * the per-(env, step) scales cannot ride on A (its bytes never pass a register): they ride on B --
//                         B' = h_prev * (isc_k / max isc) is what gets split, exact powers of two, B' within fp16's range because
//                         |h| < 1; the epilogue multiplies by max isc.  An (env, step) whose gradients are 2^-28 of the largest
//                         loses relative -- not absolute -- accuracy, as under gemm_h3's one block scale;
//                       * B: f32 rows -> registers (two slabs ahead) -> split -> LDS piece planes, as gemm_h3_tn8_kernel.
//                       Split-K over contiguous slab ranges, all tiles of a range on one XCD; deterministic slab reduce.
// Results differ from the round-4 path (f32 rows through gemm_h3_tn8_kernel) only in where the power-of-two scale is applied.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef PC_ABL
#define PC_ABL 0        // ablation builds of gemm_pc_kernel only (tools/ab_wgrad_pc.sh; wrong results, same launches): 1 no MFMAs, 2 no B
#endif                  // loads in the loop, 3 no DMA in the loop, 4 no split / piece writes

namespace {

constexpr int H = DgPack::H, NS = DgPack::NS, G4 = 4 * H;

__device__ __forceinline__ float h2f(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }

// ------------------------------------------------------------------------------------------------ unpack (tests, fallbacks)
// out f32 [N][T][4H]: a wave per (step, row tile, slab) chunk pair, lane (kq, env) converts its 8 gate rows
__global__ __launch_bounds__(256) void dg_unpack_kernel(const unsigned short* __restrict__ pieces, const float* __restrict__ isc, int N,
                                                        int T, int NP, int RT, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, env = lane & 15, kq = lane >> 4;
    const int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);          // (t * RT + rt) * NS + s
    if (chunk >= (int64_t)T * RT * NS) return;
    const int s = (int)(chunk % NS);
    const int64_t tile = chunk / NS;
    const int rt = (int)(tile % RT), t = (int)(tile / RT), n = rt * 16 + env;
    if (n >= N) return;
    const uint4 a = *reinterpret_cast<const uint4*>(pieces + chunk * 1024 + lane * 8);
    const uint4 b = *reinterpret_cast<const uint4*>(pieces + chunk * 1024 + 512 + lane * 8);
    const float sc = isc[(size_t)t * NP + n];
    const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned short p0 = (unsigned short)(aw[i >> 1] >> (16 * (i & 1))), p1 = (unsigned short)(bw[i >> 1] >> (16 * (i & 1)));
        v[i] = (h2f(p0) + h2f(p1) * H3_LO) * sc;
    }
    float* o = out + ((size_t)n * T + t) * G4 + 32 * s + 8 * kq;
    *reinterpret_cast<float4*>(o) = float4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<float4*>(o + 4) = float4{v[4], v[5], v[6], v[7]};
}

// ------------------------------------------------------------------------------------------------ column sums (+ dG^T x)
// grid (nb, 2): workgroup (b, half) sums tiles [b tpb, (b + 1) tpb) x slabs 16 half .. + 15; 8 waves, wave w slabs 2 w, 2 w + 1
// of the half; lane (kq, env) keeps 2 x 8 gate rows x (1 + I) sums in registers over all its tiles, the 16 env lanes are
// folded at the end.  partial [nb][1 + I][4H].
template <int I>
__global__ __launch_bounds__(512) void colsum_pc_kernel(const unsigned short* __restrict__ pieces, const float* __restrict__ isc, int N,
                                                        int T, int NP, int RT, int64_t tiles, int tpb, const float* __restrict__ x,
                                                        float* __restrict__ partial, unsigned* __restrict__ iscmax_bits) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, env = lane & 15, kq = lane >> 4;
    const int s0 = 16 * blockIdx.y + 2 * w;
    float acc[2][8][1 + I];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int c = 0; c <= I; ++c) acc[h][i][c] = 0.f;
    float mx = 0.f;
    const int64_t tb = (int64_t)blockIdx.x * tpb, te = tb + tpb < tiles ? tb + tpb : tiles;
    auto fetch = [&](int64_t tile, uint4 (&a)[2], uint4 (&b)[2]) {
        const unsigned short* p = pieces + (tile * NS + s0) * 1024 + lane * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            a[h] = *reinterpret_cast<const uint4*>(p + h * 1024);
            b[h] = *reinterpret_cast<const uint4*>(p + h * 1024 + 512);
        }
    };
    uint4 a[2], b[2], an[2], bn[2];
    if (tb < te) fetch(tb, a, b);
    for (int64_t tile = tb; tile < te; ++tile) {
        if (tile + 1 < te) fetch(tile + 1, an, bn);
        const int rt = (int)(tile % RT), t = (int)(tile / RT), n = rt * 16 + env;
        const float sc = isc[(size_t)t * NP + n];
        mx = fmaxf(mx, sc);
        float xv[I > 0 ? I : 1];
        if (I > 0) {
            const float* xp = x + ((size_t)(n < N ? n : N - 1) * T + t) * I;
#pragma unroll
            for (int c = 0; c < I; ++c) xv[c] = xp[c];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned aw[4] = {a[h].x, a[h].y, a[h].z, a[h].w}, bw[4] = {b[h].x, b[h].y, b[h].z, b[h].w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned short p0 = (unsigned short)(aw[i >> 1] >> (16 * (i & 1))), p1 = (unsigned short)(bw[i >> 1] >> (16 * (i & 1)));
                const float g = (h2f(p0) + h2f(p1) * H3_LO) * sc;
                acc[h][i][0] += g;
#pragma unroll
                for (int c = 0; c < I; ++c) acc[h][i][1 + c] = fmaf(g, xv[c], acc[h][i][1 + c]);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) { a[h] = an[h]; b[h] = bn[h]; }
    }
    // fold the 16 env lanes of each kq group (fixed xor tree), lane env == 0 writes
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int c = 0; c <= I; ++c) {
                float v = acc[h][i][c];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                if (env == 0) partial[((size_t)blockIdx.x * (1 + I) + c) * G4 + 32 * (s0 + h) + 8 * kq + i] = v;
            }
    if (iscmax_bits && blockIdx.y == 0 && w == 0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (lane == 0) atomicMax(iscmax_bits, __float_as_uint(mx));              // non-negative floats order like their bits
    }
}

// partial [nb][1 + I][4H] -> db [4H] (and db2, the same values) and dw_ih [4H][I]; fixed association.  A block = 32 columns x 8
// groups of partial blocks (thread (g, col) sums blocks g, g + 8, ...; the eight group sums are added in order through LDS):
// one thread per column walked all 256 partial blocks one dependent load after the other, 96 us for 9,216 sums.
__global__ __launch_bounds__(256) void colsum_pc_reduce_kernel(const float* __restrict__ partial, int nb, int I, float* __restrict__ db,
                                                               float* __restrict__ db2, float* __restrict__ dw_ih) {
    __shared__ float sm[8][32];
    const int col = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + col;                       // (1 + I) * 4H is a multiple of 32
    const int c = idx / G4, m = idx % G4;
    float s = 0.f;
    for (int b = g; b < nb; b += 8) s += partial[((size_t)b * (1 + I) + c) * G4 + m];
    sm[g][col] = s;
    __syncthreads();
    if (g != 0) return;
#pragma unroll
    for (int k = 1; k < 8; ++k) s += sm[k][col];
    if (c == 0) { db[m] = s; if (db2) db2[m] = s; }
    else dw_ih[(size_t)m * I + (c - 1)] = s;
}

// max isc alone (when the column sums come from elsewhere)
__global__ __launch_bounds__(256) void isc_max_kernel(const float* __restrict__ isc, int64_t n, unsigned* __restrict__ iscmax_bits) {
    float mx = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) mx = fmaxf(mx, isc[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(iscmax_bits, __float_as_uint(mx));
}

// ------------------------------------------------------------------------------------------------ the product
constexpr int PM = 128, PN = 256, PK = 32, PKP = PK + 8;       // tile rows (gate rows), tile columns, slab depth (envs), padded k
constexpr int A_STAGE = 16 * 1024, A_STAGES = 3;                // bytes: [row tile 2][slab 4][piece 2] chunks of 1 KB
constexpr int B_PLANE = PN * PKP;                               // halves per piece plane
constexpr int B_BUF = 2 * B_PLANE * 2;                          // bytes per buffer (two planes)
constexpr size_t PC_LDS = (size_t)A_STAGES * A_STAGE + 2 * (size_t)B_BUF;      // 48 + 80 = 128 KB

struct PcArgs {
    const unsigned short* pieces;
    const float* isc;
    const unsigned* iscmax_bits;
    int N, T, NP, RT, SPT;             // SPT = NP / 32 slabs per step
    const float* iscm;                 // isc * keep: the scales of the h_prev columns
    const float* y; const float* h0;   // columns [0, n_split): h_prev row (env, t) = y + (env T + t - 1) H for t > 0, h0 + env H at t = 0
                                       // (unmasked: keep[env][t] rides on iscm)
    const float* B1; int64_t ldb1;     // columns [n_split, Ntot): row (env, t) at B1 + (env T + t) ldb1
    int n_split, Ntot;
    float* slabs;                      // [S][4H][Ntot]
    int64_t sps, total;                // slabs per split, all slabs (SPT * T)
    int tm, tn, S;
};

struct PcRegs {
    f32x4 b[4];            // B item: columns 4 brg .. + 3 in the components, env 4 bkg + q
    f32x4 sc;              // isc of those four envs (this slab's step)
};

__global__ __launch_bounds__(512) void gemm_pc_kernel(const PcArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char pc_lds[];
    typedef __attribute__((address_space(3))) unsigned char lds_b;
    typedef short v4s __attribute__((__vector_size__(4 * sizeof(short))));
    typedef short v8s __attribute__((__vector_size__(8 * sizeof(short))));
    typedef __attribute__((address_space(3))) v4s lds_v4s;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int G = a.tm * a.tn;
    const int z = (slot / G) * 8 + xcd, within = slot % G;
    if (z >= a.S) return;
    const int mi = within % a.tm, ni = within / a.tm;
    const int64_t q0 = (int64_t)z * a.sps;
    const int64_t q1 = q0 + a.sps < a.total ? q0 + a.sps : a.total;
    const int nslab = (int)(q1 - q0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;
    const int fi = lane & 15, kq = lane >> 4;
    const int T = a.T;
    const int n0 = ni * PN;
    const bool second = n0 >= a.n_split;
    const float* Bp = second ? a.B1 + (n0 - a.n_split) : a.y + n0;
    const float* Hp = a.h0 + n0;                                            // (first operand only)
    const int64_t ldb = second ? a.ldb1 : (int64_t)H;
    const float* sc_arr = second ? a.isc : a.iscm;
    const float iscmax = __uint_as_float(*a.iscmax_bits);
    const float to_block = 1.0f / iscmax;                                   // a power of two: exact

    unsigned char* a_lds = pc_lds;
    unsigned short* b_lds = reinterpret_cast<unsigned short*>(pc_lds + A_STAGES * A_STAGE);
    const unsigned a_lds_base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)((lds_b*)a_lds));

    // ---- A: LDS-DMA.  Wave w, instruction r: chunk (row tile 2 j + r, slab 4 mi + (w >> 1), piece w & 1) -> stage chunk 8 r + w.
    // Lane L lands at granule L of the chunk image and fetches granule (o, e) = (L >> 4, (L & 15) ^ 4 (o & 1)): the image is
    // [octet o][env e ^ 4 (o & 1)][8 gate rows], which spreads the transposed reads' 32 lanes over all 64 banks.
    const unsigned dma_lane = (unsigned)(((lane >> 4) * 16 + ((lane & 15) ^ (((lane >> 4) & 1) << 2))) * 8);     // halves
    const int64_t step_halves = (int64_t)a.RT * NS * 1024;
    const unsigned short* dma_base = a.pieces + (int64_t)(4 * mi + (w >> 1)) * 1024 + (w & 1) * 512 + dma_lane;
    // slab positions (row-tile pair j, step t): divisions only here, the loop advances them; past the split's end they stay on
    // its last slab (valid addresses; such a slab's B scales are zero)
    struct Pos { int j, t; };
    auto pos_of = [&](int64_t q) { q = q < q1 ? q : q1 - 1; return Pos{(int)(q / T), (int)(q % T)}; };
    const int jt_last_j = (int)((q1 - 1) / T), jt_last_t = (int)((q1 - 1) % T);
    auto advance = [&](Pos& p) {
        if (p.j == jt_last_j && p.t == jt_last_t) return;
        if (++p.t == T) { p.t = 0; ++p.j; }
    };
    auto issue_dma = [&](const Pos& p, int stage) {
        const int j = p.j, t = p.t;
        const unsigned short* s0 = dma_base + (int64_t)t * step_halves + (int64_t)(2 * j) * NS * 1024;
        const unsigned short* s1 = s0 + (int64_t)NS * 1024;
        const unsigned dst = a_lds_base + (unsigned)(stage * A_STAGE + w * 1024);
        unsigned m0save;
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                     "s_add_u32 m0, m0, 8192\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(m0save) : "v"(s0), "v"(s1), "s"(dst) : "memory", "scc");
    };
    // transposed fragment reads: lane (kq, li = 4 q + p) supplies env 8 (kq & 1) + q (+ 4: the `hi` read) of row tile kq >> 1,
    // gate rows 4 p .. 4 p + 3 of the 16-row tile; octet 2 (i & 1) + (p >> 1)
    const int tq = (lane & 15) >> 2, tp = lane & 3;
    const unsigned a_lo = (unsigned)((kq >> 1) * 8192 + (((tp >> 1) * 16 + ((8 * (kq & 1) + tq) ^ ((tp >> 1) << 2))) * 16) + 8 * (tp & 1));
    // (32-bit LDS addresses: one base register per read kind, the fragment's (row tile, piece) as an instruction offset)
    auto a_frag = [&](unsigned stage_lo, unsigned stage_hi, int i, int piece) -> f16x8 {
        const unsigned c = (unsigned)((i >> 1) * 2048 + piece * 1024 + (i & 1) * 512);
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_v4s*>(stage_lo + c));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_v4s*>(stage_hi + c));
        const v8s r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(f16x8, r);
    };
    const unsigned a_rd0 = a_lds_base + a_lo + (unsigned)(wm * 4096);      // stage 0, this lane's `lo` read; stages are A_STAGE apart (bit 6 untouched)

    // ---- B: as gemm_h3_tn8_kernel (thread = 4 columns x 4 envs, dwordx4 along the row), the envs of a slab being rows
    // (32 j + k) T + t of the operand; a slab that reaches past N (ragged last tile) clamps per lane
    const int brg = (lane & 7) | ((lane >> 4) & 3) << 3 | (w & 1) << 5;       // columns 4 brg .. (64 groups)
    const int bkl = (lane >> 3) & 1, bkg = bkl | (w >> 1) << 1;              // envs 4 bkg .. (8 groups)
    const unsigned obb = (unsigned)(((int64_t)(4 * bkl) * T * ldb + 4 * brg) * 4);
    const unsigned obh = (unsigned)(((int64_t)(4 * bkl) * H + 4 * brg) * 4);            // rows of h0 (step 0 of the h_prev operand)
    auto load_b = [&](PcRegs& r, const Pos& p, int qq) {
        const int j = p.j, t = p.t;
        const int env = 32 * j + 8 * (w >> 1) + qq;                          // uniform; the lane adds 4 bkl
        const bool from_h0 = !second && t == 0;                              // uniform
        const int tr = second ? t : t - 1;
        if (env + 4 < a.N) {
            const char* ub = reinterpret_cast<const char*>(from_h0 ? Hp + (int64_t)env * H : Bp + ((int64_t)env * T + tr) * ldb);
            r.b[qq] = *reinterpret_cast<const f32x4*>(ub + (from_h0 ? obh : obb));
        } else {
            int e = env + 4 * bkl;
            e = e < a.N ? e : a.N - 1;
            r.b[qq] = *reinterpret_cast<const f32x4*>((from_h0 ? Hp + (int64_t)e * H : Bp + ((int64_t)e * T + tr) * ldb) + 4 * brg);
        }
    };
    auto load_sc = [&](PcRegs& r, const Pos& p, bool live) {
        const int j = p.j, t = p.t;
        const f32x4 v = *reinterpret_cast<const f32x4*>(sc_arr + (size_t)t * a.NP + 32 * j + 4 * bkg);
        r.sc = v * (live ? to_block : 0.f);                                   // a slab past the split's end multiplies by zero
    };
    // the split of two values with their own scales (gemm_h3.hip: split_pair)
    auto split_pair = [&](float x, float y, float sx, float sy, unsigned& P, unsigned& Q) {
        float rx, ry;
        const float c2048 = 2048.0f;
        asm("v_fma_mixlo_f16 %0, %4, %6, 0\n\t"
            "v_fma_mixhi_f16 %0, %5, %7, 0\n\t"
            "v_fma_mix_f32 %2, %4, %6, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mix_f32 %3, %5, %7, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixlo_f16 %1, %2, %8, 0\n\t"
            "v_fma_mixhi_f16 %1, %3, %8, 0"
            : "=&v"(P), "=&v"(Q), "=&v"(rx), "=&v"(ry) : "v"(x), "v"(y), "v"(sx), "v"(sy), "s"(c2048));
    };
    uint2 blo, bhi;
    auto commit_pair = [&](const PcRegs& r, unsigned short* np, int pp) {          // pp 0..7: column e = pp >> 1, envs 2 (pp & 1), + 1
        const int e = pp >> 1, h = pp & 1;
        unsigned P, Q;
        split_pair(r.b[2 * h][e], r.b[2 * h + 1][e], r.sc[2 * h], r.sc[2 * h + 1], P, Q);
        if (h == 0) { blo.x = P; bhi.x = Q; }
        else {
            blo.y = P; bhi.y = Q;
            unsigned short* d = np + (e * (PN / 4) + brg) * PKP + 4 * bkg;
            *reinterpret_cast<uint2*>(d) = blo;
            *reinterpret_cast<uint2*>(d + B_PLANE) = bhi;
        }
    };

    constexpr int NTW = 4;
    f32x4 acc0[4][NTW], acc1[4][NTW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            acc0[i][j] = acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+a"(acc0[i][j]), "+a"(acc1[i][j]));
        }
#define PC_MFMA(ACC, FA, FB) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(FA), "v"(FB))

    // One slab.  Its A fragments and the first column tile's B fragments are ALREADY in registers (read in the tail of the
    // previous body).  The DMA of slab sl + 2 goes out first; column tiles 0 .. 2: 12 MFMAs each, a value pair of slab sl + 1 split
    // after MFMAs 1, 3, 5 (8 pairs), B's loads of slab sl + 3 behind them; then the ONE barrier of the slab (behind it slab sl + 1's
    // stage and piece planes are complete, and nobody reads slab sl's any more: the last tile's operands are in registers); the
    // last column tile runs row-tile pair by row-tile pair and re-fills each fragment register with slab sl + 1's as it falls free,
    // so the LDS latency of the next slab's first reads hides under this slab's last 12 MFMAs instead of in front of the first.
    Pos pa = pos_of(q0 + 2), pb = pos_of(q0 + 3);       // the DMA's next slab (sl + 2) and B's (sl + 3)
    f16x8 a0[4], a1[4], bq[2][2];
    auto read_frags = [&](int stage, int buf) {
        const unsigned as = a_rd0 + (unsigned)(stage * A_STAGE), ah = as ^ 64u;
        const unsigned short* b0p = b_lds + buf * (B_BUF / 2) + (wn * 64 + fi) * PKP + 8 * kq;
#pragma unroll
        for (int i = 0; i < 4; ++i) { a0[i] = a_frag(as, ah, i, 0); a1[i] = a_frag(as, ah, i, 1); }
        bq[0][0] = *reinterpret_cast<const f16x8*>(b0p);
        bq[0][1] = *reinterpret_cast<const f16x8*>(b0p + B_PLANE);
    };
    auto slab_body = [&](PcRegs& r, int buf, int sl) {
#if PC_ABL != 3
        issue_dma(pa, (sl + 2) % A_STAGES);
#endif
        unsigned short* np = b_lds + (buf ^ 1) * (B_BUF / 2);
        const unsigned short* b0p = b_lds + buf * (B_BUF / 2) + (wn * 64 + fi) * PKP + 8 * kq;
#pragma unroll
        for (int j = 0; j < NTW - 1; ++j) {
            bq[(j + 1) & 1][0] = *reinterpret_cast<const f16x8*>(b0p + (j + 1) * 16 * PKP);
            bq[(j + 1) & 1][1] = *reinterpret_cast<const f16x8*>(b0p + B_PLANE + (j + 1) * 16 * PKP);
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                const int i = m & 3;
#if PC_ABL != 1
                if (m < 4) PC_MFMA(acc0[i][j], a0[i], bq[j & 1][0]);
                else if (m < 8) PC_MFMA(acc1[i][j], a0[i], bq[j & 1][1]);
                else PC_MFMA(acc1[i][j], a1[i], bq[j & 1][0]);
#else
                asm volatile("" :: "v"(a0[i]), "v"(a1[i]), "v"(bq[j & 1][0]), "v"(bq[j & 1][1]));
#endif
#if PC_ABL != 4
                if (m < 6 && (m & 1) && 3 * j + (m >> 1) < 8) commit_pair(r, np, 3 * j + (m >> 1));
#else
                if (m == 0 && j == 0) asm volatile("" :: "v"(r.b[0]), "v"(r.b[1]), "v"(r.b[2]), "v"(r.b[3]), "v"(r.sc));
#endif
#if PC_ABL != 2
                if (j == 2 && m >= 4 && (m & 1) == 0) load_b(r, pb, (m - 4) >> 1);          // m = 4, 6, 8, 10
#endif
                if (m % 3 == 2) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // vector-memory operations of this wave behind slab sl + 2's B loads and scales (previous body): this body's DMA (2) and
        // four B loads -- with those six in flight, slab sl + 1's DMA (older still) has landed as well
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        lds_barrier();
        {
            constexpr int j = NTW - 1;
            // (opaque to the optimiser: left alone it re-associates base + stage + offset, hoists one address register per
            //  (stage-independent) fragment offset out of the loop and spills them)
            unsigned as = a_rd0 + (unsigned)(((sl + 1) % A_STAGES) * A_STAGE), ah = as ^ 64u;
            asm volatile("" : "+v"(as), "+v"(ah));
            const unsigned short* n0p = np + (wn * 64 + fi) * PKP + 8 * kq;
            bq[0][0] = *reinterpret_cast<const f16x8*>(n0p);                    // free since tile 2
            bq[0][1] = *reinterpret_cast<const f16x8*>(n0p + B_PLANE);
            // (the empty asm statements with a memory clobber pin the reads in the IR: the MFMA statements clobber nothing, and
            //  the first body's reads -- whose uses are further down the same iteration -- were sunk behind all twelve of them)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i0 = 2 * h, i1 = 2 * h + 1;
#if PC_ABL != 1
                PC_MFMA(acc0[i0][j], a0[i0], bq[1][0]);
                PC_MFMA(acc0[i1][j], a0[i1], bq[1][0]);
                PC_MFMA(acc1[i0][j], a0[i0], bq[1][1]);
                PC_MFMA(acc1[i1][j], a0[i1], bq[1][1]);
#endif
                __builtin_amdgcn_sched_barrier(0);
                a0[i0] = a_frag(as, ah, i0, 0);
                a0[i1] = a_frag(as, ah, i1, 0);
#if PC_ABL != 2
                if (h == 0) load_sc(r, pb, sl + 3 < nslab);
#endif
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#if PC_ABL != 1
                PC_MFMA(acc1[i0][j], a1[i0], bq[1][0]);
                PC_MFMA(acc1[i1][j], a1[i1], bq[1][0]);
#endif
                __builtin_amdgcn_sched_barrier(0);
                a1[i0] = a_frag(as, ah, i0, 1);
                a1[i1] = a_frag(as, ah, i1, 1);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        advance(pa);
        advance(pb);
    };

    // prologue: stages 0, 1 and B of slabs 0, 1, 2
    PcRegs r0, r1;
    {
        const Pos p0 = pos_of(q0), p1 = pos_of(q0 + 1);
        issue_dma(p0, 0);
        issue_dma(p1, 1);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) { load_b(r0, p0, qq); load_b(r1, p1, qq); }
        load_sc(r0, p0, true);
        load_sc(r1, p1, 1 < nslab);
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) commit_pair(r0, b_lds, pp);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) load_b(r0, pa, qq);
        load_sc(r0, pa, 2 < nslab);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    read_frags(0, 0);
    // iteration s: slab s out of B buffer s & 1 and A stage s % 3; slab s + 1's B into the other buffer from the register set
    // that then takes slab s + 3.  A rolled loop of two bodies; an odd count runs one slab of zeros.
    for (int s = 0; s < nslab; s += 2) {
        slab_body(r1, 0, s);
        slab_body(r0, 1, s + 1);
    }
#undef PC_MFMA
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int pc = wn * 64 + j * 16 + fi;
            const int col = n0 + 4 * (pc & (PN / 4 - 1)) + pc / (PN / 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = mi * PM + wm * 64 + i * 16 + 4 * kq + r;
                a.slabs[((size_t)z * G4 + row) * a.Ntot + col] = (acc0[i][j][r] + H3_LO * acc1[i][j][r]) * iscmax;
            }
        }
}

// slabs [S][4H][Ntot] -> C0 [4H][n_split] (+ C1 [4H][Ntot - n_split]); fixed association
__global__ __launch_bounds__(256) void pc_reduce_kernel(const float* __restrict__ slabs, int S, int Ntot, int n_split, float* __restrict__ C0,
                                                        int64_t ldc0, float* __restrict__ C1, int64_t ldc1) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4, MN = (int64_t)G4 * Ntot;
    if (i >= MN) return;
    float4 s = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < S; ++z) {
        const float4 v = *reinterpret_cast<const float4*>(slabs + (int64_t)z * MN + i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int64_t row = i / Ntot;
    const int col = (int)(i % Ntot);
    float* c = col < n_split ? C0 + row * ldc0 + col : C1 + row * ldc1 + (col - n_split);
    *reinterpret_cast<float4*>(c) = s;
}

}  // namespace

// h_prev rows [N][T][H] = y one step back under the restart mask (h0 at t = 0): what the f32-rows fallback multiplies, since the
// forward pass of this mode writes no h_prev slot into the stash
__global__ __launch_bounds__(256) void hprev_rows_kernel(const float* __restrict__ y, const float* __restrict__ keep, const float* __restrict__ h0,
                                                         int N, int T, float* __restrict__ out) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= (int64_t)N * T * H) return;
    const int64_t row = i / H;
    const int c = (int)(i % H), t = (int)(row % T), n = (int)(row / T);
    const float k = keep ? keep[row] : 1.f;
    const float4 v = *reinterpret_cast<const float4*>(t ? y + (row - 1) * H + c : h0 + (int64_t)n * H + c);
    *reinterpret_cast<float4*>(out + i) = float4{v.x * k, v.y * k, v.z * k, v.w * k};
}
int lstm_pc_hprev_rows(const float* y, const float* keep, const float* h0, int N, int T, float* out, hipStream_t st) {
    const int64_t n4 = (int64_t)N * T * H / 4;
    hipLaunchKernelGGL(hprev_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, y, keep, h0, N, T, out);
    UAV_LAUNCH_CHECK();
    return 0;
}

// dgates (packed) -> f32 rows [N][T][4H]
int lstm_pc_unpack(const void* dgates, int N, int T, float* out, hipStream_t st) {
    const DgPack P(N, T);
    const int64_t chunks = (int64_t)T * P.RT * NS;
    hipLaunchKernelGGL(dg_unpack_kernel, dim3((unsigned)((chunks + 3) / 4)), dim3(256), 0, st, P.pieces(dgates, 0), P.isc(dgates, 0), N, T,
                       P.NP, P.RT, out);
    UAV_LAUNCH_CHECK();
    return 0;
}

// db (+ db_hh), dw_hh = dG^T h_prev (h_prev[n][t] = y[n][t-1] keep[n][t], h0[n] keep[n][0] at t = 0: rows of y / h0 under the masked
// scales iscm -- the stash's h_prev slot is not read), dw_ih = dG^T x for I <= 8 (on the column-sum pass) or I == 256 (second half of the
// product's columns) from the packed gate gradients.  Workspace: [slabs ... | partial sums, max isc word] (the tail).
int lstm_pc_wgrad(uav_ctx* ctx, const float* x, const float* y, const float* h0, const void* dgates, int N, int T, int I, float* dw_ih,
                  float* dw_hh, float* db, float* db_hh, hipStream_t st) {
    UAV_REQUIRE(I <= 8 || I == H, "lstm_pc_wgrad: input width %d (1..8 or 256)", I);
    const DgPack P(N, T);
    const bool wide = I == H;
    const int nb = 256, CI = wide ? 0 : I;
    const size_t part_floats = (size_t)nb * (1 + CI) * G4 + 64;
    UAV_REQUIRE(ctx->ws_bytes >= part_floats * 4 + (64u << 20), "uav_lstm_wgrad: workspace too small");
    float* partial = (float*)((char*)ctx->ws + ctx->ws_bytes) - part_floats;
    unsigned* iscmax = reinterpret_cast<unsigned*>(partial + (size_t)nb * (1 + CI) * G4);
    UAV_CHECK_HIP(hipMemsetAsync(iscmax, 0, sizeof(unsigned), st));
    const int64_t tiles = (int64_t)T * P.RT;
    const int tpb = (int)((tiles + nb - 1) / nb);
    const int nbu = (int)((tiles + tpb - 1) / tpb);
#define PC_COLSUM(I_)                                                                                                             \
    hipLaunchKernelGGL(colsum_pc_kernel<I_>, dim3(nbu, 2), dim3(512), 0, st, P.pieces(dgates, 0), P.isc(dgates, 0), N, T, P.NP, P.RT, \
                       tiles, tpb, x, partial, iscmax)
    switch (CI) {
        case 0: PC_COLSUM(0); break;
        case 1: PC_COLSUM(1); break;
        case 2: PC_COLSUM(2); break;
        case 3: PC_COLSUM(3); break;
        case 4: PC_COLSUM(4); break;
        case 5: PC_COLSUM(5); break;
        case 6: PC_COLSUM(6); break;
        case 7: PC_COLSUM(7); break;
        default: PC_COLSUM(8); break;
    }
#undef PC_COLSUM
    hipLaunchKernelGGL(colsum_pc_reduce_kernel, dim3((1 + CI) * G4 / 32), dim3(256), 0, st, partial, nbu, CI, db, db_hh, dw_ih);

    PcArgs a;
    a.pieces = P.pieces(dgates, 0);
    a.isc = P.isc(dgates, 0);
    a.iscmax_bits = iscmax;
    a.N = N; a.T = T; a.NP = P.NP; a.RT = P.RT; a.SPT = P.NP / 32;
    a.iscm = P.iscm(dgates, 0);
    a.y = y; a.h0 = h0;
    a.B1 = wide ? x : nullptr; a.ldb1 = I;
    a.n_split = H; a.Ntot = wide ? 2 * H : H;
    a.total = (int64_t)a.SPT * T;
    a.tm = G4 / PM; a.tn = a.Ntot / PN;
    UAV_REQUIRE((int64_t)4 * T * H * 4 < (1ll << 31), "uav_lstm_wgrad (h=256): T = %d too long for 32-bit lane offsets", T);
    // split-K: whole groups of 8 splits (one per XCD), each split at least 16 slabs, the slabs inside the workspace
    const int tiles_mn = a.tm * a.tn;
    int64_t S = (ctx->num_cu + tiles_mn - 1) / tiles_mn;
    S = (S + 7) / 8 * 8;
    const int64_t max_k = a.total / 16 > 0 ? a.total / 16 : 1;
    const int64_t max_ws = (int64_t)((ctx->ws_bytes - part_floats * 4) / sizeof(float)) / ((int64_t)G4 * a.Ntot);
    if (S > max_k) S = max_k;
    if (S > max_ws) S = max_ws;
    UAV_REQUIRE(S >= 1, "uav_lstm_wgrad: workspace too small for one slab of the weight-gradient product");
    a.sps = (a.total + S - 1) / S;
    S = (a.total + a.sps - 1) / a.sps;
    a.S = (int)S;
    a.slabs = (float*)ctx->ws;
    const int64_t grid = 8 * (int64_t)tiles_mn * ((S + 7) / 8);
    UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(&gemm_pc_kernel), (int)PC_LDS));
    hipLaunchKernelGGL(gemm_pc_kernel, dim3((unsigned)grid), dim3(512), PC_LDS, st, a);
    const int64_t MN = (int64_t)G4 * a.Ntot;
    hipLaunchKernelGGL(pc_reduce_kernel, dim3((unsigned)((MN / 4 + 255) / 256)), dim3(256), 0, st, a.slabs, a.S, a.Ntot, a.n_split, dw_hh,
                       (int64_t)H, dw_ih, (int64_t)I);
    UAV_LAUNCH_CHECK();
    return 0;
}

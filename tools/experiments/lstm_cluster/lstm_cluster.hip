// lstm_cluster.hip -- h = 256 LSTM layers as PERSISTENT CLUSTER kernels with RESIDENT weights (BASELINE C5's stacked policy;
// semantics: torch.nn.LSTM as used at PPOV2.0/model.py:206-212, gate order i, f, g, o).
//
// Why: the two fp16 pieces of a 1024 x 256 weight matrix are 1 MB -- no CU can hold them, so lstm_generic.hip runs ONE launch
// per time step and layer and every launch streams its weight pieces out of L2 again (C5: ~8,000 dependent launches of
// ~20 us per PPO iteration, 0.11-0.19 of the fp16 pipe by the SQ counters).  Here a layer's whole time loop is one launch:
//
//   cluster = 8 workgroups (one per CU, co-resident: the grid never exceeds the CU count and every workgroup needs more
//             than half a CU's LDS), dealt to ONE XCD (blocks b and b + 8 share an XCD under round-robin placement --
//             a speed matter only, correctness never depends on placement);
//   each workgroup owns 32 hidden units = 128 gate rows and keeps BOTH fp16 pieces of its W_hh slice (and, for a stacked
//             layer, of its W_ih slice) in REGISTERS for the whole launch: 4 waves x (128 + 128) VGPRs, one wave per SIMD;
//   a cluster advances one 64-env tile through all T steps; per step every workgroup computes the gates of its 32 units
//             for the 64 envs (weights as the MFMA A operand, h_{t-1} / x_t piece planes in LDS as B), runs the cell for
//             them (c never leaves the registers), and hands its 32 units of h_t to the other seven through L2:
//             write-through (sc1) 16-byte stores -> every wave's vmcnt(0) -> workgroup barrier -> one sc1 flag store;
//             consumers poll the seven flags with sc1 loads, then read the 4 KB blocks with sc1 buffer loads
//             (MI355X_MICROARCH.md, "Valid forms", first row of the sc1 table).  The exchange buffer is double-buffered
//             by step parity: a producer overwrites a slot only after it has seen every peer's flag of the step in
//             between, which a peer publishes after it has consumed that slot.  Every step carries the flag handshake
//             (also the last of a tile, which has no payload), so the invariant holds across tiles.
//   every spin is bounded: a peer that never shows up sets an error counter and the cluster runs on without waiting
//             (garbage results, reported by uav_lstm_cluster_errors) -- the grid always drains.
//
// Arithmetic and accumulation order are EXACTLY those of step_fwd_h3_kernel (bias, the eight recurrent slabs, then the input
// slabs; per slab cross += a1 b0, main += a0 b0, cross += a0 b1; the same activations and cell expressions, contraction
// off), so results are BIT-identical to the per-step path: tests/test_gpu_lstm_cluster.py.
#include <type_traits>
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace c8 {
constexpr int H = 256, G = 8, UC = 32, E = 64, NCT = E / 16, NS = H / 32;
constexpr int RS = H + 8;                 // halves per env row of a piece plane (+8: conflict-free 16-byte fragment reads)
constexpr int PLANE = E * RS;             // halves per piece plane
constexpr int XBLK = E * UC;              // halves per (workgroup, piece) block of the exchange buffer (4 KB)
constexpr int FLAG_STRIDE = 32;           // one flag per 128-byte line
constexpr unsigned SPIN_LIMIT = 1u << 21;
constexpr int MAX_CLUSTERS = 32;
}  // namespace c8

struct C8FwdArgs {
    const float* x;        // [N][T][I]
    const float* w_ih;     // [4H][I]
    const float* w_hh;     // [4H][H]
    const float* b_ih;
    const float* b_hh;
    const float* keep;     // [N][T] or null
    const float* h0;       // [N][H]
    const float* c0;
    float* y;              // [N][T][H]
    float* stash;          // [N][T][6H]
    float* hn;
    float* cn;
    unsigned short* xh;    // exchange [2][ncl][G][2 pieces][4 chunks of 8 units][E][8] halves
    unsigned* flags;       // [ncl][G][FLAG_STRIDE]
    unsigned* err;
    int I, N, T, ntile, ncl;
    unsigned xh_bytes;
    unsigned abl;          // measurement-only switches (UAV_DEBUG_CLUSTER_ABL bits): 1 no peer wait, 2 no stash / y stores, 4 no peer fetch, 8 no products,
                           // 0x10 hand-off payload through write-through (sc1) stores even when the cluster sits on one XCD (results unchanged)
};

__device__ __forceinline__ f16x8 c8_ldh8(const unsigned short* p) { return *reinterpret_cast<const f16x8*>(p); }

// 8 consecutive f32 -> the two fp16 pieces as MFMA fragments
__device__ __forceinline__ void c8_split8(const float (&v)[8], f16x8& p0, f16x8& p1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        _Float16 a, b;
        split2h(v[i], a, b);
        p0[i] = a;
        p1[i] = b;
    }
}

// f32 x4 -> the two fp16 pieces, packed 4 halves per piece
__device__ __forceinline__ void c8_split4(const float4 v, uint2& q0, uint2& q1) {
    _Float16 p0[4], p1[4];
    split2h(v.x, p0[0], p1[0]); split2h(v.y, p0[1], p1[1]); split2h(v.z, p0[2], p1[2]); split2h(v.w, p0[3], p1[3]);
    q0.x = (unsigned)h_bits(p0[0]) | ((unsigned)h_bits(p0[1]) << 16); q0.y = (unsigned)h_bits(p0[2]) | ((unsigned)h_bits(p0[3]) << 16);
    q1.x = (unsigned)h_bits(p1[0]) | ((unsigned)h_bits(p1[1]) << 16); q1.y = (unsigned)h_bits(p1[2]) | ((unsigned)h_bits(p1[3]) << 16);
}

// NSX = input slabs of 32: 1 (a first layer: I <= 32 observation channels) or 8 (a stacked layer: I = 256)
// -DUAV_C8_PROFILE (tools/build_prof.sh): cycles per phase of the time loop, accumulated by wave 0 of workgroup 0 and left in
// the words behind the error counter (read with uav_c8_profile; tools/perf_cluster_fwd.py prof)
#ifdef UAV_C8_PROFILE
#define C8_STAMP(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); prof_[k] += now_ - last_; last_ = now_; } while (0)
#else
#define C8_STAMP(k) do { } while (0)
#endif

template <int NSX>
__global__ __launch_bounds__(256, 1) void lstm_fwd_c8_kernel(const C8FwdArgs a) {
    using namespace c8;
#ifdef UAV_C8_PROFILE
    unsigned long long prof_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_ = 0;
#endif
    // DB: a first layer's planes fit twice, so h_t / x_{t+1} are written into the OTHER buffer while h_{t-1} / x_t are still
    // being read -- no barrier between the products and the cell, and the cell of one column tile runs under the next
    // column tile's products.  A stacked layer (NSX = 8) has LDS for one set only: its own slice of h_t waits in registers
    // for the barrier behind the last product.
    constexpr bool DB = NSX == 1;
    constexpr int RSX = 32 * NSX + 8, XPLANE = E * RSX;
    constexpr int HSET = 2 * PLANE, XSET = 2 * XPLANE;
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    unsigned short* hpl = lds;                                   // h pieces [DB ? 2 : 1][2][E][RS]
    unsigned short* xpl = lds + (DB ? 2 : 1) * HSET;             // x pieces [DB ? 2 : 1][2][E][RSX]
    unsigned* kb = reinterpret_cast<unsigned*>(xpl + (DB ? 2 : 1) * XSET);      // keep != 0 as one bit per env: [T][2] words
    __shared__ int s_dead;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, rg = lane >> 4;       // B fragment: env j, k quarter rg; accumulator: env j, rows 4 rg .. 4 rg + 3
    const int b = blockIdx.x;
    const int cl = (b >> 6) * 8 + (b & 7), cu = (b >> 3) & 7;      // blocks b, b + 8, .. b + 56 form a cluster (one XCD)
    if (cl >= a.ncl) return;
    const int T = a.T, N = a.N, I = a.I;
    const int u_w = UC * cu + 8 * w;               // first of this wave's 8 units
    const bool up = rg >= 2;                       // upper half-wave: f, o, c, h of the wave's units; lower: i, g
    if (tid == 0) s_dead = 0;

    // ---- resident weights: row tile 0 = [i(8 units) | f(8 units)], row tile 1 = [g | o]; lane (j, rg) feeds row j, k = 32 s + 8 rg ..
    f16x8 wa[2][NS][2];
    f16x8 wx[2][NSX][2];
    {
        const int unit = u_w + (j & 7);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int row = (2 * rt + (j >> 3)) * H + unit;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float* src = a.w_hh + (size_t)row * H + 32 * s + 8 * rg;
                const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
                const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                c8_split8(v, wa[rt][s][0], wa[rt][s][1]);
                asm volatile("" : "+a"(wa[rt][s][0]));      // pinned at once (see below): the 256 fragments never meet in VGPRs
                asm volatile("" : "+a"(wa[rt][s][1]));
            }
#pragma unroll
            for (int s = 0; s < NSX; ++s) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k = 32 * s + 8 * rg + i;
                    const float t = a.w_ih[(size_t)row * I + min(k, I - 1)];
                    v[i] = k < I ? t : 0.f;
                }
                c8_split8(v, wx[rt][s][0], wx[rt][s][1]);
                asm volatile("" : "+a"(wx[rt][s][0]));
                asm volatile("" : "+a"(wx[rt][s][1]));
            }
        }
    }
    // pin the weight fragments in the accumulation-register half of the file: AGPRs feed the MFMA A operand directly, the
    // accumulators take the VGPR form (Makefile: -mllvm -amdgpu-mfma-vgpr-form for this file).  Left to itself the register
    // allocator treats AGPRs as spill slots: every fragment then costs four v_accvgpr_read per use and, with 256 weight
    // registers + 64 AGPR accumulators, the stacked-layer kernel spilled 66 registers to scratch inside the time loop
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
        for (int s = 0; s < NS; ++s) { asm volatile("" : "+a"(wa[rt][s][0])); asm volatile("" : "+a"(wa[rt][s][1])); }
#pragma unroll
        for (int s = 0; s < NSX; ++s) { asm volatile("" : "+a"(wx[rt][s][0])); asm volatile("" : "+a"(wx[rt][s][1])); }
    }
    // accumulator (row tile rt, lane (j, rg), register r) <-> gate 2 rt + (rg >> 1), unit u_acc + r
    const int u_acc = u_w + 4 * (rg & 1);
    f32x4 bias[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int o = (2 * rt + (rg >> 1)) * H + u_acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[rt][r] = a.b_ih[o + r] + a.b_hh[o + r];
    }
    const float m_act = up ? -1.0f : 2.0f;         // tile 1 holds g (tanh) in the lower half-wave, o (sigmoid) in the upper

    const __amdgpu_buffer_rsrc_t xh_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.xh, 0, a.xh_bytes, 0x00020000);
    unsigned* my_flag = a.flags + (size_t)(cl * G + cu) * FLAG_STRIDE;
    const unsigned* peer_flag = a.flags + (size_t)(cl * G + (lane & 7)) * FLAG_STRIDE;
    bool dead = (a.abl & 1u) != 0;
    unsigned gstep = 0;                            // steps this cluster has published so far (all tiles)

    // ---- are the eight workgroups of this cluster on ONE XCD?  Placement is not ours to choose (blocks b and b + 8 are dealt to
    // the same XCD in practice, nothing promises it), so it is MEASURED: every workgroup publishes its HW_REG_XCC_ID, the
    // cluster reads all eight.  On one XCD the hand-off payload can stay in that XCD's L2: plain producer stores (performed at
    // L2 once vmcnt(0) returns: the vector L1 is write-through), consumer loads that bypass their L1 (sc1) and hit the shared
    // L2 -- 104-122 GB/s per workgroup instead of the 12-20 GB/s of a payload that write-through (sc1) stores pushed out to the
    // memory side (MI355X_MICROARCH.md, handoff-payload).  Otherwise: the sc1 / sc1 form, correct under any placement.
    bool same_xcd = false;
    {
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 0xfu;      // HW_REG_XCC_ID, bits 3:0
        if (tid == 0) __hip_atomic_store(my_flag + 1, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __shared__ int s_same;
        if (w == 0) {
            unsigned v = lane < G ? 0u : 1u, spins = 0;
            while (true) {
                if (v == 0u) v = __hip_atomic_load(peer_flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_ballot_w64(v == 0u) == 0ull) break;
                if (++spins > SPIN_LIMIT) {
                    if (lane == 0) { atomicAdd(a.err, 1u); s_dead = 1; }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            const bool mine = lane >= G || v == xcc + 1u;
            if (lane == 0) s_same = (__builtin_amdgcn_ballot_w64(!mine) == 0ull && !(a.abl & 0x10u)) ? 1 : 0;
        }
        __syncthreads();
        same_xcd = s_same != 0;
        dead = dead || s_dead != 0;
    }

    for (int tile = cl; tile < a.ntile; tile += a.ncl) {
        const int e0 = tile * E;
        // x_t (f32 rows) -> piece planes, in two halves: the loads (issued early, all of them in flight at once) and the
        // split + LDS stores.  NSX = 8: a 1 KB row per env, 16 float4 per thread; NSX = 1: 8 values per thread, zero beyond I
        constexpr int XR = NSX == 8 ? E * H / 4 / 256 : 2;
        float4 xreg[XR];
        auto load_x = [&](int t) {
            if (NSX == 8) {
#pragma unroll
                for (int it = 0; it < XR; ++it) {
                    const int q = it * 256 + tid, env = q >> 6, u4 = (q & 63) * 4;
                    xreg[it] = *reinterpret_cast<const float4*>(a.x + ((size_t)min(e0 + env, N - 1) * T + t) * H + u4);
                }
            } else {
                const int env = tid >> 2, k0 = (tid & 3) * 8;
                const float* xr = a.x + ((size_t)min(e0 + env, N - 1) * T + t) * I;
                xreg[0] = xreg[1] = float4{0.f, 0.f, 0.f, 0.f};
                if (I == 8) {                      // the common shape (6 observation channels + 2 trends): two 16-byte loads, one test
                    if (k0 == 0) { xreg[0] = *reinterpret_cast<const float4*>(xr); xreg[1] = *reinterpret_cast<const float4*>(xr + 4); }
                } else if (k0 < I) {
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float f = xr[min(k0 + i, I - 1)];
                        v[i] = k0 + i < I ? f : 0.f;
                    }
                    xreg[0] = float4{v[0], v[1], v[2], v[3]};
                    xreg[1] = float4{v[4], v[5], v[6], v[7]};
                }
            }
        };
        auto store_x = [&](unsigned short* xdst) {
            if (NSX == 8) {
#pragma unroll
                for (int it = 0; it < XR; ++it) {
                    const int q = it * 256 + tid, env = q >> 6, u4 = (q & 63) * 4;
                    uint2 q0, q1;
                    c8_split4(xreg[it], q0, q1);
                    *reinterpret_cast<uint2*>(xdst + env * RSX + u4) = q0;
                    *reinterpret_cast<uint2*>(xdst + XPLANE + env * RSX + u4) = q1;
                }
            } else {
                const int env = tid >> 2, k0 = (tid & 3) * 8;
                const float v[8] = {xreg[0].x, xreg[0].y, xreg[0].z, xreg[0].w, xreg[1].x, xreg[1].y, xreg[1].z, xreg[1].w};
                f16x8 p0, p1;
                c8_split8(v, p0, p1);
                *reinterpret_cast<f16x8*>(xdst + env * RSX + k0) = p0;
                *reinterpret_cast<f16x8*>(xdst + XPLANE + env * RSX + k0) = p1;
            }
        };
        lds_barrier();                                               // the previous tile's last reads of the planes
        // ---- restart masks of the whole tile as bits (no global load of them inside the time loop)
        for (int t = w; t < T; t += 4) {
            const bool kv = a.keep ? a.keep[(size_t)min(e0 + lane, N - 1) * T + t] != 0.f : true;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(kv);
            if (lane == 0) { kb[2 * t] = (unsigned)m; kb[2 * t + 1] = (unsigned)(m >> 32); }
        }
        // ---- h_{-1} = h0 * keep[:, 0] as piece planes (every workgroup holds the whole tile's h) -> buffer 0
#pragma unroll 4
        for (int it = 0; it < E * H / 4 / 256; ++it) {
            const int q = it * 256 + tid, env = q >> 6, u4 = (q & 63) * 4;
            const int n = min(e0 + env, N - 1);
            float4 v = *reinterpret_cast<const float4*>(a.h0 + (size_t)n * H + u4);
            const float k = a.keep ? a.keep[(size_t)n * T] : 1.f;
            v.x *= k; v.y *= k; v.z *= k; v.w *= k;
            uint2 q0, q1;
            c8_split4(v, q0, q1);
            *reinterpret_cast<uint2*>(hpl + env * RS + u4) = q0;
            *reinterpret_cast<uint2*>(hpl + PLANE + env * RS + u4) = q1;
        }
        // the cell state of this lane's units (upper half-wave), f32; row 0 of the stash gets the masked initial state
        f32x4 c_st[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const int n = e0 + 16 * c + j, nc = min(n, N - 1);
            const float4 cv = *reinterpret_cast<const float4*>(a.c0 + (size_t)nc * H + u_acc);
            c_st[c] = f32x4{cv.x, cv.y, cv.z, cv.w};
            if (up && n < N && !(a.abl & 2u)) {
                const float4 hv = *reinterpret_cast<const float4*>(a.h0 + (size_t)nc * H + u_acc);
                const float k = a.keep ? a.keep[(size_t)nc * T] : 1.f;
                float* sp = a.stash + ((size_t)n * T) * (6 * H) + u_acc;
                *reinterpret_cast<float4*>(sp + 4 * H) = float4{cv.x * k, cv.y * k, cv.z * k, cv.w * k};
                *reinterpret_cast<float4*>(sp + 5 * H) = float4{hv.x * k, hv.y * k, hv.z * k, hv.w * k};
            }
        }
        load_x(0);
        store_x(xpl);
        lds_barrier();

#ifdef UAV_C8_PROFILE
        last_ = __builtin_readcyclecounter();
#endif
        for (int t = 0; t < T; ++t, ++gstep) {
            const unsigned short* hcur = hpl + (DB ? (t & 1) * HSET : 0);
            unsigned short* hnxt = hpl + (DB ? ((t + 1) & 1) * HSET : 0);
            const unsigned short* xcur = xpl + (DB ? (t & 1) * XSET : 0);
            unsigned short* xnxt = xpl + (DB ? ((t + 1) & 1) * XSET : 0);
            // restart masks from the bit table: kc of this step (applied to c_prev), kn of the next (applied to the state handed on)
            const uint2 mc = *reinterpret_cast<const uint2*>(kb + 2 * t);
            uint2 mn = uint2{0xffffffffu, 0xffffffffu};
            if (t + 1 < T) mn = *reinterpret_cast<const uint2*>(kb + 2 * (t + 1));
            float kc[NCT], kn[NCT];
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                kc[c] = (((c < 2 ? mc.x : mc.y) >> ((16 * c + j) & 31)) & 1u) ? 1.f : 0.f;
                kn[c] = (((c < 2 ? mn.x : mn.y) >> ((16 * c + j) & 31)) & 1u) ? 1.f : 0.f;
            }
            const bool kf = ((((lane >> 5) ? mn.y : mn.x) >> (lane & 31)) & 1u) != 0u;      // env (tid & 63) of a peer block
            if (DB && t + 1 < T) load_x(t + 1);    // next step's input rows: in flight under the products
            C8_STAMP(0);

            // ---- per column tile: gates = bias + W_hh h_{t-1} + W_ih x_t (three piece products per slab; recurrent slabs first,
            // as the step kernels), then the cell of that tile -- written so that tile c + 1's products sit next to tile c's cell
            const unsigned slot = (gstep & 1u) * (unsigned)(a.ncl * G * 2 * XBLK) + (unsigned)((cl * G + cu) * 2 * XBLK);
            f32x4 act0[NCT], act1[NCT], hout[NCT], cnext[NCT];
            u32x4 hown[NCT];
            f32x4 acc[2][2], acl[2][2];          // [column-tile parity][row tile]
            auto products = [&](int c) {
                f32x4(&ac)[2] = acc[c & 1];
                f32x4(&al)[2] = acl[c & 1];
                ac[0] = bias[0]; ac[1] = bias[1];
                al[0] = f32x4{0.f, 0.f, 0.f, 0.f}; al[1] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (a.abl & 8u) return;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const unsigned short* hr = hcur + (16 * c + j) * RS + 32 * s + 8 * rg;
                    const f16x8 b0 = c8_ldh8(hr), b1 = c8_ldh8(hr + PLANE);
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        al[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[rt][s][1], b0, al[rt], 0, 0, 0);
                        ac[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[rt][s][0], b0, ac[rt], 0, 0, 0);
                        al[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[rt][s][0], b1, al[rt], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int s = 0; s < NSX; ++s) {
                    const unsigned short* xr = xcur + (16 * c + j) * RSX + 32 * s + 8 * rg;
                    const f16x8 b0 = c8_ldh8(xr), b1 = c8_ldh8(xr + XPLANE);
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        al[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wx[rt][s][1], b0, al[rt], 0, 0, 0);
                        ac[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wx[rt][s][0], b0, ac[rt], 0, 0, 0);
                        al[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wx[rt][s][0], b1, al[rt], 0, 0, 0);
                    }
                }
            };
            // After the products the gates of a unit sit in two lanes (j, rg) and (j, rg + 2): i, g below, f, o above
            auto cell = [&](int c) {
                const f32x4(&ac)[2] = acc[c & 1];
                const f32x4(&al)[2] = acl[c & 1];
                f32x4 hh;
                float ig[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pre0 = ac[0][r] + al[0][r] * H3_LO, pre1 = ac[1][r] + al[1][r] * H3_LO;
                    act0[c][r] = fast_sigmoid(pre0);                                  // i (lower) | f (upper)
                    const float rr = __builtin_amdgcn_rcpf(1.0f + __expf(m_act * pre1));
                    act1[c][r] = up ? rr : 1.0f - 2.0f * rr;                          // tanh g (lower) | sigmoid o (upper)
                    ig[r] = act0[c][r] * act1[c][r];                                  // i g (lower half-wave)
                }
                _Float16 q0[4], q1[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, ig[r]), __builtin_bit_cast(unsigned, ig[r]), false, false);
                    const float igu = __builtin_bit_cast(float, (unsigned)sw[0]);     // upper lanes: i g of lane - 32
                    const float cp = c_st[c][r] * kc[c];
                    const float cc = act0[c][r] * cp + igu;
                    hh[r] = act1[c][r] * fast_tanh(cc);
                    c_st[c][r] = cc;
                    cnext[c][r] = cc * kn[c];                 // the masked state entering step t + 1 (stash row t + 1)
                    split2h(hh[r], q0[r], q1[r]);
                }
                hout[c] = hh;
                if (t + 1 < T) {
                    const unsigned p0lo = (unsigned)h_bits(q0[0]) | ((unsigned)h_bits(q0[1]) << 16), p0hi = (unsigned)h_bits(q0[2]) | ((unsigned)h_bits(q0[3]) << 16);
                    const unsigned p1lo = (unsigned)h_bits(q1[0]) | ((unsigned)h_bits(q1[1]) << 16), p1hi = (unsigned)h_bits(q1[2]) | ((unsigned)h_bits(q1[3]) << 16);
                    // rows 2, 3 of the wave (rg = 2, 3) hold units u_w .. + 3 and + 4 .. + 7: after the row swap lane (j, 2) has piece 0
                    // of all 8 units, lane (j, 3) piece 1 -- one 16-byte store each
                    const auto slo = __builtin_amdgcn_permlane16_swap(p0lo, p1lo, false, false);
                    const auto shi = __builtin_amdgcn_permlane16_swap(p0hi, p1hi, false, false);
                    u32x4 v = {(unsigned)slo[0], (unsigned)shi[0], (unsigned)slo[1], (unsigned)shi[1]};
                    if (up) {
                        const int piece = rg - 2, env = 16 * c + j;
                        const unsigned off = (slot + (unsigned)(piece * XBLK + w * (E * 8) + env * 8)) * 2u;      // [piece][8-unit chunk = wave][env][8]
                        if (same_xcd) __builtin_amdgcn_raw_buffer_store_b128(v, xh_rsrc, off, 0, 0);      // stays in the XCD's L2
                        else __builtin_amdgcn_raw_buffer_store_b128(v, xh_rsrc, off, 0, 16);              // sc1: write-through
                        if (kn[c] == 0.f) v = u32x4{0u, 0u, 0u, 0u};
                        if (DB) *reinterpret_cast<u32x4*>(hnxt + piece * PLANE + env * RS + UC * cu + 8 * w) = v;
                        else hown[c] = v;
                    }
                }
            };
            products(0);
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                if (c + 1 < NCT) products(c + 1);
                cell(c);
            }
            C8_STAMP(1);
            if (!DB) {
                lds_barrier();               // #1 every wave has read the planes of this step
                if (up && t + 1 < T) {
#pragma unroll
                    for (int c = 0; c < NCT; ++c)
                        *reinterpret_cast<u32x4*>(hnxt + (rg - 2) * PLANE + (16 * c + j) * RS + UC * cu + 8 * w) = hown[c];
                }
            }
            if (!DB && t + 1 < T) load_x(t + 1);    // (a stacked layer has no registers for 16 rows in flight under the products)
            // ---- publish: every wave's stores performed, then ONE flag store
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            C8_STAMP(4);
            __builtin_amdgcn_s_barrier();    // #2
            C8_STAMP(5);
            if (tid == 0) __hip_atomic_store(my_flag, gstep + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

            // ---- next step's input -> x planes; does not depend on the peers
            if (t + 1 < T) store_x(xnxt);
            C8_STAMP(6);

            // ---- wait for the seven peers (bounded), then pull their 32 units of h_t
            if (w == 0 && !dead) {
                bool ok = lane >= G || lane == cu;
                unsigned spins = 0;
                while (true) {
                    if (!ok) ok = __hip_atomic_load(peer_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gstep + 1u;
                    if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                    if (++spins > SPIN_LIMIT) {
                        if (lane == 0) { atomicAdd(a.err, 1u); s_dead = 1; }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            C8_STAMP(7);
            __builtin_amdgcn_s_barrier();    // #3
            C8_STAMP(8);
            dead = s_dead != 0 || (a.abl & 1u);
            if (t + 1 < T && !(a.abl & 4u)) {
                const unsigned base = (gstep & 1u) * (unsigned)(a.ncl * G * 2 * XBLK) + (unsigned)(cl * G * 2 * XBLK);
                // ALL eight blocks (this workgroup's own included: the same bytes it parked in LDS itself), no branch between the
                // loads: with a `pc != cu` test per block the compiler put an s_waitcnt vmcnt(0) in front of every load -- fourteen
                // L2 round trips in series, 5.4 k of a 23 k-cycle step (profiles/r04_k_*)
                constexpr int NB = DB ? 1 : 2;       // a first layer: all sixteen 16-byte loads in flight at once; stacked: two batches
                auto fetch = [&](auto policy) {
                    constexpr int AUX = decltype(policy)::value;
#pragma unroll
                    for (int half = 0; half < NB; ++half) {
                        u32x4 pv[G / NB][2];
#pragma unroll
                        for (int q = 0; q < G / NB; ++q)
#pragma unroll
                            for (int piece = 0; piece < 2; ++piece)
                                pv[q][piece] = __builtin_amdgcn_raw_buffer_load_b128(
                                    xh_rsrc, (base + (unsigned)(((half * (G / NB) + q) * 2 + piece) * XBLK + tid * 8)) * 2u, 0, AUX);
#ifdef UAV_C8_PROFILE
                        C8_STAMP(2);                              // (profile build: issue | latency | LDS stores of the peer fetch)
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        C8_STAMP(3);
#endif
#pragma unroll
                        for (int q = 0; q < G / NB; ++q)
#pragma unroll
                            for (int piece = 0; piece < 2; ++piece) {
                                u32x4 v = pv[q][piece];
                                if (!kf) v = u32x4{0u, 0u, 0u, 0u};
                                // block layout [chunk][env][8 units]: thread = (chunk tid >> 6, env tid & 63) -> a wave's 64 lanes write 16
                                // bytes each into 64 consecutive rows (stride 528 B = 16 B past a bank period): conflict-free
                                *reinterpret_cast<u32x4*>(hnxt + piece * PLANE + (tid & 63) * RS + UC * (half * (G / NB) + q) + (tid >> 6) * 8) = v;
                            }
                    }
                };
                // one XCD: nt = past the L1, served by the shared L2 the producers' plain stores sit in; otherwise sc1 (device scope)
                if (same_xcd) fetch(std::integral_constant<int, 2>{});
                else fetch(std::integral_constant<int, 16>{});
            }
            C8_STAMP(9);

            // ---- BPTT stash: gates of row t, the masked state entering step t + 1 in row t + 1's c_prev | h_prev slots; layer
            // output.  Issued last, so these stores drain under the next step's products and no load waits behind them
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const int n = e0 + 16 * c + j;
                if (n >= N || (a.abl & 2u)) continue;
                const size_t row = (size_t)n * T + t;
                float* sp = a.stash + row * (6 * H) + u_acc;
                const int g0 = rg >> 1;                                   // 0: this lane holds i and g; 1: f and o
                *reinterpret_cast<float4*>(sp + g0 * H) = float4{act0[c][0], act0[c][1], act0[c][2], act0[c][3]};
                *reinterpret_cast<float4*>(sp + (2 + g0) * H) = float4{act1[c][0], act1[c][1], act1[c][2], act1[c][3]};
                if (up) {
                    *reinterpret_cast<float4*>(a.y + row * H + u_acc) = float4{hout[c][0], hout[c][1], hout[c][2], hout[c][3]};
                    if (t + 1 < T) {
                        *reinterpret_cast<float4*>(sp + 6 * H + 4 * H) = float4{cnext[c][0], cnext[c][1], cnext[c][2], cnext[c][3]};
                        *reinterpret_cast<float4*>(sp + 6 * H + 5 * H) = float4{hout[c][0] * kn[c], hout[c][1] * kn[c], hout[c][2] * kn[c], hout[c][3] * kn[c]};
                    } else {
                        *reinterpret_cast<float4*>(a.hn + (size_t)n * H + u_acc) = float4{hout[c][0], hout[c][1], hout[c][2], hout[c][3]};
                        *reinterpret_cast<float4*>(a.cn + (size_t)n * H + u_acc) = float4{c_st[c][0], c_st[c][1], c_st[c][2], c_st[c][3]};
                    }
                }
            }
            C8_STAMP(10);
            lds_barrier();                   // #4 the planes of step t + 1 are complete
            C8_STAMP(11);
        }
    }
#ifdef UAV_C8_PROFILE
    if (blockIdx.x == 0 && tid == 0)
        for (int k = 0; k < 12; ++k) reinterpret_cast<unsigned long long*>(a.err + 2)[k] = prof_[k];
#endif
}

#ifdef UAV_C8_PROFILE
extern "C" int uav_c8_profile(uav_ctx* ctx, unsigned long long* out12) {
    return hipMemcpy(out12, ctx->cluster_err + 2, 12 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif

// ================================================================================================ backward (BPTT)
// One launch = the whole BPTT of one h = 256 layer, same clusters as the forward: a workgroup owns the gate rows of its 32 units.
//   per step t (descending), per 64-env tile:
//     cell backward for (env, 8 units) per thread -- stash rows, dy (or dheads . W_head for the top layer) and the dgates rows are
//       whole 128-byte lines per (env, gate), four lanes each;
//     dG (this workgroup's 128 gate rows of the tile) block-scaled per env by a power of two (largest magnitude in [2^13, 2^14),
//       exact), split into two fp16 pieces, parked in LDS as the MFMA B operand;
//     partial dh_{t-1}[all 256 units] = W_hh[own 128 rows]^T dG: wave w forms output units 64 w .. + 63 over the whole K = 128,
//       weights resident (W_hh^T slices: 128 AGPRs per wave; a stacked layer also W_ih^T for dx_t: another 128);
//     the partial tiles go to the workgroups that OWN those units through the same L2 hand-off as the forward's h_t (f32, 8 KB
//       per (owner, source) pair and step); an owner sums its eight source blocks in a fixed order (deterministic), masks by
//       keep[:, t] (folded into the scale) and has dh_{t-1} of its units; dx_t likewise, written to HBM by the owner.
// The per-env scale is per WORKGROUP here (the per-step kernels scale an env's whole 1024-row dG by one power of two), so the
// products round differently: equal to the per-step path to f32 accuracy (tests/test_gpu_lstm_cluster.py), not bit for bit.
struct C8BwdArgs {
    const float* keep;     // [N][T] or null
    const float* stash;    // [N][T][6H]
    const float* w_hh;     // [4H][H]
    const float* w_ih;     // [4H][H] (dx only)
    const float* dy;       // [N][T][H], or null: the top layer forms dy = dheads . w_head itself
    const float* dheads;   // [N*T][NH]
    const float* w_head;   // [NH][H]
    const float* dhn;      // [N][H] or null
    const float* dcn;
    float* dgates;         // [N][T][4H]
    float* dx;             // [N][T][H] or null
    float* dh0;
    float* dc0;
    float* xp;             // partial exchange [2][ncl][G owners][G sources][E][UC] f32 (dh), then the same for dx
    unsigned* flags;
    unsigned* err;
    int NH, N, T, ntile, ncl;
    unsigned xp_bytes, xq_off;     // bytes of the whole exchange buffer; byte offset of the dx half
    unsigned abl;
};

namespace c8 {
constexpr int RSG = 128 + 8;              // halves per env row of a dG piece plane
constexpr int GPLANE = E * RSG;
constexpr int PBLK = E * UC;              // floats per (owner, source) block of the partial exchange (8 KB)
}  // namespace c8

// TOP: the layer under the heads (dy = dheads . w_head formed here); compile-time, so that the rows loaded a step ahead reach the
// next iteration in the registers they were loaded into (a run-time `if (dy)` merges two definitions: copies, hence a wait on the spot)
template <bool DX, bool TOP>
__global__ __launch_bounds__(256, 1) void lstm_bwd_c8_kernel(const C8BwdArgs a) {
    using namespace c8;
#ifdef UAV_C8_PROFILE
    unsigned long long prof_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_ = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    unsigned short* gpl = lds;                                              // dG pieces [2][E][RSG]
    float* isc_h = reinterpret_cast<float*>(lds + 2 * GPLANE);              // [E] inverse scale x keep[:, t]  (dh_{t-1})
    float* isc_x = isc_h + E;                                               // [E] inverse scale               (dx_t)
    float* whd = isc_x + E;                                                 // [8][UC] head weights of the own units (top layer)
    f16x8* wxl = reinterpret_cast<f16x8*>(whd + 8 * UC);                    // DX: row tiles 2, 3 of W_ih^T per wave [2][4 waves][4 s][2][64 lanes]
    unsigned* kb = reinterpret_cast<unsigned*>(wxl + (DX ? 2 * 4 * 4 * 2 * 64 : 0));  // keep != 0 bits [T][2]
    __shared__ int s_dead, s_same;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, rg = lane >> 4;
    const int b = blockIdx.x;
    const int cl = (b >> 6) * 8 + (b & 7), cu = (b >> 3) & 7;
    if (cl >= a.ncl) return;
    const int T = a.T, N = a.N, NH = a.NH;
    const int ev = tid >> 2, q8 = (tid & 3) * 8;            // element view: env ev of the tile, own units q8 .. q8 + 7
    const int ug = UC * cu + q8;                           // their global index
    if (tid == 0) s_dead = 0;

    // ---- resident weights (A operand): rows = the 64 output units of this wave, K = this workgroup's 128 gate rows in the
    // order (gate s, own unit): lane (j, rg) of row tile rt feeds output unit 64 w + 16 rt + j, k = 32 s + 8 rg + i
    f16x8 wa[4][4][2];
    f16x8 wxa[2][4][2];                // (DX only; dead otherwise) row tiles 0, 1: 2 x 128 weight registers would take EVERY AGPR and the
                                       // compiler then spilled 40 of them to scratch; row tiles 2, 3 are read from LDS (16 reads per step)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int col = 64 * w + 16 * rt + j;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = a.w_hh[(size_t)(s * H + UC * cu + 8 * rg + i) * H + col];
            c8_split8(v, wa[rt][s][0], wa[rt][s][1]);
            asm volatile("" : "+a"(wa[rt][s][0]));
            asm volatile("" : "+a"(wa[rt][s][1]));
            if (DX) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = a.w_ih[(size_t)(s * H + UC * cu + 8 * rg + i) * H + col];
                if (rt < 2) {
                    c8_split8(v, wxa[rt][s][0], wxa[rt][s][1]);
                    asm volatile("" : "+a"(wxa[rt][s][0]));
                    asm volatile("" : "+a"(wxa[rt][s][1]));
                } else {
                    f16x8 p0, p1;
                    c8_split8(v, p0, p1);
                    wxl[((((rt - 2) * 4 + w) * 4 + s) * 2 + 0) * 64 + lane] = p0;
                    wxl[((((rt - 2) * 4 + w) * 4 + s) * 2 + 1) * 64 + lane] = p1;
                }
            }
        }
    if (TOP)
        for (int i = tid; i < 8 * UC; i += 256) whd[i] = (i / UC < NH) ? a.w_head[(size_t)(i / UC) * H + UC * cu + (i % UC)] : 0.f;

    const __amdgpu_buffer_rsrc_t xp_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.xp, 0, a.xp_bytes, 0x00020000);
    unsigned* my_flag = a.flags + (size_t)(cl * G + cu) * FLAG_STRIDE;
    const unsigned* peer_flag = a.flags + (size_t)(cl * G + (lane & 7)) * FLAG_STRIDE;
    bool dead = (a.abl & 1u) != 0;
    unsigned gstep = 0;
    bool same_xcd = false;
    {   // placement of the cluster, measured (see the forward kernel)
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 0xfu;
        if (tid == 0) __hip_atomic_store(my_flag + 1, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w == 0) {
            unsigned v = lane < G ? 0u : 1u, spins = 0;
            while (true) {
                if (v == 0u) v = __hip_atomic_load(peer_flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_ballot_w64(v == 0u) == 0ull) break;
                if (++spins > SPIN_LIMIT) {
                    if (lane == 0) { atomicAdd(a.err, 1u); s_dead = 1; }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            const bool mine = lane >= G || v == xcc + 1u;
            if (lane == 0) s_same = (__builtin_amdgcn_ballot_w64(!mine) == 0ull && !(a.abl & 0x10u)) ? 1 : 0;
        }
        __syncthreads();
        same_xcd = s_same != 0;
        dead = dead || s_dead != 0;
    }

    for (int tile = cl; tile < a.ntile; tile += a.ncl) {
        const int e0 = tile * E;
        const int n = e0 + ev, nc = min(n, N - 1);
        const bool live = n < N;
        __syncthreads();
        for (int t = w; t < T; t += 4) {
            const bool kv = a.keep ? a.keep[(size_t)min(e0 + lane, N - 1) * T + t] != 0.f : true;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(kv);
            if (lane == 0) { kb[2 * t] = (unsigned)m; kb[2 * t + 1] = (unsigned)(m >> 32); }
        }
        float dh_rec[8], dc_nx[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            dh_rec[i] = a.dhn ? a.dhn[(size_t)nc * H + ug + i] : 0.f;
            dc_nx[i] = a.dcn ? a.dcn[(size_t)nc * H + ug + i] : 0.f;
        }
        // the step's inputs of this thread's (env, 8 units): gates i f g o, c_prev (stash row) and dy -- or the row's dheads
        f32x4 sv[5][2], dyv[2];          // (kept in the loads' own vector type: a conversion here would wait for them on the spot)
        float dhd[8];
        auto load_step = [&](int t) {
            const size_t row = (size_t)nc * T + t;
            const float* sp = a.stash + row * (6 * H) + ug;
#pragma unroll
            for (int g = 0; g < 5; ++g) {
                // (nontemporal: streamed once -- keep the L2 for the hand-off blocks)
                sv[g][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sp + g * H));
                sv[g][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sp + g * H + 4));
            }
            if (!TOP) {
                dyv[0] = *reinterpret_cast<const f32x4*>(a.dy + row * H + ug);
                dyv[1] = *reinterpret_cast<const f32x4*>(a.dy + row * H + ug + 4);
            } else {
#pragma unroll
                for (int h = 0; h < 8; ++h) dhd[h] = a.dheads[row * NH + min(h, NH - 1)];
            }
        };
        load_step(T - 1);
        __syncthreads();

#ifdef UAV_C8_PROFILE
        last_ = __builtin_readcyclecounter();
#endif
        for (int t = T - 1; t >= 0; --t, ++gstep) {
            const uint2 mk = *reinterpret_cast<const uint2*>(kb + 2 * t);
            const float kp = (((ev >> 5) ? mk.y : mk.x) >> (ev & 31)) & 1u ? 1.f : 0.f;
            // ---- cell backward (train_ppo2.0.py:85's autograd through nn.LSTM's cell, as cell_bwd_h3_kernel)
            float gi[8], gf[8], gg[8], go[8], cp[8], dyr[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    gi[4 * h + r] = sv[0][h][r]; gf[4 * h + r] = sv[1][h][r]; gg[4 * h + r] = sv[2][h][r]; go[4 * h + r] = sv[3][h][r];
                    cp[4 * h + r] = sv[4][h][r];
                }
            }
            if (!TOP) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { dyr[r] = dyv[0][r]; dyr[4 + r] = dyv[1][r]; }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) dyr[i] = 0.f;
                for (int h = 0; h < NH; ++h) {
                    const float d = dhd[h];
#pragma unroll
                    for (int i = 0; i < 8; ++i) dyr[i] = fmaf(d, whd[h * UC + q8 + i], dyr[i]);
                }
            }
            float g4[4][8], m = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float dh = dyr[i] + dh_rec[i];
                const float c = gf[i] * cp[i] + gi[i] * gg[i];
                const float tch = fast_tanh(c);
                const float dc = dh * go[i] * (1.0f - tch * tch) + dc_nx[i];
                g4[0][i] = dc * gg[i] * gi[i] * (1.0f - gi[i]);
                g4[1][i] = dc * cp[i] * gf[i] * (1.0f - gf[i]);
                g4[2][i] = dc * gi[i] * (1.0f - gg[i] * gg[i]);
                g4[3][i] = dh * tch * go[i] * (1.0f - go[i]);
                dc_nx[i] = dc * gf[i] * kp;
#pragma unroll
                for (int g = 0; g < 4; ++g) m = fmaxf(m, fabsf(g4[g][i]));
            }
            C8_STAMP(0);
            if (live && !(a.abl & 2u)) {
                float* gp = a.dgates + ((size_t)n * T + t) * (4 * H) + ug;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    __builtin_nontemporal_store(f32x4{g4[g][0], g4[g][1], g4[g][2], g4[g][3]}, reinterpret_cast<f32x4*>(gp + g * H));
                    __builtin_nontemporal_store(f32x4{g4[g][4], g4[g][5], g4[g][6], g4[g][7]}, reinterpret_cast<f32x4*>(gp + g * H + 4));
                }
            }
            C8_STAMP(1);
            // the env's 128 values of this workgroup: max over its four lanes; power of two 2^e with m 2^e in [2^13, 2^14)
            m = fmaxf(m, __shfl_xor(m, 1, 64));
            m = fmaxf(m, __shfl_xor(m, 2, 64));
            int ex = 0;
            if (m > 0.f && m < 3.0e38f) {
                ex = 13 - (int)((__float_as_uint(m) >> 23) & 0xff) + 127;
                ex = ex > 100 ? 100 : (ex < -100 ? -100 : ex);
            }
            const float sc = __uint_as_float((unsigned)(127 + ex) << 23), isc = __uint_as_float((unsigned)(127 - ex) << 23);
            if ((tid & 3) == 0) { isc_h[ev] = isc * kp; isc_x[ev] = isc; }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = g4[g][i] * sc;
                f16x8 p0, p1;
                c8_split8(v, p0, p1);
                *reinterpret_cast<f16x8*>(gpl + ev * RSG + 32 * g + q8) = p0;
                *reinterpret_cast<f16x8*>(gpl + GPLANE + ev * RSG + 32 * g + q8) = p1;
            }
            C8_STAMP(2);
            lds_barrier();                   // B1: the tile's dG pieces and scales are in LDS
            C8_STAMP(3);
            // next step's rows NOW: their HBM latency passes under the products, and nothing slow is in flight when wave 0 polls
            // (a poll's flag load returns behind every older load of its wave: issued after the flag they cost 12 k cycles a step)
            if (t > 0) load_step(t - 1);

            // ---- partial dh_{t-1} (and dx_t) of ALL 256 units from this workgroup's 128 gate rows
            const unsigned par = (gstep & 1u) * (unsigned)(a.ncl * G * G * PBLK);
            auto product = [&](auto wfrag, const float* iscale, unsigned half_off) {
#pragma unroll
              for (int ch = 0; ch < NCT; ch += 2) {      // two column tiles at a time: 64 accumulator registers beside the rows in flight
                f32x4 acc[4][NCT], acl[4][NCT];           // (only [.][ch], [.][ch + 1] are live)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int c = ch; c < ch + 2; ++c) acc[rt][c] = acl[rt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (!(a.abl & 8u)) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        f16x8 w0[4], w1[4];
#pragma unroll
                        for (int rt = 0; rt < 4; ++rt) wfrag(rt, s, w0[rt], w1[rt]);
#pragma unroll
                        for (int c = ch; c < ch + 2; ++c) {
                            const unsigned short* gr = gpl + (16 * c + j) * RSG + 32 * s + 8 * rg;
                            const f16x8 b0 = c8_ldh8(gr), b1 = c8_ldh8(gr + GPLANE);
#pragma unroll
                            for (int rt = 0; rt < 4; ++rt) {
                                acl[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[rt], b0, acl[rt][c], 0, 0, 0);
                                acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[rt], b0, acc[rt][c], 0, 0, 0);
                                acl[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[rt], b1, acl[rt][c], 0, 0, 0);
                            }
                        }
                    }
                }
                // lane (j, rg) of tile (rt, c): output units 64 w + 16 rt + 4 rg .. + 3 of env 16 c + j.  A row-tile PAIR (rt even, odd)
                // is the 32 units of one owner: 128 bytes per env, of which a lane holds two 16-byte quarters 64 bytes apart.  The
                // odd tile's quarters are exchanged between lanes j and j ^ 8, so that ONE store instruction
                // writes whole 128-byte lines of 8 envs (lanes j < 8: the even tile's quarter of env j; lanes j >= 8: the odd tile's
                // quarter of env j - 8) and a second one the other 8 envs -- instead of four instructions of 64-byte half lines
#pragma unroll
                for (int c = ch; c < ch + 2; ++c) {
                    const float is = iscale[16 * c + j];
#pragma unroll
                    for (int rp = 0; rp < 2; ++rp) {
                        const f32x4 ve = (acc[2 * rp][c] + acl[2 * rp][c] * H3_LO) * is;
                        const f32x4 vo = (acc[2 * rp + 1][c] + acl[2 * rp + 1][c] * H3_LO) * is;
                        u32x4 rot;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            rot[r] = __builtin_bit_cast(unsigned, __shfl_xor(vo[r], 8, 64));      // (a DPP row_ror:8 of these MFMA results read wrong values)
                        const u32x4 eb = __builtin_bit_cast(u32x4, ve);
                        const int owner = 2 * w + rp;
                        const unsigned blk = par + (unsigned)(((cl * G + owner) * G + cu) * PBLK);
                        const bool lo8 = j < 8;
                        // instruction 1: envs 16 c + 0..7; instruction 2: envs 16 c + 8..15
                        const u32x4 d1 = lo8 ? eb : rot, d2 = lo8 ? rot : eb;
                        const unsigned o1 = (blk + (unsigned)((16 * c + (j & 7)) * UC + (lo8 ? 0 : 16) + 4 * rg)) * 4u + half_off;
                        const unsigned o2 = (blk + (unsigned)((16 * c + 8 + (j & 7)) * UC + (lo8 ? 16 : 0) + 4 * rg)) * 4u + half_off;
                        if (same_xcd) {
                            __builtin_amdgcn_raw_buffer_store_b128(d1, xp_rsrc, o1, 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b128(d2, xp_rsrc, o2, 0, 0);
                        } else {
                            __builtin_amdgcn_raw_buffer_store_b128(d1, xp_rsrc, o1, 0, 16);
                            __builtin_amdgcn_raw_buffer_store_b128(d2, xp_rsrc, o2, 0, 16);
                        }
                    }
                }
              }
            };
            product([&](int rt, int s, f16x8& p0, f16x8& p1) { p0 = wa[rt][s][0]; p1 = wa[rt][s][1]; }, isc_h, 0u);
            if constexpr (DX)
                product([&](int rt, int s, f16x8& p0, f16x8& p1) {
                    if (rt < 2) { p0 = wxa[rt][s][0]; p1 = wxa[rt][s][1]; }
                    else { p0 = wxl[((((rt - 2) * 4 + w) * 4 + s) * 2 + 0) * 64 + lane]; p1 = wxl[((((rt - 2) * 4 + w) * 4 + s) * 2 + 1) * 64 + lane]; }
                }, isc_x, a.xq_off);
            C8_STAMP(4);
            // ---- publish
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            C8_STAMP(5);
            __builtin_amdgcn_s_barrier();    // B2 (also: every wave has read the dG pieces and scales of this step)
            C8_STAMP(6);
            if (tid == 0) __hip_atomic_store(my_flag, gstep + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (w == 0 && !dead) {
                bool ok = lane >= G || lane == cu;
                unsigned spins = 0;
                while (true) {
                    if (!ok) ok = __hip_atomic_load(peer_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gstep + 1u;
                    if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                    if (++spins > SPIN_LIMIT) {
                        if (lane == 0) { atomicAdd(a.err, 1u); s_dead = 1; }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            C8_STAMP(7);
            __builtin_amdgcn_s_barrier();    // B3
            C8_STAMP(8);
            dead = s_dead != 0 || (a.abl & 1u);
            // ---- this workgroup's units: sum of the eight source blocks, fixed order
            auto gather = [&](auto policy, unsigned half_off, float (&out)[8]) {
                constexpr int AUX = decltype(policy)::value;
                u32x4 pv[G][2];
#pragma unroll
                for (int src = 0; src < G; ++src)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        pv[src][h] = __builtin_amdgcn_raw_buffer_load_b128(
                            xp_rsrc, (par + (unsigned)(((cl * G + cu) * G + src) * PBLK + ev * UC + q8 + 4 * h)) * 4u + half_off, 0, AUX);
#pragma unroll
                for (int i = 0; i < 8; ++i) out[i] = 0.f;
#pragma unroll
                for (int src = 0; src < G; ++src)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x4 v = __builtin_bit_cast(f32x4, pv[src][h]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) out[4 * h + r] += v[r];
                    }
            };
            if (!(a.abl & 4u)) {
                if (same_xcd) gather(std::integral_constant<int, 2>{}, 0u, dh_rec);
                else gather(std::integral_constant<int, 16>{}, 0u, dh_rec);
                if (DX) {
                    float dxv[8];
                    if (same_xcd) gather(std::integral_constant<int, 2>{}, a.xq_off, dxv);
                    else gather(std::integral_constant<int, 16>{}, a.xq_off, dxv);
                    if (live && !(a.abl & 2u)) {
                        float* xo = a.dx + ((size_t)n * T + t) * H + ug;
                        *reinterpret_cast<float4*>(xo) = float4{dxv[0], dxv[1], dxv[2], dxv[3]};
                        *reinterpret_cast<float4*>(xo + 4) = float4{dxv[4], dxv[5], dxv[6], dxv[7]};
                    }
                }
            }
            C8_STAMP(9);
        }
        if (live) {
            if (a.dh0) {
                *reinterpret_cast<float4*>(a.dh0 + (size_t)n * H + ug) = float4{dh_rec[0], dh_rec[1], dh_rec[2], dh_rec[3]};
                *reinterpret_cast<float4*>(a.dh0 + (size_t)n * H + ug + 4) = float4{dh_rec[4], dh_rec[5], dh_rec[6], dh_rec[7]};
            }
            if (a.dc0) {
                *reinterpret_cast<float4*>(a.dc0 + (size_t)n * H + ug) = float4{dc_nx[0], dc_nx[1], dc_nx[2], dc_nx[3]};
                *reinterpret_cast<float4*>(a.dc0 + (size_t)n * H + ug + 4) = float4{dc_nx[4], dc_nx[5], dc_nx[6], dc_nx[7]};
            }
        }
    }
#ifdef UAV_C8_PROFILE
    if (blockIdx.x == 0 && tid == 0)
        for (int k = 0; k < 12; ++k) reinterpret_cast<unsigned long long*>(a.err + 2)[k] = prof_[k];
#endif
}

// ------------------------------------------------------------------------------------------------ host side
static size_t c8_lds_bytes(bool wide, int T) {      // a first layer double-buffers its planes
    return (size_t)(wide ? 1 : 2) * ((size_t)2 * c8::PLANE * 2 + (size_t)2 * c8::E * (wide ? 264 : 40) * 2) + (size_t)T * 8;
}

bool lstm_c8_fits(int I, int T) { return c8_lds_bytes(I == 256, T) <= (160u << 10) - 64; }
bool lstm_c8_ok(const uav_ctx* ctx, int I, int H) {
    return H == 256 && (I <= 32 || I == 256) && ctx->num_cu >= 8 * 8 && uav_debug(UAV_DEBUG_CLUSTER);      // (T: see lstm_c8_fwd_fits)
}

// scratch of the cluster kernels inside the workspace tail handed over by the caller: exchange buffer | flags | (err lives in ctx)
size_t lstm_c8_scratch_bytes() {
    using namespace c8;
    return (size_t)2 * MAX_CLUSTERS * G * 2 * XBLK * 2 + (size_t)MAX_CLUSTERS * G * FLAG_STRIDE * 4;
}

int lstm_c8_fwd(uav_ctx* ctx, const float* x, int I, const float* w_ih, const float* b_ih, const float* b_hh, const float* keep,
                const float* h0, const float* c0, const float* w_hh, int N, int T, float* y, float* hn, float* cn, float* stash,
                hipStream_t st) {
    using namespace c8;
    const bool wide = I == 256;
    const int ntile = (N + E - 1) / E;
    int max_cl = ctx->num_cu / G;
    if (max_cl > MAX_CLUSTERS) max_cl = MAX_CLUSTERS;
    const int ncl = ntile < max_cl ? ntile : max_cl;
    const size_t need = lstm_c8_scratch_bytes();
    UAV_REQUIRE(need + (64u << 20) <= ctx->ws_bytes, "lstm cluster: workspace too small");
    char* base = (char*)ctx->ws + ctx->ws_bytes - need;
    C8FwdArgs a;
    a.x = x; a.w_ih = w_ih; a.w_hh = w_hh; a.b_ih = b_ih; a.b_hh = b_hh; a.keep = keep; a.h0 = h0; a.c0 = c0;
    a.y = y; a.stash = stash; a.hn = hn; a.cn = cn;
    a.xh = (unsigned short*)base;
    a.xh_bytes = (unsigned)((size_t)2 * MAX_CLUSTERS * G * 2 * XBLK * 2);
    a.flags = (unsigned*)(base + a.xh_bytes);
    a.err = ctx->cluster_err;
    a.I = I; a.N = N; a.T = T; a.ntile = ntile; a.ncl = ncl;
    a.abl = (g_uav_debug >> 8) & 0x1fu;
    UAV_CHECK_HIP(hipMemsetAsync(a.flags, 0, (size_t)MAX_CLUSTERS * G * FLAG_STRIDE * 4, st));
    const int grid = 64 * ((ncl + 7) / 8);
    const size_t lds = c8_lds_bytes(wide, T);
    UAV_REQUIRE(lds <= (160u << 10) - 64, "lstm cluster: T = %d needs %zu bytes of LDS", T, lds);
    if (wide) {
        UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(&lstm_fwd_c8_kernel<8>), 160 * 1024 - 64));
        hipLaunchKernelGGL(lstm_fwd_c8_kernel<8>, dim3(grid), dim3(256), lds, st, a);
    } else {
        UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(&lstm_fwd_c8_kernel<1>), 160 * 1024 - 64));
        hipLaunchKernelGGL(lstm_fwd_c8_kernel<1>, dim3(grid), dim3(256), lds, st, a);
    }
    UAV_LAUNCH_CHECK();
    return 0;
}

static size_t c8_bwd_lds_bytes(int T, bool dx = true) {
    return (size_t)2 * c8::GPLANE * 2 + (size_t)(2 * c8::E + 8 * c8::UC) * 4 + (dx ? (size_t)2 * 4 * 4 * 2 * 64 * 16 : 0) + (size_t)T * 8;
}
bool lstm_c8_bwd_fits(int T) { return c8_bwd_lds_bytes(T) <= (160u << 10) - 64; }

// one layer's BPTT as one cluster launch; dx (and w_ih) only for a layer that feeds one below (I = H)
int lstm_c8_bwd(uav_ctx* ctx, const float* keep, const float* stash, const float* w_hh, const float* dy, const float* dheads,
                const float* w_head, int n_heads, const float* dhn, const float* dcn, int N, int T, float* dgates, float* dh0,
                float* dc0, const float* w_ih, float* dx, hipStream_t st) {
    using namespace c8;
    UAV_REQUIRE(dy || (dheads && w_head && n_heads > 0 && n_heads <= 8), "lstm cluster bwd: dy or dheads + w_head (1..8 heads)");
    const int ntile = (N + E - 1) / E;
    int max_cl = ctx->num_cu / G;
    if (max_cl > MAX_CLUSTERS) max_cl = MAX_CLUSTERS;
    const int ncl = ntile < max_cl ? ntile : max_cl;
    const size_t half = (size_t)2 * MAX_CLUSTERS * G * G * PBLK * 4;          // bytes of one exchange buffer (both parities)
    const size_t flag_bytes = (size_t)MAX_CLUSTERS * G * FLAG_STRIDE * 4;
    const size_t need = 2 * half + flag_bytes;
    UAV_REQUIRE(need + (64u << 20) <= ctx->ws_bytes, "lstm cluster bwd: workspace too small");
    char* base = (char*)ctx->ws + ctx->ws_bytes - need;
    C8BwdArgs a;
    a.keep = keep; a.stash = stash; a.w_hh = w_hh; a.w_ih = w_ih; a.dy = dy; a.dheads = dheads; a.w_head = w_head;
    a.dhn = dhn; a.dcn = dcn; a.dgates = dgates; a.dx = dx; a.dh0 = dh0; a.dc0 = dc0;
    a.xp = (float*)base;
    a.xp_bytes = (unsigned)(2 * half);
    a.xq_off = (unsigned)half;
    a.flags = (unsigned*)(base + 2 * half);
    a.err = ctx->cluster_err;
    a.NH = n_heads; a.N = N; a.T = T; a.ntile = ntile; a.ncl = ncl;
    a.abl = (g_uav_debug >> 8) & 0x1fu;
    UAV_CHECK_HIP(hipMemsetAsync(a.flags, 0, flag_bytes, st));
    const int grid = 64 * ((ncl + 7) / 8);
    const size_t lds = c8_bwd_lds_bytes(T, dx != nullptr);
    UAV_REQUIRE(lds <= (160u << 10) - 64, "lstm cluster bwd: T = %d needs %zu bytes of LDS", T, lds);
    UAV_REQUIRE(!dx || w_ih, "lstm cluster bwd: dx needs w_ih");
#define C8_BWD_LAUNCH(DX_, TOP_)                                                                                            \
    do {                                                                                                                    \
        UAV_CHECK_HIP(uav_dyn_lds(reinterpret_cast<const void*>(&lstm_bwd_c8_kernel<DX_, TOP_>), 160 * 1024 - 64));         \
        hipLaunchKernelGGL((lstm_bwd_c8_kernel<DX_, TOP_>), dim3(grid), dim3(256), lds, st, a);                             \
    } while (0)
    if (dx) { if (dy) C8_BWD_LAUNCH(true, false); else C8_BWD_LAUNCH(true, true); }
    else { if (dy) C8_BWD_LAUNCH(false, false); else C8_BWD_LAUNCH(false, true); }
#undef C8_BWD_LAUNCH
    UAV_LAUNCH_CHECK();
    return 0;
}

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

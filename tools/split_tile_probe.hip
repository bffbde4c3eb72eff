// split_tile_probe.hip -- the "small-N schedule" experiment (VERDICT r2 item 5 / DESIGN 7 "small-N occupancy"):
// with fewer 16-env tiles than CUs (C2: 16, C4: 64 of 256) the sequence kernels take T x one workgroup's step latency
// whatever N is.  The only way to shorten a step is to split ONE tile's gate rows over TWO CUs, which puts an h_t
// hand-off through L2 on every step.  This probe measures exactly that at h = 128, forward recurrence (the structure of
// lstm_fwd_h3_kernel: W_hh as two fp16 pieces in VGPRs, h_t in two fp16 LDS planes, three MFMA products per K = 32
// slab, v_exp/v_rcp gate pointwise, one lane = one env x four units), no stash stores:
//
//   whole    one workgroup (8 waves) per 16-env tile                                   -- today's schedule
//   split    two workgroups (4 waves each) per tile, each owning 64 of the 128 units (all four gates of them, so the
//            cell stays local); per step each writes its 64 units of h_t (two fp16 planes, 4 KB) to a global slot,
//            releases a step flag, acquires the partner's flag and reads the partner's 4 KB
//            split-same: partners on the SAME XCD (block b and b + 8 share an L2)
//            split-cross: partners on different XCDs (block b and b + 1)
//
//   hipcc -O3 --offload-arch=gfx950 tools/split_tile_probe.hip -o tools/bin/split_tile_probe && tools/bin/split_tile_probe [tiles=64] [T=128]
// Prints microseconds per time step for each schedule, and checks that all three produce the same h_T.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int H = 128, MT = 16, NS = H / 32, RS = H + 8, PLANE = MT * RS;
constexpr float LO = 1.0f / 2048.0f;

__device__ __forceinline__ float fsig(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }
__device__ __forceinline__ void split2h(float a, _Float16& p0, _Float16& p1) {
    p0 = (_Float16)a;
    p1 = (_Float16)((a - (float)p0) * 2048.0f);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// NWV waves per workgroup, each owning 16 units (x 4 gates).  SPLIT: this workgroup owns units [64 half, 64 half + 64).
template <bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 256 : 512) void lstm_probe_kernel(const float* __restrict__ w_hh, const float* __restrict__ bias,
                                                                      const float* __restrict__ h0, float* __restrict__ h_out,
                                                                      int T, int pair_stride, unsigned short* __restrict__ xbuf,
                                                                      unsigned* __restrict__ flags, unsigned* __restrict__ err) {
    __shared__ __attribute__((aligned(16))) unsigned short hpl[2 * PLANE];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, kq = lane >> 4;
    int tile, half;
    if (SPLIT) {
        // pair_stride 8: blocks b and b + 8 (same XCD under round-robin placement) form a pair; 1: blocks b and b + 1
        if (pair_stride == 8) { tile = (blockIdx.x / 16) * 8 + (blockIdx.x % 8); half = (blockIdx.x / 8) & 1; }
        else { tile = blockIdx.x / 2; half = blockIdx.x & 1; }
    } else { tile = blockIdx.x; half = 0; }
    const int wu = (SPLIT ? 4 * half : 0) + w;                 // global wave index = 16-unit block
    const int uw = 16 * wu + j, uo = 16 * wu + 4 * kq;
    f16x8 wb[4][NS][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float* src = w_hh + (size_t)(q * H + uw) * H + 32 * s + 8 * kq;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                _Float16 p0, p1;
                split2h(src[i], p0, p1);
                wb[q][s][0][i] = p0; wb[q][s][1][i] = p1;
            }
        }
    f32x4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[q] = f32x4{bias[q * H + uo], bias[q * H + uo + 1], bias[q * H + uo + 2], bias[q * H + uo + 3]};
    float c_reg[4] = {0.f, 0.f, 0.f, 0.f};
    auto put_h = [&](const float (&hv)[4], unsigned short* gl) {            // split and park h[env j][uo .. uo+3]
        unsigned short b[2][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            _Float16 p0, p1;
            split2h(hv[r], p0, p1);
            b[0][r] = __builtin_bit_cast(unsigned short, p0); b[1][r] = __builtin_bit_cast(unsigned short, p1);
        }
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            uint2 v;
            v.x = (unsigned)b[pc][0] | ((unsigned)b[pc][1] << 16);
            v.y = (unsigned)b[pc][2] | ((unsigned)b[pc][3] << 16);
            *reinterpret_cast<uint2*>(hpl + pc * PLANE + j * RS + uo) = v;
            // [piece][env][64 units]; a relaxed AGENT-scope atomic store = a plain store with the coherent cache policy
            // (sc1): it is performed where the partner's coherent loads read it, no L2 write-back fence needed
            if (SPLIT) __hip_atomic_store(reinterpret_cast<unsigned long long*>(gl + pc * (MT * 64) + j * 64 + (uo - 64 * half)),
                                          (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    // h_{-1}: every workgroup loads the whole tile's h0 itself
    for (int i = threadIdx.x; i < MT * H; i += blockDim.x) {
        const int e = i / H, u = i % H;
        _Float16 p0, p1;
        split2h(h0[(size_t)(tile * MT + e) * H + u], p0, p1);
        hpl[e * RS + u] = __builtin_bit_cast(unsigned short, p0);
        hpl[PLANE + e * RS + u] = __builtin_bit_cast(unsigned short, p1);
    }
    lds_barrier();
    bool dead = false;
    float hh[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < T; ++t) {
        f32x4 acc[4], acl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc[q] = bq[q]; acl[q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const unsigned short* hrow = hpl + j * RS + 8 * kq;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const f16x8 a0 = *reinterpret_cast<const f16x8*>(hrow + 32 * s);
            const f16x8 a1 = *reinterpret_cast<const f16x8*>(hrow + PLANE + 32 * s);
#pragma unroll
            for (int q = 0; q < 4; ++q) acl[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[q][s][1], a0, acl[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[q][s][0], a0, acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acl[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[q][s][0], a1, acl[q], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = acc[q] + acl[q] * LO;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float gi = fsig(acc[0][r]), gf = fsig(acc[1][r]), gg = ftanh(acc[2][r]), go = fsig(acc[3][r]);
            c_reg[r] = gf * c_reg[r] + gi * gg;
            hh[r] = go * ftanh(c_reg[r]);
        }
        lds_barrier();                                   // every wave has read h_{t-1}
        unsigned short* mine = SPLIT ? xbuf + ((size_t)((t & 1) * gridDim.x / 2 + tile) * 2 + half) * (2 * MT * 64) : nullptr;
        put_h(hh, mine);
        if (SPLIT) {
            // release: __syncthreads() waits for every wave's coherent stores to be acknowledged (vmcnt(0)), then ONE lane
            // publishes the step flag.  (A first version used __threadfence() = buffer_wbl2 + buffer_inv in every thread:
            // 27 us per step at 64 tiles, growing with the number of workgroups -- the L2 write-back serialises.)
            __syncthreads();
            unsigned* fl = flags + tile * 2;
            if (threadIdx.x == 0) {
                __hip_atomic_store(fl + half, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // acquire the partner's flag (bounded: a partner that never shows up must not hang the GPU)
                if (!dead) {
                    unsigned spins = 0;
                    while (__hip_atomic_load(fl + (1 - half), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(t + 1)) {
                        if (++spins > (1u << 24)) { atomicAdd(err, 1u); dead = true; break; }
                    }
                }
            }
            __syncthreads();
            const unsigned short* theirs = xbuf + ((size_t)((t & 1) * gridDim.x / 2 + tile) * 2 + (1 - half)) * (2 * MT * 64);
            // 2 pieces x 16 envs x 64 units = 4 KB: 256 threads x 16 B, coherent (sc1) loads
            {
                const int pc = threadIdx.x >> 7, e = (threadIdx.x >> 3) & 15, u8 = (threadIdx.x & 7) * 8;
                const unsigned long long* src = reinterpret_cast<const unsigned long long*>(theirs + pc * (MT * 64) + e * 64 + u8);
                const unsigned long long v0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long v1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned long long* dst = reinterpret_cast<unsigned long long*>(hpl + pc * PLANE + e * RS + 64 * (1 - half) + u8);
                dst[0] = v0;
                dst[1] = v1;
            }
        }
        lds_barrier();                                   // h_t complete in LDS
    }
    // h_T of this workgroup's units (f32, for the cross-check)
    {
        const int e = tile * MT + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) h_out[(size_t)e * H + uo + r] = hh[r];
    }
}

// "halves": verdict r3 item 5.  ONE workgroup per 16-env tile as in `whole`, but the tile's two 8-env halves run half a step
// apart: while the gate arithmetic of half B's step t - 1 (lanes j >= 8) issues, the MFMAs of half A's step t are in flight, and
// vice versa.  Every product still computes all 16 columns (the other half's are ignored: the matrix pipe is mostly idle anyway) and
// every gate instruction runs with half its lanes masked -- twice the instructions, but the MFMA latency and the VALU work of ONE
// wave overlap (an MFMA holds the SIMD's vector issue for half its duration only).  Same arithmetic per env as `whole`: same h_T.
__global__ __launch_bounds__(512) void lstm_probe_halves_kernel(const float* __restrict__ w_hh, const float* __restrict__ bias,
                                                               const float* __restrict__ h0, float* __restrict__ h_out, int T) {
    __shared__ __attribute__((aligned(16))) unsigned short hpl[2 * PLANE];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, kq = lane >> 4;
    const int tile = blockIdx.x;
    const int uw = 16 * w + j, uo = 16 * w + 4 * kq;
    f16x8 wb[4][NS][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float* src = w_hh + (size_t)(q * H + uw) * H + 32 * s + 8 * kq;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                _Float16 p0, p1;
                split2h(src[i], p0, p1);
                wb[q][s][0][i] = p0; wb[q][s][1][i] = p1;
            }
        }
    f32x4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[q] = f32x4{bias[q * H + uo], bias[q * H + uo + 1], bias[q * H + uo + 2], bias[q * H + uo + 3]};
    float c_reg[4] = {0.f, 0.f, 0.f, 0.f}, hh[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < MT * H; i += blockDim.x) {
        const int e = i / H, u = i % H;
        _Float16 p0, p1;
        split2h(h0[(size_t)(tile * MT + e) * H + u], p0, p1);
        hpl[e * RS + u] = __builtin_bit_cast(unsigned short, p0);
        hpl[PLANE + e * RS + u] = __builtin_bit_cast(unsigned short, p1);
    }
    lds_barrier();
    const bool isA = j < 8;
    auto products = [&](f32x4 (&acc)[4], f32x4 (&acl)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc[q] = bq[q]; acl[q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const unsigned short* hrow = hpl + j * RS + 8 * kq;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const f16x8 a0 = *reinterpret_cast<const f16x8*>(hrow + 32 * s);
            const f16x8 a1 = *reinterpret_cast<const f16x8*>(hrow + PLANE + 32 * s);
#pragma unroll
            for (int q = 0; q < 4; ++q) acl[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[q][s][1], a0, acl[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[q][s][0], a0, acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acl[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[q][s][0], a1, acl[q], 0, 0, 0);
        }
    };
    // the gate arithmetic of the lanes of one half (`mine`), from that half's accumulators; parks h in LDS
    auto cell = [&](const f32x4 (&acc)[4], const f32x4 (&acl)[4], bool mine) {
        if (mine) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float gi = fsig(acc[0][r] + acl[0][r] * LO), gf = fsig(acc[1][r] + acl[1][r] * LO);
                const float gg = ftanh(acc[2][r] + acl[2][r] * LO), go = fsig(acc[3][r] + acl[3][r] * LO);
                c_reg[r] = gf * c_reg[r] + gi * gg;
                hh[r] = go * ftanh(c_reg[r]);
            }
            unsigned short b[2][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                _Float16 p0, p1;
                split2h(hh[r], p0, p1);
                b[0][r] = __builtin_bit_cast(unsigned short, p0); b[1][r] = __builtin_bit_cast(unsigned short, p1);
            }
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                uint2 v;
                v.x = (unsigned)b[pc][0] | ((unsigned)b[pc][1] << 16);
                v.y = (unsigned)b[pc][2] | ((unsigned)b[pc][3] << 16);
                *reinterpret_cast<uint2*>(hpl + pc * PLANE + j * RS + uo) = v;
            }
        }
    };
    f32x4 accA[4], aclA[4], accB[4], aclB[4];
    // prologue: half B's first products (its columns of the tile's h_{-1})
    products(accB, aclB);
    for (int t = 0; t < T; ++t) {
        // phase 1: half A's products of step t in flight under half B's cell of step t  (B's accumulators came from phase 2 of the
        // iteration before -- or the prologue -- and saw h_B(t-1))
        products(accA, aclA);
        cell(accB, aclB, !isA);
        lds_barrier();                                   // h_B(t) in LDS; nobody reads h_A(t-1) any more
        // phase 2: half B's products of step t + 1 in flight under half A's cell of step t
        if (t + 1 < T) products(accB, aclB);
        cell(accA, aclA, isA);
        lds_barrier();                                   // h_A(t) in LDS
    }
    {
        const int e = tile * MT + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) h_out[(size_t)e * H + uo + r] = hh[r];
    }
}

template <typename F>
static double time_ms(F&& launch, int reps = 7) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
    const int tiles = argc > 1 ? atoi(argv[1]) : 64, T = argc > 2 ? atoi(argv[2]) : 128;
    if (tiles < 8 || tiles % 8 || tiles > 128) { fprintf(stderr, "tiles must be a multiple of 8 in 8..128 (both halves of every pair must be resident)\n"); return 2; }
    const int N = tiles * MT;
    std::vector<float> w((size_t)4 * H * H), b(4 * H), h0((size_t)N * H);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : w) v = rnd() * 0.18f;
    for (auto& v : b) v = rnd() * 0.1f;
    for (auto& v : h0) v = rnd();
    float *dw, *db, *dh0, *dout[4];
    unsigned short* xbuf;
    unsigned *flags, *err;
    CK(hipMalloc(&dw, w.size() * 4)); CK(hipMalloc(&db, b.size() * 4)); CK(hipMalloc(&dh0, h0.size() * 4));
    for (auto& p : dout) CK(hipMalloc(&p, h0.size() * 4));
    CK(hipMalloc(&xbuf, (size_t)2 * tiles * 2 * (2 * MT * 64) * 2));
    CK(hipMalloc(&flags, tiles * 2 * 4)); CK(hipMalloc(&err, 4));
    CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dh0, h0.data(), h0.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(err, 0, 4));
    printf("h = %d, %d tiles of %d envs, T = %d steps; median of 7 launches\n", H, tiles, MT, T);
    const double whole = time_ms([&] { hipLaunchKernelGGL(lstm_probe_kernel<false>, dim3(tiles), dim3(512), 0, 0, dw, db, dh0, dout[0], T, 0, xbuf, flags, err); });
    printf("whole  (1 workgroup x 8 waves per tile)              : %.3f ms = %.3f us per step\n", whole, 1e3 * whole / T);
    const char* names[2] = {"split-same  (2 workgroups per tile, same XCD)     ", "split-cross (2 workgroups per tile, different XCDs)"};
    for (int v = 0; v < 2; ++v) {
        const int stride = v == 0 ? 8 : 1;
        const double t = time_ms([&] {
            CK(hipMemsetAsync(flags, 0, tiles * 2 * 4, 0));
            hipLaunchKernelGGL(lstm_probe_kernel<true>, dim3(2 * tiles), dim3(256), 0, 0, dw, db, dh0, dout[1 + v], T, stride, xbuf, flags, err);
        });
        printf("%s: %.3f ms = %.3f us per step  (%.2fx the whole-tile schedule)\n", names[v], t, 1e3 * t / T, whole / t);
    }
    const double halves = time_ms([&] { hipLaunchKernelGGL(lstm_probe_halves_kernel, dim3(tiles), dim3(512), 0, 0, dw, db, dh0, dout[3], T); });
    printf("halves (1 workgroup per tile, its two 8-env halves half a step apart): %.3f ms = %.3f us per step  (%.2fx the whole-tile schedule)\n",
           halves, 1e3 * halves / T, whole / halves);
    {
        std::vector<float> r0h(h0.size()), r3(h0.size());
        CK(hipMemcpy(r0h.data(), dout[0], r0h.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(r3.data(), dout[3], r3.size() * 4, hipMemcpyDeviceToHost));
        size_t badh = 0;
        double worst = 0.0;
        for (size_t i = 0; i < r0h.size(); ++i) {
            badh += (r0h[i] != r3[i]);
            const double d = fabs((double)r0h[i] - (double)r3[i]);
            if (d > worst) worst = d;
        }
        // (the two kernels are compiled with contraction on: acc + acl * LO fuses in one and not in the other, hence last-bit differences)
        printf("h_T differences halves vs whole: %zu of %zu values, largest %.3g\n", badh, r0h.size(), worst);
        if (worst > 1e-5) return 1;
    }
    unsigned herr = 0;
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    std::vector<float> r0(h0.size()), r1(h0.size()), r2(h0.size());
    CK(hipMemcpy(r0.data(), dout[0], r0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r1.data(), dout[1], r1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r2.data(), dout[2], r2.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < r0.size(); ++i) bad += (r0[i] != r1[i]) + (r0[i] != r2[i]);
    printf("partner time-outs: %u; h_T differences between the schedules: %zu of %zu values\n", herr, bad, 2 * r0.size());
    return (herr || bad) ? 1 : 0;
}

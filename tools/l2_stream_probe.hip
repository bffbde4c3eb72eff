// l2_stream_probe.hip -- can the L2 feed a PERSISTENT h = 256 step kernel that streams its weights every step?
//
// The candidate: one workgroup per 32-env tile owns all 1024 gate rows for all T steps; the recurrent state and the
// accumulators stay on chip; the weight pieces (fragment order, 1 KB per wave load) come from L2 every step: 1 MB for W_hh (K = 256), 2 MB
// with a hidden-wide W_ih (K = 512).  This probe times exactly that stream with the MFMAs it feeds and nothing else: NWG workgroups of 8
// waves (two per SIMD), wave w = gate row tiles 8 w .. 8 w + 7 x two 16-env column tiles (B operand constant in registers), per step and
// K slab 16 chunk loads (a0, a1 of 8 row tiles) + 48 MFMAs, a six-item register ring.  All workgroups read the SAME weights (L2-resident).
//
//   hipcc -O3 --offload-arch=gfx950 tools/l2_stream_probe.hip -o tools/bin/l2_stream_probe && tools/bin/l2_stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// w: [row tile 64][slab KS][piece 2][512 halves]
template <int KS, int MODE>      // MODE 0: loads + MFMAs, 1: loads only, 2: MFMAs only
__global__ __launch_bounds__(512) void stream_kernel(const unsigned short* __restrict__ w, int T, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: scalar bases
    const unsigned short* ub = w + (size_t)(8 * wv) * KS * 1024;
    const unsigned lo = lane * 8;
    f32x4 acc[8][2], acl[8][2];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            acc[r][c] = acl[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+a"(acc[r][c]), "+a"(acl[r][c]));          // accumulators live in AGPRs (tied inline-asm MFMAs below)
        }
#define P_MFMA(ACC, FA, FB) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(FA), "v"(FB))
    f16x8 b0[2], b1[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) { b0[c][i] = (_Float16)(0.001f * (lane + i + c)); b1[c][i] = (_Float16)(0.0001f * (lane - i)); }
    // items = (slab, row-tile pair): 4 chunk loads + 12 MFMAs each; a[rp] holds item (s, rp) and is refilled with (s + 1, rp) as soon as its
    // MFMAs are issued: three items (36 MFMAs) of flight time, static register indices under a rolled slab loop
    f16x8 a[4][2][2];
    auto fetch = [&](int s, int rp) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            a[rp][r][0] = *reinterpret_cast<const f16x8*>(ub + (((2 * rp + r) * KS + s) * 1024 + lo));
            a[rp][r][1] = *reinterpret_cast<const f16x8*>(ub + (((2 * rp + r) * KS + s) * 1024 + 512 + lo));
        }
    };
    for (int t = 0; t < T; ++t) {
        if (MODE != 2) {
#pragma unroll
            for (int rp = 0; rp < 4; ++rp) fetch(0, rp);
        }
#pragma unroll 1
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int rp = 0; rp < 4; ++rp) {
                if (MODE == 1) {
#pragma unroll
                    for (int r = 0; r < 2; ++r) asm volatile("" :: "v"(a[rp][r][0]), "v"(a[rp][r][1]));
                } else {
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            P_MFMA(acl[2 * rp + r][c], a[rp][r][1], b0[c]);
                            P_MFMA(acc[2 * rp + r][c], a[rp][r][0], b0[c]);
                            P_MFMA(acl[2 * rp + r][c], a[rp][r][0], b1[c]);
                        }
                }
                if (MODE != 2) fetch(s + 1 < KS ? s + 1 : s, rp);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                     // the step boundary of the real kernel (h_t exchanged through LDS)
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) s += acc[r][c][0] + acl[r][c][1];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

template <int KS, int MODE>
static float run(const unsigned short* w, int nwg, int T, float* out) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((stream_kernel<KS, MODE>), dim3(nwg), dim3(512), 0, 0, w, 8, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((stream_kernel<KS, MODE>), dim3(nwg), dim3(512), 0, 0, w, T, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / T;
}

int main() {
    const int T = 256;
    const size_t halves = (size_t)64 * 16 * 1024;                  // K = 512: 2 MB
    std::vector<unsigned short> h(halves);
    for (size_t i = 0; i < halves; ++i) h[i] = (unsigned short)(0x2000 + (i * 2654435761u >> 22 & 0x3ff));     // small finite fp16 values
    unsigned short* w;
    float* out;
    CHECK(hipMalloc(&w, halves * 2));
    CHECK(hipMalloc(&out, (size_t)256 * 1024 * 4));
    CHECK(hipMemcpy(w, h.data(), halves * 2, hipMemcpyHostToDevice));
    printf("persistent weight stream, us per step (8 waves per workgroup, one workgroup per CU)\n");
    printf("%-28s %12s %12s %12s\n", "", "loads+MFMA", "loads only", "MFMA only");
    for (int nwg : {64, 128, 256}) {
        printf("K = 256 (1 MB / step), %3d WG  %12.2f %12.2f %12.2f\n", nwg, run<8, 0>(w, nwg, T, out), run<8, 1>(w, nwg, T, out), run<8, 2>(w, nwg, T, out));
        printf("K = 512 (2 MB / step), %3d WG  %12.2f %12.2f %12.2f\n", nwg, run<16, 0>(w, nwg, T, out), run<16, 1>(w, nwg, T, out), run<16, 2>(w, nwg, T, out));
    }
    printf("(today's per-step launches: ~19 us layer 1, ~22-24 us layer 2, on 256 CUs)\n");
    return 0;
}

// simd_overlap_probe.hip -- which waves of a 512-thread workgroup share a SIMD, and do one wave's MFMAs run under another
// wave's VALU work on the same SIMD?  Each wave runs `role[w]`: 0 idle, 1 = a stream of independent fp16 MFMAs (8
// accumulators), 2 = a stream of independent f32 FMAs, 3 = transcendentals (v_exp_f32).  One workgroup; cycles per wave.
//
//   hipcc -O3 --offload-arch=gfx950 tools/simd_overlap_probe.hip -o tools/bin/simd_overlap_probe && tools/bin/simd_overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct Roles { int r[8]; };

__global__ __launch_bounds__(512) void probe(Roles roles, int iters, unsigned long long* cyc, float* sink) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = roles.r[w];
    f32x4 acc[8];
    for (int q = 0; q < 8; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 1e-3f); b[i] = (_Float16)1e-3f; }
    float v[8];
    for (int q = 0; q < 8; ++q) v[q] = lane * 1e-3f + q;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (role == 1) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[q], 0, 0, 0);
    } else if (role == 2) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = __builtin_fmaf(v[q], 1.0001f, 0.5f);      // 32 FMAs = 128 issue cycles per iteration
    } else if (role == 3) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = __builtin_amdgcn_exp2f(v[q] * 0.001f);        // 8 x 16 cycles
    }
    float s = 0.f;
    for (int q = 0; q < 8; ++q) s += acc[q][0] + acc[q][3] + v[q];
    asm volatile("" ::"v"(s));
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[w] = t1 - t0;
    sink[threadIdx.x] = s;
}

int main() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 64); hipMalloc(&sink, 512 * 4);
    const int iters = 2000;
    struct Case { const char* name; int r[8]; };
    const Case cases[] = {
        {"wave 0 MFMA alone                ", {1, 0, 0, 0, 0, 0, 0, 0}},
        {"waves 0,1 MFMA                   ", {1, 1, 0, 0, 0, 0, 0, 0}},
        {"waves 0,4 MFMA                   ", {1, 0, 0, 0, 1, 0, 0, 0}},
        {"waves 0,2 MFMA                   ", {1, 0, 1, 0, 0, 0, 0, 0}},
        {"wave 0 FMA alone                 ", {2, 0, 0, 0, 0, 0, 0, 0}},
        {"waves 0,4 FMA                    ", {2, 0, 0, 0, 2, 0, 0, 0}},
        {"wave 0 MFMA + wave 4 FMA         ", {1, 0, 0, 0, 2, 0, 0, 0}},
        {"wave 0 MFMA + wave 1 FMA         ", {1, 2, 0, 0, 0, 0, 0, 0}},
        {"wave 0 exp alone                 ", {3, 0, 0, 0, 0, 0, 0, 0}},
        {"wave 0 MFMA + wave 4 exp         ", {1, 0, 0, 0, 3, 0, 0, 0}},
        {"wave 0 FMA + wave 4 exp          ", {2, 0, 0, 0, 3, 0, 0, 0}},
        {"all 8 MFMA                       ", {1, 1, 1, 1, 1, 1, 1, 1}},
        {"0-3 MFMA, 4-7 FMA                ", {1, 1, 1, 1, 2, 2, 2, 2}},
    };
    for (const Case& c : cases) {
        Roles r; memcpy(r.r, c.r, sizeof(r.r));
        unsigned long long h[8];
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, r, iters, d, sink);
            hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        }
        printf("%s cycles/iter per wave:", c.name);
        for (int w = 0; w < 8; ++w) printf(" %6.1f", (double)h[w] / iters);
        printf("\n");
    }
    return 0;
}

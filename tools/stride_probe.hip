// stride_probe.hip -- does the per-step LSTM path (h = 256) pay for its [env][time][..] layout?  One time step of
// cell_bwd_h3_kernel reads 5 KB of stash per env and writes 4 KB of gate gradients per env; with the env-major layout the
// 4096 envs of a step sit 1.5 MB (stash) and 1 MB (dG) apart, with a time-major layout they would be contiguous.
// The probe moves exactly those bytes (one wave per env, 16 envs per workgroup, float4 per lane) at both strides.
//   hipcc -O3 --offload-arch=gfx950 tools/stride_probe.hip -o tools/bin/stride_probe && tools/bin/stride_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(1024) void step_kernel(const float* __restrict__ src, size_t rstride, float* __restrict__ dst, size_t wstride, int N) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 16 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* sp = src + (size_t)n * rstride + 4 * lane;
    float4 v[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) v[q] = *reinterpret_cast<const float4*>(sp + q * 256);
    float* dp = dst + (size_t)n * wstride + 4 * lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 o = v[q];
        o.x += v[4].x; o.y += v[4].y; o.z += v[4].z; o.w += v[4].w;
        *reinterpret_cast<float4*>(dp + q * 256) = o;
    }
}

int main() {
    const int N = 4096, T = 256, H = 256;
    const size_t stash_floats = (size_t)N * T * 6 * H, dg_floats = (size_t)N * T * 4 * H;
    float *stash, *dg;
    CK(hipMalloc(&stash, stash_floats * 4));
    CK(hipMalloc(&dg, dg_floats * 4));
    CK(hipMemset(stash, 0, stash_floats * 4));
    CK(hipMemset(dg, 0, dg_floats * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct { const char* name; size_t rs, ws, tr, tw; } cases[] = {
        {"env-major  (envs 1.5 MB / 1 MB apart; step t at + t rows)", (size_t)T * 6 * H, (size_t)T * 4 * H, (size_t)6 * H, (size_t)4 * H},
        {"time-major (a step's 4096 envs contiguous: 25 MB / 17 MB)", (size_t)6 * H, (size_t)4 * H, (size_t)N * 6 * H, (size_t)N * 4 * H},
    };
    for (auto& c : cases) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            for (int t = T - 1; t >= 0; --t)
                hipLaunchKernelGGL(step_kernel, dim3(N / 16), dim3(1024), 0, 0, stash + t * c.tr, c.rs, dg + t * c.tw, c.ws, N);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%-62s %7.2f us per step  (%.2f TB/s of 37.7 MB)\n", c.name, ms * 1e3 / T, 37.7e6 / (ms * 1e-3 / T) / 1e12);
        }
    }
    return 0;
}

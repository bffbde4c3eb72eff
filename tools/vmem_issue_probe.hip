// vmem_issue_probe.hip -- what does ISSUING a global load cost the wave?  Each wave runs ITER rounds of NL independent loads
// (addresses inside a small window: L2 / L1 hits after the first round) and stamps s_memtime before and after the ISSUE of the
// burst (no s_waitcnt in between), then waits.  Reported: ticks per load instruction for dword / dwordx2 / dwordx4, with 1, 2, 4, 8
// waves per CU issuing at the same time (one workgroup per CU).
//   hipcc -O3 --offload-arch=gfx950 tools/vmem_issue_probe.hip -o tools/bin/vmem_issue_probe && tools/bin/vmem_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int W>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ src, unsigned long long* __restrict__ out, float* __restrict__ sink, int iters) {
    constexpr int NL = 8;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* p = src + ((size_t)blockIdx.x * 8 + w) * 64 * 1024 + lane * W;      // 256 KB per wave, lanes contiguous
    unsigned long long issue = 0, total = 0;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const float* q = p + (it & 15) * 4096;
        float v[NL][W];
        const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            if (W == 1) { float x; asm volatile("global_load_dword %0, %1, off" : "=v"(x) : "v"(q + i * 64 * W)); v[i][0] = x; }
            if (W == 2) { f32x2 x; asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(x) : "v"(q + i * 64 * W)); v[i][0] = x[0]; v[i][W - 1] = x[1]; }
            if (W == 4) { f32x4 x; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x) : "v"(q + i * 64 * W)); v[i][0] = x[0]; v[i][W - 1] = x[3]; }
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < NL; ++i) acc += v[i][0] + v[i][W - 1];
        if (it >= 16) { issue += t1 - t0; total += t2 - t0; }
    }
    if (lane == 0) { out[(blockIdx.x * 8 + w) * 2] = issue; out[(blockIdx.x * 8 + w) * 2 + 1] = total; }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    const int NB = 256, iters = 16 + 256;
    float* src; unsigned long long* out; float* sink;
    CK(hipMalloc(&src, (size_t)NB * 8 * 64 * 1024 * 4 + (1 << 20)));
    CK(hipMemset(src, 0, (size_t)NB * 8 * 64 * 1024 * 4 + (1 << 20)));
    CK(hipMalloc(&out, NB * 8 * 2 * 8)); CK(hipMalloc(&sink, 4));
    unsigned long long* h = (unsigned long long*)malloc(NB * 8 * 2 * 8);
    printf("ticks (s_memtime) per load instruction, 8 loads per burst, averaged over 256 bursts and all waves; 256 workgroups\n");
    printf("%-10s %6s %12s %14s\n", "width", "waves", "issue/load", "issue+wait/load");
    for (int W : {1, 2, 4})
        for (int waves : {1, 2, 4, 8}) {
            CK(hipMemset(out, 0, NB * 8 * 2 * 8));
            if (W == 1) hipLaunchKernelGGL(probe<1>, dim3(NB), dim3(64 * waves), 0, 0, src, out, sink, iters);
            if (W == 2) hipLaunchKernelGGL(probe<2>, dim3(NB), dim3(64 * waves), 0, 0, src, out, sink, iters);
            if (W == 4) hipLaunchKernelGGL(probe<4>, dim3(NB), dim3(64 * waves), 0, 0, src, out, sink, iters);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h, out, NB * 8 * 2 * 8, hipMemcpyDeviceToHost));
            double si = 0, st = 0; int n = 0;
            for (int b = 0; b < NB; ++b) for (int w = 0; w < waves; ++w) { si += h[(b * 8 + w) * 2]; st += h[(b * 8 + w) * 2 + 1]; ++n; }
            printf("dwordx%-4d %6d %12.1f %14.1f\n", W, waves, si / n / 256 / 8, st / n / 256 / 8);
        }
    return 0;
}

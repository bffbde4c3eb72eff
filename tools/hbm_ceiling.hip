// hbm_ceiling.hip -- measured HBM ceilings of this MI355X with HAND kernels ("confirm with a device-copy
// microbench and quote the measured ceiling"): float4 copy (read + write), float4 read-only (sum), float4 write-only,
// grid-stride over 2 GiB per array (>> the 256 MiB Infinity Cache), one dwordx4 per lane per trip.
//   hipcc -O3 --offload-arch=gfx950 tools/hbm_ceiling.hip -o tools/bin/hbm_ceiling && tools/bin/hbm_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int UNROLL>
__global__ __launch_bounds__(256) void copy_kernel(const f4* __restrict__ src, f4* __restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    for (size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n4; i += stride) {
        f4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) if (i + u * 256 < n4) v[u] = __builtin_nontemporal_load(src + i + u * 256);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) if (i + u * 256 < n4) __builtin_nontemporal_store(v[u], dst + i + u * 256);
    }
}
template <int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(const f4* __restrict__ src, float* __restrict__ out, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n4; i += stride) {
        f4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = (i + u * 256 < n4) ? __builtin_nontemporal_load(src + i + u * 256) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += (v[u].x + v[u].y) + (v[u].z + v[u].w);
    }
    if (acc == 1.2345e-30f) out[0] = acc;          // keeps the loads alive, never true in practice
}
template <int UNROLL>
__global__ __launch_bounds__(256) void write_kernel(f4* __restrict__ dst, size_t n4, float val) {
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    const f4 v = {val, val, val, val};
    for (size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n4; i += stride) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) if (i + u * 256 < n4) __builtin_nontemporal_store(v, dst + i + u * 256);
    }
}

template <typename F>
static double time_ms(F&& launch, int reps = 9) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch();
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

int main() {
    const size_t bytes = (size_t)2 << 30, n4 = bytes / 16;
    f4 *src, *dst;
    float* out;
    CK(hipMalloc(&src, bytes));
    CK(hipMalloc(&dst, bytes));
    CK(hipMalloc(&out, 4));
    CK(hipMemset(src, 1, bytes));
    CK(hipMemset(dst, 0, bytes));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs; arrays of %zu MiB, median of 9 launches each\n", prop.gcnArchName, cus, bytes >> 20);
    double best[3] = {0, 0, 0};
    for (int per_cu : {4, 8, 16, 32}) {
        const int grid = cus * per_cu;
        const double c = time_ms([&] { hipLaunchKernelGGL(copy_kernel<4>, dim3(grid), dim3(256), 0, 0, src, dst, n4); });
        const double r = time_ms([&] { hipLaunchKernelGGL(read_kernel<4>, dim3(grid), dim3(256), 0, 0, src, out, n4); });
        const double w = time_ms([&] { hipLaunchKernelGGL(write_kernel<4>, dim3(grid), dim3(256), 0, 0, dst, n4, 1.0f); });
        const double tc = 2.0 * bytes / c / 1e9, tr = 1.0 * bytes / r / 1e9, tw = 1.0 * bytes / w / 1e9;
        printf("grid %5d (%2d WG/CU): copy %.3f ms = %.2f TB/s (read+write) | read %.3f ms = %.2f TB/s | write %.3f ms = %.2f TB/s\n",
               grid, per_cu, c, tc, r, tr, w, tw);
        best[0] = std::max(best[0], tc); best[1] = std::max(best[1], tr); best[2] = std::max(best[2], tw);
    }
    printf("ceilings: f4 copy %.2f TB/s, read-only %.2f TB/s, write-only %.2f TB/s (spec 8.0)\n", best[0], best[1], best[2]);
    return 0;
}

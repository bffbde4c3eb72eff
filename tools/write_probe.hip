// write_probe.hip -- how fast can 256 workgroups x 8 waves stream the forward kernel's per-step output
// (y + 5 stash arrays, 16 envs x 6 x 128 floats per workgroup and step) to HBM when nothing else runs?
//   mode 0: the kernel's own pattern  [env][t][6H], lane = (env j, units 4kq..), 64-B pieces per row
//   mode 1: time-major                [t][env][6H]
//   mode 2: fully linear              each workgroup writes one contiguous 48-KB block per step
//   mode 3: pattern 0 with non-temporal stores
//   hipcc -O3 --offload-arch=gfx950 tools/write_probe.hip -o /tmp/wp && /tmp/wp
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(512) void wr(float* out, int N, int T) {
    const int H = 128, lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 15, kq = lane >> 4;
    const int n = blockIdx.x * 16 + j, uo = 16 * w + 4 * kq;
    const float4 v = {1.f * lane, 2.f, 3.f, 4.f * w};
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            size_t off;
            if (MODE == 0 || MODE == 3) off = ((size_t)n * T + t) * (6 * H) + q * H + uo;
            else if (MODE == 1) off = ((size_t)t * N + n) * (6 * H) + q * H + uo;
            else off = ((size_t)t * gridDim.x + blockIdx.x) * (16 * 6 * H) + (size_t)q * 16 * H + threadIdx.x * 4;
            typedef float f4v __attribute__((ext_vector_type(4)));
            if (MODE == 3) __builtin_nontemporal_store(f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<f4v*>(out + off));
            else *reinterpret_cast<float4*>(out + off) = v;
        }
        __syncthreads();
    }
}

int main() {
    const int N = 4096, T = 128;
    const size_t bytes = (size_t)N * T * 768 * 4;
    float* d;
    hipMalloc(&d, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            for (int k = 0; k < 5; ++k) {
                if (mode == 0) hipLaunchKernelGGL(wr<0>, dim3(N / 16), dim3(512), 0, 0, d, N, T);
                if (mode == 1) hipLaunchKernelGGL(wr<1>, dim3(N / 16), dim3(512), 0, 0, d, N, T);
                if (mode == 2) hipLaunchKernelGGL(wr<2>, dim3(N / 16), dim3(512), 0, 0, d, N, T);
                if (mode == 3) hipLaunchKernelGGL(wr<3>, dim3(N / 16), dim3(512), 0, 0, d, N, T);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms / 5 < best) best = ms / 5;
        }
        printf("mode %d: %.3f ms for %.2f GB -> %.2f TB/s\n", mode, best, bytes / 1e9, bytes / best / 1e9);
    }
    return 0;
}

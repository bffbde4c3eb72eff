// bf16x6_probe.hip -- does a 3-way bf16 split of both f32 operands, multiplied as six
// v_mfma_f32_16x16x32_bf16 products, reproduce an f32 GEMM to f32 rounding accuracy, and how fast is
// it against the exact v_mfma_f32_16x16x4_f32 chain?   (a = a0 + a1 + a2 with 8 significand bits per
// piece; the products a_i b_j with i + j <= 2 are kept, the dropped ones are below 2^-24 |a b|.)
//
//   hipcc -O3 --offload-arch=gfx950 tools/bf16x6_probe.hip -o gpurun_out/bf16x6_probe && gpurun_out/bf16x6_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float a, __bf16& p0, __bf16& p1, __bf16& p2) {
    p0 = (__bf16)a;
    const float r1 = a - (float)p0;
    p1 = (__bf16)r1;
    const float r2 = r1 - (float)p1;
    p2 = (__bf16)r2;
}

// C[16][16] = A[16][K] * B[K][16], K multiple of 32; one wave.  mode 0: f32 MFMA, 1: x6, 2: x3 (2 planes)
__global__ void probe(const float* A, const float* B, float* C, int K, int mode) {
    const int lane = threadIdx.x, j = lane & 15, kq = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (mode == 0) {
        for (int k = 0; k < K; k += 4)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j * K + k + kq], B[(k + kq) * 16 + j], acc, 0, 0, 0);
    } else {
        for (int k0 = 0; k0 < K; k0 += 32) {
            bf16x8 a[3], b[3];
            for (int i = 0; i < 8; ++i) {
                __bf16 p0, p1, p2;
                split3(A[j * K + k0 + kq * 8 + i], p0, p1, p2);
                a[0][i] = p0; a[1][i] = p1; a[2][i] = p2;
                split3(B[(k0 + kq * 8 + i) * 16 + j], p0, p1, p2);
                b[0][i] = p0; b[1][i] = p1; b[2][i] = p2;
            }
            if (mode == 1) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
        }
    }
    for (int r = 0; r < 4; ++r) C[(4 * kq + r) * 16 + j] = acc[r];
}

// throughput: 8 waves per WG, 256 WGs, each wave loops `iters` times over 4 tiles x 4 slabs
template <int MODE>
__global__ __launch_bounds__(512) void rate(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[4];
    for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
        float a = lane * 1e-3f, b = 1e-3f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int s = 0; s < 32; ++s)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
    } else {
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(lane * 1e-3f); b[i] = (__bf16)1e-3f; }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int s = 0; s < 4 * MODE; ++s)          // MODE = number of bf16 products per K=32 slab
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[q], 0, 0, 0);
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    const int K = 128;
    std::vector<float> A(16 * K), B(K * 16), C(256);
    srand(1);
    for (auto& v : A) v = (rand() / (float)RAND_MAX * 2 - 1);
    for (auto& v : B) v = (rand() / (float)RAND_MAX * 2 - 1) * 0.09f;
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    const char* names[3] = {"f32 mfma", "bf16 x6", "bf16 x3"};
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, mode);
        hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
        double emax = 0, scale = 0;
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                double ref = 0, mag = 0;
                for (int k = 0; k < K; ++k) { ref += (double)A[m * K + k] * B[k * 16 + n]; mag += fabs((double)A[m * K + k] * B[k * 16 + n]); }
                emax = fmax(emax, fabs(C[m * 16 + n] - ref));
                scale = fmax(scale, mag);
            }
        printf("%-9s max |err| %.3e  (sum|ab| %.3f -> rel %.3e, f32 eps 5.96e-08)\n", names[mode], emax, scale, emax / scale);
    }
    float* dout;
    hipMalloc(&dout, 256 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(512), 0, 0, dout, iters);
            if (mode == 1) hipLaunchKernelGGL(rate<6>, dim3(256), dim3(512), 0, 0, dout, iters);
            if (mode == 2) hipLaunchKernelGGL(rate<3>, dim3(256), dim3(512), 0, 0, dout, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 256.0 * 8 * iters * 2.0 * 16 * 64 * 128;       // algorithmic: 16 x 64 x K=128 per wave-iter
        printf("%-9s %.3f ms for %d iters -> %.1f algorithmic TFLOP/s (%.2f us per 16x512x128 step)\n", names[mode], ms,
               iters, flop / ms * 1e-9, ms * 1e3 / iters);
    }
    return 0;
}

// f16x3_probe.hip -- f32 products from THREE fp16 MFMA products: a = p0 + 2^-11 p1 with p0 = fp16(a),
// p1 = fp16((a - p0) * 2^11) (the residual is scaled back into fp16's normal range), so a is carried to 2^-24 |a|;
// a b ~ p0 q0 + 2^-11 (p0 q1 + p1 q0): the main product and the two cross products go to separate f32 accumulators,
// combined once at the end.  Compared with the exact-f32 MFMA chain and the six-product bf16 split (bf16x6_probe.hip)
// on (i) LSTM-like operands (weights +-0.09, h in (-1, 1)) and (ii) gradient-like operands with a wide dynamic range,
// block-scaled by a power of two.
//
//   hipcc -O3 --offload-arch=gfx950 tools/f16x3_probe.hip -o gpurun_out/f16x3_probe && gpurun_out/f16x3_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float a, __bf16& p0, __bf16& p1, __bf16& p2) {
    p0 = (__bf16)a;
    const float r1 = a - (float)p0;
    p1 = (__bf16)r1;
    p2 = (__bf16)(r1 - (float)p1);
}
__device__ __forceinline__ void split2h(float a, _Float16& p0, _Float16& p1) {
    p0 = (_Float16)a;
    p1 = (_Float16)((a - (float)p0) * 2048.0f);
}

// C[16][16] = A[16][K] * B[K][16]; one wave.  mode 0: f32 MFMA, 1: bf16 x6, 2: fp16 x3 (scaled residual), 3: fp16 x4
__global__ void probe(const float* A, const float* B, float* C, int K, int mode, float bscale) {
    const int lane = threadIdx.x, j = lane & 15, kq = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = acc, acc2 = acc;
    if (mode == 0) {
        for (int k = 0; k < K; k += 4)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j * K + k + kq], B[(k + kq) * 16 + j], acc, 0, 0, 0);
    } else if (mode == 1) {
        for (int k0 = 0; k0 < K; k0 += 32) {
            bf16x8 a[3], b[3];
            for (int i = 0; i < 8; ++i) {
                __bf16 p0, p1, p2;
                split3(A[j * K + k0 + kq * 8 + i], p0, p1, p2);
                a[0][i] = p0; a[1][i] = p1; a[2][i] = p2;
                split3(B[(k0 + kq * 8 + i) * 16 + j], p0, p1, p2);
                b[0][i] = p0; b[1][i] = p1; b[2][i] = p2;
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
        }
    } else {
        for (int k0 = 0; k0 < K; k0 += 32) {
            f16x8 a[2], b[2];
            for (int i = 0; i < 8; ++i) {
                _Float16 p0, p1;
                split2h(A[j * K + k0 + kq * 8 + i], p0, p1);
                a[0][i] = p0; a[1][i] = p1;
                split2h(B[(k0 + kq * 8 + i) * 16 + j] * bscale, p0, p1);      // power-of-two block scale: exact
                b[0][i] = p0; b[1][i] = p1;
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], acc1, 0, 0, 0);
            if (mode == 3) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[1], acc2, 0, 0, 0);
        }
        for (int r = 0; r < 4; ++r)
            acc[r] = (acc[r] + (1.0f / 2048.0f) * acc1[r] + (1.0f / 4194304.0f) * acc2[r]) / bscale;
    }
    for (int r = 0; r < 4; ++r) C[(4 * kq + r) * 16 + j] = acc[r];
}

template <int MODE>
__global__ __launch_bounds__(512) void rate(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[4];
    for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(lane * 1e-3f); b[i] = (__bf16)1e-3f; }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int s = 0; s < 4 * 6; ++s)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[q], 0, 0, 0);
    } else {
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 1e-3f); b[i] = (_Float16)1e-3f; }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int s = 0; s < 4 * 3; ++s)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[q], 0, 0, 0);
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

static double urand() { return rand() / (double)RAND_MAX; }

int main() {
    const int K = 128, TRIALS = 64;
    std::vector<float> A(16 * K), B(K * 16), C(256);
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024);
    const char* names[4] = {"f32 mfma", "bf16 x6", "fp16 x3", "fp16 x4"};
    for (int kind = 0; kind < 3; ++kind) {
        double emax[4] = {0, 0, 0, 0}, esq[4] = {0, 0, 0, 0}, mag2 = 0;
        long cnt = 0;
        srand(7 + kind);
        for (int tr = 0; tr < TRIALS; ++tr) {
            float bmax = 0.f;
            for (auto& v : A) v = (float)((urand() * 2 - 1) * 0.09);                       // weights
            for (auto& v : B) {
                if (kind == 0) v = (float)(urand() * 2 - 1);                               // h
                else if (kind == 1) v = (float)((urand() * 2 - 1) * 1e-6 * exp2(-20.0 * urand() * urand()));  // gradients, ~20 binades
                else v = (float)((urand() * 2 - 1) * 3e-5);       // all below fp16's smallest normal (6.1e-5), NOT scaled: subnormal pieces
                bmax = fmaxf(bmax, fabsf(v));
            }
            int e; frexpf(bmax, &e);
            const float bscale = kind != 1 ? 1.0f : ldexpf(1.0f, 14 - e);                  // max |B| -> [2^13, 2^14)
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            std::vector<double> ref(256), mag(256);
            for (int m = 0; m < 16; ++m)
                for (int n = 0; n < 16; ++n) {
                    double r = 0, g = 0;
                    for (int k = 0; k < K; ++k) { const double p = (double)A[m * K + k] * B[k * 16 + n]; r += p; g += p * p; }
                    ref[m * 16 + n] = r; mag[m * 16 + n] = sqrt(g);
                }
            for (int mode = 0; mode < 4; ++mode) {
                hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, mode, bscale);
                hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
                for (int i = 0; i < 256; ++i) {
                    const double er = (C[i] - ref[i]) / mag[i];          // error in units of the rms-sum of the terms
                    emax[mode] = fmax(emax[mode], fabs(er));
                    esq[mode] += er * er;
                }
            }
            cnt += 256;
        }
        (void)mag2;
        printf("%s operands, K=%d, %ld outputs: error / sqrt(sum (a b)^2)\n",
               kind == 0 ? "LSTM-like" : kind == 1 ? "gradient-like (block-scaled)" : "B below fp16's normal range, unscaled (do the MFMAs honour fp16 subnormals?)", K, cnt);
        for (int mode = 0; mode < 4; ++mode)
            printf("  %-9s max %.3e  rms %.3e\n", names[mode], emax[mode], sqrt(esq[mode] / cnt));
    }
    float* dout;
    hipMalloc(&dout, 256 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(512), 0, 0, dout, iters);
            if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(512), 0, 0, dout, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 256.0 * 8 * iters * 2.0 * 16 * 64 * 128;
        printf("%-9s %.3f ms for %d iters -> %.1f algorithmic TFLOP/s (%.2f us per 16x512x128 step)\n", mode ? "fp16 x3" : "bf16 x6", ms,
               iters, flop / ms * 1e-9, ms * 1e3 / iters);
    }
    return 0;
}

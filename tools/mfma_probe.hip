// mfma_probe.hip -- where do the cycles of the persistent LSTM step go?  (diagnostic, not shipped)
// 256 WGs x 512 threads; per "step": [LDS A-fragment reads] + 136 x v_mfma_f32_16x16x4_f32 with B in
// VGPRs, optional pointwise-like VALU, optional barrier.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int S = 152, SEG = 36, KS = 32;

template <int MODE>   // bit0: LDS reads, bit1: barrier, bit2: pointwise VALU, bit3: LDS write of h, bit4: 28 global stores
__global__ __launch_bounds__(512) void probe(const float* __restrict__ w, float* __restrict__ out, int steps) {
    __shared__ __attribute__((aligned(16))) float hbuf[2][16 * S];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, kq = lane >> 4;
    float wh[4][KS];
    for (int q = 0; q < 4; ++q)
        for (int s = 0; s < KS; ++s) wh[q][s] = w[(q * KS + s) * 512 + threadIdx.x];
    for (int i = threadIdx.x; i < 2 * 16 * S; i += 512) (&hbuf[0][0])[i] = 0.001f * i;
    __syncthreads();
    float c[4] = {0, 0, 0, 0};
    int cur = 0;
    for (int t = 0; t < steps; ++t) {
        f32x4 acc[4];
        for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.1f, 0.2f, 0.3f, 0.4f};
        const float* hrow = hbuf[cur] + j * S + kq * SEG;
#pragma unroll
        for (int s = 0; s < KS; s += 4) {
            float4 a;
            if (MODE & 1) a = *reinterpret_cast<const float4*>(hrow + s);
            else a = make_float4(c[0], c[1], c[2], c[3]);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wh[q][s], acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wh[q][s + 1], acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wh[q][s + 2], acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wh[q][s + 3], acc[q], 0, 0, 0);
        }
        // 8 more MFMAs (x part)
        for (int q = 0; q < 4; ++q) {
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(c[0], wh[q][0], acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(c[1], wh[q][1], acc[q], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float h;
            if (MODE & 4) {
                const float gi = 1.f / (1.f + __expf(-acc[0][r])), gf = 1.f / (1.f + __expf(-acc[1][r]));
                const float gg = 1.f - 2.f / (__expf(2.f * acc[2][r]) + 1.f), go = 1.f / (1.f + __expf(-acc[3][r]));
                c[r] = gf * c[r] + gi * gg;
                h = go * (1.f - 2.f / (__expf(2.f * c[r]) + 1.f));
                if (MODE & 16) {
                    float* o = out + ((size_t)(blockIdx.x * 16 + 4 * kq + r) * steps + t) * 896 + 16 * wv + j;
                    o[0] = gi; o[128] = gf; o[256] = gg; o[384] = go; o[512] = c[r]; o[640] = h; o[768] = h;
                }
            } else {
                c[r] = c[r] * 0.5f + acc[0][r] + acc[1][r] + acc[2][r] + acc[3][r];
                h = c[r];
            }
            if (MODE & 8) hbuf[cur ^ 1][(4 * kq + r) * S + (16 * wv + j) / 32 * SEG + (16 * wv + j) % 32] = h;
        }
        if (MODE & 2) { cur ^= 1; __syncthreads(); }
    }
    if (!(MODE & 16)) out[blockIdx.x * 512 + threadIdx.x] = c[0] + c[1] + c[2] + c[3];
}

template <int MODE>
void run(const char* name, const float* w, float* out, int steps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE><<<256, 512>>>(w, out, steps);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int i = 0; i < 5; ++i) {
        hipEventRecord(a);
        probe<MODE><<<256, 512>>>(w, out, steps);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    const double mfma = 256.0 * 8 * steps * 136;
    printf("%-44s %.3f ms  %.2f us/step  %.1f TFLOP/s\n", name, best, best * 1e3 / steps, mfma * 2048 / (best * 1e-3) / 1e12);
}

int main() {
    const int steps = 128;
    float *w, *out;
    hipMalloc(&w, 4 * KS * 512 * 4);
    hipMalloc(&out, (size_t)4096 * steps * 896 * 4);
    std::vector<float> hw(4 * KS * 512);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    run<0>("mfma only (B regs, A regs)", w, out, steps);
    run<1>("+ LDS A-fragment reads", w, out, steps);
    run<1 | 2>("+ barrier", w, out, steps);
    run<1 | 2 | 8>("+ LDS h write", w, out, steps);
    run<1 | 2 | 4 | 8>("+ gate pointwise (exp/rcp)", w, out, steps);
    run<1 | 2 | 4 | 8 | 16>("+ 28 global stores / lane / step", w, out, steps);
    run<4>("mfma + pointwise, no LDS/barrier", w, out, steps);
    return 0;
}

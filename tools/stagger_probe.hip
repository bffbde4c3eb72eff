// stagger_probe.hip -- does the A/B half-step stagger hide the gate pointwise?  (diagnostic)
// GROUP: 0 = group B is waves >= 4, 1 = group B is odd waves, 2 = waves {2,3,6,7}
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int S = 168, SEG = 20, Q = 16, KH = 16;

template <int GROUP, bool STAG>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ w, float* __restrict__ out, int steps) {
    __shared__ __attribute__((aligned(16))) float hbuf[2][16 * S];
    const int lane = threadIdx.x & 63, j = lane & 15, kq = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool grp_b = GROUP == 0 ? (wv >= 4) : (GROUP == 1 ? (wv & 1) : ((wv >> 1) & 1));
    float wh[4][32];
    for (int q = 0; q < 4; ++q)
        for (int s = 0; s < 32; ++s) wh[q][s] = w[(q * 32 + s) * 512 + threadIdx.x];
    for (int i = threadIdx.x; i < 2 * 16 * S; i += 512) (&hbuf[0][0])[i] = 0.001f * i;
    __syncthreads();
    float c[4] = {0, 0, 0, 0};
    f32x4 acc[4];
    const int u = 16 * wv + j;
    auto start = [&]() {
        for (int q = 0; q < 4; ++q) {
            acc[q] = f32x4{0.1f, 0.2f, 0.3f, 0.4f};
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(c[0], wh[q][0], acc[q], 0, 0, 0);
            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(c[1], wh[q][1], acc[q], 0, 0, 0);
        }
    };
    auto half = [&](int hf, const float* hb) {
        const float* hrow = hb + j * S + (hf * 4 + kq) * SEG;
#pragma unroll
        for (int s = 0; s < Q; s += 4) {
            const float4 a = *reinterpret_cast<const float4*>(hrow + s);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wh[q][hf * KH + s], acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wh[q][hf * KH + s + 1], acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wh[q][hf * KH + s + 2], acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wh[q][hf * KH + s + 3], acc[q], 0, 0, 0);
        }
    };
    auto pw = [&](float* hnext) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float gi = __builtin_amdgcn_rcpf(1.f + __expf(-acc[0][r])), gf = __builtin_amdgcn_rcpf(1.f + __expf(-acc[1][r]));
            const float gg = 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * acc[2][r]) + 1.f), go = __builtin_amdgcn_rcpf(1.f + __expf(-acc[3][r]));
            c[r] = gf * c[r] + gi * gg;
            const float h = go * (1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * c[r]) + 1.f));
            hnext[(4 * kq + r) * S + (u / 16) * SEG + (u % 16)] = h;
        }
    };
    if (!STAG) {
        for (int t = 0; t < steps; ++t) {
            start(); half(0, hbuf[t & 1]); half(1, hbuf[t & 1]);
            pw(hbuf[(t + 1) & 1]);
            __syncthreads();
        }
    } else if (!grp_b) {
        start(); half(0, hbuf[0]);
        for (int t = 0; t < steps; ++t) {
            __syncthreads();
            half(1, hbuf[t & 1]);
            pw(hbuf[(t + 1) & 1]);
            __syncthreads();
            if (t + 1 < steps) { start(); half(0, hbuf[(t + 1) & 1]); }
        }
    } else {
        for (int t = 0; t < steps; ++t) {
            if (t > 0) pw(hbuf[t & 1]);
            __syncthreads();
            start(); half(0, hbuf[t & 1]); half(1, hbuf[t & 1]);
            __syncthreads();
        }
        pw(hbuf[steps & 1]);
    }
    out[blockIdx.x * 512 + threadIdx.x] = c[0] + c[1] + c[2] + c[3];
}

template <int GROUP, bool STAG>
void run(const char* name, const float* w, float* out, int steps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    probe<GROUP, STAG><<<256, 512>>>(w, out, steps);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int i = 0; i < 5; ++i) {
        hipEventRecord(a);
        probe<GROUP, STAG><<<256, 512>>>(w, out, steps);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    printf("%-40s %.3f ms  %.2f us/step  %.1f TFLOP/s\n", name, best, best * 1e3 / steps, 256.0 * 8 * steps * 136 * 2048 / (best * 1e-3) / 1e12);
}
int main() {
    const int steps = 128;
    float *w, *out;
    hipMalloc(&w, 4 * 32 * 512 * 4);
    hipMalloc(&out, 256 * 512 * 4);
    std::vector<float> hw(4 * 32 * 512);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    run<0, false>("lockstep", w, out, steps);
    run<0, true>("stagger, B = waves >= 4", w, out, steps);
    run<1, true>("stagger, B = odd waves", w, out, steps);
    run<2, true>("stagger, B = waves {2,3,6,7}", w, out, steps);
    return 0;
}

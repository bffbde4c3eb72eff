// mfma_calib.hip -- calibration of the SQ_VALU_MFMA_BUSY_CYCLES normalisation (diagnostic, not shipped):
// 1024 workgroups x 256 threads (one wave per SIMD, 4 workgroups per CU queued), each wave issues ITER x 8 independent
// v_mfma_f32_16x16x32_f16 back to back (8 accumulators: no RAW stall).  A saturated matrix pipe: the counter-derived
// mfma_busy_frac of this kernel is what "1.0" looks like, and flops / duration is the fp16 pipe's sustained dense rate.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_calib.hip -o tools/bin/mfma_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int DUTY>   // DUTY = 1: MFMA only; 2: every other slot is 16 cycles of VALU work (about half busy)
__global__ __launch_bounds__(256) void calib(float* out, int iters) {
    h8 a, b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.001f * (threadIdx.x + i));
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int i = 0; i < 8; ++i) b[q][i] = (_Float16)(0.002f * (threadIdx.x - i) + 0.01f * q);      // distinct operands: eight independent chains
    f4 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = f4{0.1f * q, 0.f, 0.f, 0.f};
    float v = threadIdx.x;
    for (int t = 0; t < iters; ++t) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            // inline asm with VGPR accumulators: the builtin form made the register allocator rotate misaligned AGPR tuples
            // through v_accvgpr copies on every trip (half the issue rate)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(a), "v"(b[q]));
            if (DUTY == 2) {
                asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n"
                             "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0" : "+v"(v));
            }
        }
    }
    float s = v;
    for (int q = 0; q < 8; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 1024 * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int duty = 1; duty <= 2; ++duty) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (duty == 1) calib<1><<<1024, 256>>>(out, iters); else calib<2><<<1024, 256>>>(out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flops = 1024.0 * 4 * iters * 8 * 2.0 * 16 * 16 * 32;
            printf("duty %d: %.3f ms, %.1f TFLOP/s dense fp16 (f32 accumulate)\n", duty, ms, flops / ms * 1e-9);
        }
    }
    return 0;
}

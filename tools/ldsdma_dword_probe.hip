// What does global_load_lds_dword (saddr form, per-lane 32-bit offset) leave in LDS?  Expected: lane L's dword at M0 + 4 L.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* src, float* out) {
    __shared__ __attribute__((aligned(16))) float buf[256];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 256; i += 64) buf[i] = -1.f;
    __syncthreads();
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) float*)buf;
    const unsigned voff = 4u * (lane & 31);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(voff), "s"(src), "s"(lds0 + 256u) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = buf[i];
}
int main() {
    float h[256], *d, *o;
    for (int i = 0; i < 256; ++i) h[i] = 100.f + i;
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof h);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
    for (int i = 56; i < 136; ++i) printf("%g%c", h[i], (i % 16 == 15) ? '\n' : ' ');
    return 0;
}

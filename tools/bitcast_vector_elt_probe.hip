// bitcast_vector_elt_probe.hip -- reproducer of a hipcc (ROCm 7.2.0, clang 20) miscompile found in round 5 while chasing the
// "DPP row_ror:8 of these MFMA results read wrong values" note of the h = 256 cluster kernels:
//
//     __builtin_bit_cast(T, v[r])   with v an ext_vector_type value and r a loop index
//
// reads ELEMENT 0 for every r.  The operand is a vector-element lvalue; the front end emits the cast from the vector's
// base address.  It has nothing to do with DPP or with MFMA hazards: the rotation was applied to element 0 four times
// (ISA of the cluster kernel with the builtin applied directly: ONE v_mov_b32_dpp per four uses; through a by-value
// helper -- float t = v[r]; bit_cast(t) -- four).  Rule for this code base: never bit_cast a subscripted ext-vector
// directly; go through a by-value temporary or a helper taking the scalar (csrc/mlp_fused.hip: row16_sum(float) does).
//
//   hipcc -O3 --offload-arch=gfx950 -emit-llvm -S --cuda-device-only tools/bitcast_vector_elt_probe.hip -o - | grep extractelement
//     direct:  ONE  `extractelement <4 x i32> %v, i64 0`     (wrong)
//     viatmp:  FOUR extractelements, indices 0..3            (right)
//   hipcc -O3 --offload-arch=gfx950 tools/bitcast_vector_elt_probe.hip -o /tmp/bvp && /tmp/bvp     (on a GPU: prints the mismatch)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void direct(const f32x4* in, int* out) {
    const int lane = threadIdx.x;
    const f32x4 vo = in[lane];
    for (int r = 0; r < 4; ++r) out[r * 64 + lane] = __builtin_bit_cast(int, vo[r]);
}
__global__ void viatmp(const f32x4* in, int* out) {
    const int lane = threadIdx.x;
    const f32x4 vo = in[lane];
    for (int r = 0; r < 4; ++r) { const float t = vo[r]; out[r * 64 + lane] = __builtin_bit_cast(int, t); }
}
int main() {
    float h[256];
    for (int i = 0; i < 256; ++i) h[i] = 1.0f + i;
    f32x4* din; int* dout;
    hipMalloc(&din, sizeof(h)); hipMalloc(&dout, 2 * 256 * sizeof(int));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    direct<<<1, 64>>>(din, dout); viatmp<<<1, 64>>>(din, dout + 256);
    int o[512]; hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += o[i] != o[256 + i];
    printf("bit_cast of a subscripted ext-vector element: %d of 256 values differ from the by-value form (0 = toolchain fixed)\n", bad);
    return 0;
}

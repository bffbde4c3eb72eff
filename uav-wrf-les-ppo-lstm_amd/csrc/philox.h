// philox.h -- Philox4x32-10 counter RNG (Salmon et al., SC'11), keyed by (seed), counter =
// (stream id, index, purpose).  Stateless: any (env, episode, cell) or (env, step) value can be
// regenerated anywhere, which is what makes the procedural plume field O(1) memory per env.
#pragma once
#include <stdint.h>

struct Philox4 { uint32_t x, y, z, w; };

__host__ __device__ inline uint32_t mulhi32(uint32_t a, uint32_t b) {
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
}

__host__ __device__ inline Philox4 philox4x32_10(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2,
                                                 uint32_t c3) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    Philox4 c{c0, c1, c2, c3};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = mulhi32(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = mulhi32(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = Philox4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// purposes (c3)
enum : uint32_t { RNG_SOURCE = 1, RNG_FIELD = 2, RNG_STEP = 3, RNG_ACTION = 4 };

// 53-bit uniform in [0,1) from two words
__host__ __device__ inline double u01_f64(uint32_t a, uint32_t b) {
    const uint64_t v = (((uint64_t)a << 32) | b) >> 11;
    return (double)v * (1.0 / 9007199254740992.0);
}
// uniform in (0,1] for log()
__host__ __device__ inline double u01_open_f64(uint32_t a, uint32_t b) {
    const uint64_t v = (((uint64_t)a << 32) | b) >> 11;
    return ((double)v + 1.0) * (1.0 / 9007199254740992.0);
}
// 24-bit uniform in [0,1) as f32 (what torch.multinomial's uniform provides)
__host__ __device__ inline float u01_f32(uint32_t a) { return (float)(a >> 8) * (1.0f / 16777216.0f); }

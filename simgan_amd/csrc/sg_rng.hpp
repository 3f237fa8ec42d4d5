// sg_rng.hpp -- counter-based random numbers for the library's own (non-injected) stochastic
// inputs: minibatch permutations (a2c/storage.py:159-162, DataLoader shuffle
// a2c/main_gail_dyn_ppo.py:171), the mixup alpha (a2c/algo/gail.py:72) and action noise
// (a2c/model.py:96).  Stateless: value = f(seed, stream, counter), so queued launches need no
// generator state and a run is reproducible from its seed.  (Parity tests inject the
// reference's own draws instead; torch's Philox/MT streams are not reproduced.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__host__ __device__ __forceinline__ uint64_t sg_mix64(uint64_t x) {  // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__host__ __device__ __forceinline__ uint64_t sg_key(uint64_t seed, uint64_t stream, uint64_t ctr) {
    return sg_mix64(sg_mix64(seed ^ (stream * 0xD6E8FEB86659FD93ull)) + ctr);
}

// uniform in [0, 1) with 24 random bits (the resolution torch.rand gives float32)
__host__ __device__ __forceinline__ float sg_uniform(uint64_t seed, uint64_t stream, uint64_t ctr) {
    return (float)(sg_key(seed, stream, ctr) >> 40) * (1.0f / 16777216.0f);
}

// one standard normal via Box-Muller (two uniforms from one 64-bit key)
__device__ __forceinline__ float sg_normal(uint64_t seed, uint64_t stream, uint64_t ctr) {
    const uint64_t k = sg_key(seed, stream, ctr);
    const float u1 = ((float)(k >> 40) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
    const float u2 = (float)((k >> 16) & 0xFFFFFF) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// Pseudo-random permutation of [0, n) evaluated per index: balanced Feistel network on
// 2*half_bits bits (2^(2*half_bits) >= n) with cycle walking.  Bijective by construction.
__host__ __device__ __forceinline__ uint32_t sg_feistel(uint32_t x, int half_bits, uint64_t key) {
    const uint32_t mask = (1u << half_bits) - 1u;
    uint32_t l = x >> half_bits, r = x & mask;
#pragma unroll
    for (int round = 0; round < 6; ++round) {
        const uint32_t f = (uint32_t)(sg_mix64(key + (uint64_t)round * 0x632BE59BD9B4E019ull + r) >> 32) & mask;
        const uint32_t nl = r;
        r = l ^ f;
        l = nl;
    }
    return (l << half_bits) | r;
}

__host__ __device__ __forceinline__ int sg_perm_half_bits(uint64_t n) {
    int hb = 1;
    while ((1ull << (2 * hb)) < n) ++hb;
    return hb;
}

__host__ __device__ __forceinline__ int64_t sg_perm_at(int64_t i, int64_t n, int half_bits, uint64_t key) {
    uint32_t x = (uint32_t)i;
    do { x = sg_feistel(x, half_bits, key); } while ((int64_t)x >= n);
    return (int64_t)x;
}

// In-kernel N(0,1) draws shared by the kernels that consume the iteration's randn_like tensors (pointwise.hip,
// edge_conv.hip).  Include inside an anonymous namespace user (the helpers are static inline).
#pragma once
#include "common.hpp"

namespace {

// ---- in-kernel N(0,1) draws: the three randn_like of an iteration (vaegan_code.py:77,91,92) -------------------
// Counter-based Philox4x32-10 (Salmon et al., SC'11): key = 64-bit seed, counter = (element index, draw id, step).
// state[0] = seed, state[1] = iteration counter, both in device memory: a replayed hipGraph reads the counter the
// one-thread vg_rng_advance kernel bumped at the top of the iteration, so every replay draws fresh noise, and the
// backward of the reparameterisation regenerates exactly the eps its forward used.  Box-Muller on two of the four
// output words.  A NoiseSrc with eps != NULL reads injected noise instead (parity runs).
struct NoiseSrc {
    const float* eps;
    const unsigned long long* rng;
    uint32_t draw;
};

__device__ __forceinline__ float philox_randn(unsigned long long seed, unsigned long long step, uint32_t draw,
                                              unsigned long long idx) {
    uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32), c2 = (uint32_t)step, c3 = ((uint32_t)(step >> 32) << 8) | draw;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const float u1 = ((float)c0 + 0.5f) * 2.3283064365386963e-10f;            // (0, 1]: 2^-32 * (x + 0.5), never 0
    const float u2 = ((float)c1 + 0.5f) * 2.3283064365386963e-10f;
    return sqrtf(-2.f * __logf(fmaxf(u1, 1e-30f))) * __cosf(6.283185307179586f * u2);
}

__device__ __forceinline__ float noise_at(const NoiseSrc& n, int64_t idx) {
    if (n.eps) return n.eps[idx];
    return philox_randn(n.rng[0], n.rng[1], n.draw, (unsigned long long)idx);
}


}  // namespace

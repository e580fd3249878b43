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

struct Normal4 { float v[4]; };

__device__ __forceinline__ void philox4x32_10(unsigned long long seed, unsigned long long step, uint32_t draw,
                                              unsigned long long blk, uint32_t (&w)[4]) {
    uint32_t c0 = (uint32_t)blk, c1 = (uint32_t)(blk >> 32), c2 = (uint32_t)step, c3 = ((uint32_t)(step >> 32) << 8) | draw;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    w[0] = c0; w[1] = c1; w[2] = c2; w[3] = c3;
}
// Box-Muller radius and angle from two words: u = 2^-32 * (x + 0.5) in (0, 1], never 0
__device__ __forceinline__ void bm_polar(uint32_t wa, uint32_t wb, float& r, float& t) {
    const float ua = ((float)wa + 0.5f) * 2.3283064365386963e-10f;
    const float ub = ((float)wb + 0.5f) * 2.3283064365386963e-10f;
    r = sqrtf(-2.f * __logf(fmaxf(ua, 1e-30f)));
    t = 6.283185307179586f * ub;
}

// One Philox4x32-10 block = FOUR normals (two Box-Muller pairs, cosine and sine of each): element idx of a draw is
// component idx & 3 of block idx >> 2.  A kernel that owns four consecutive elements (idx % 4 == 0) pays for one block;
// the one-element form evaluates the same expressions for its component only (bit-identical values).
__device__ __forceinline__ Normal4 philox_randn4(unsigned long long seed, unsigned long long step, uint32_t draw,
                                                 unsigned long long blk) {
    uint32_t w[4];
    philox4x32_10(seed, step, draw, blk, w);
    float r1, t1, r2, t2;
    bm_polar(w[0], w[1], r1, t1);
    bm_polar(w[2], w[3], r2, t2);
    Normal4 n;
    n.v[0] = r1 * __cosf(t1); n.v[1] = r1 * __sinf(t1);
    n.v[2] = r2 * __cosf(t2); n.v[3] = r2 * __sinf(t2);
    return n;
}

__device__ __forceinline__ float philox_randn(unsigned long long seed, unsigned long long step, uint32_t draw,
                                              unsigned long long idx) {
    uint32_t w[4];
    philox4x32_10(seed, step, draw, idx >> 2, w);
    const int k = (int)(idx & 3);
    float r, t;
    bm_polar(k < 2 ? w[0] : w[2], k < 2 ? w[1] : w[3], r, t);
    return (k & 1) ? r * __sinf(t) : r * __cosf(t);
}

__device__ __forceinline__ float noise_at(const NoiseSrc& n, int64_t idx) {
    if (n.eps) return n.eps[idx];
    return philox_randn(n.rng[0], n.rng[1], n.draw, (unsigned long long)idx);
}

// four consecutive elements idx .. idx + 3, idx % 4 == 0 (injected noise: eps + idx 16-byte aligned by contract of the caller)
__device__ __forceinline__ Normal4 noise4_at(const NoiseSrc& n, int64_t idx) {
    if (n.eps) {
        const float4 e = *reinterpret_cast<const float4*>(n.eps + idx);
        return Normal4{{e.x, e.y, e.z, e.w}};
    }
    return philox_randn4(n.rng[0], n.rng[1], n.draw, (unsigned long long)idx >> 2);
}

}  // namespace

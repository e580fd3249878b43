// Narrow-K direct convolution: the image-side layers whose INPUT has 3 (padded 8) channels, i.e. one pixel = one
// 16-byte unit = one 8-element MFMA k-group -- the Discriminator's first Conv2d(3 -> C, k4 s2 p1) (gan_code.py:61),
// the Encoder's first Conv2d(3 -> 32, k4 s2 p0) (main_vae.py:23) and the data gradient of the Generator's last
// ConvTranspose2d(C -> 3, k3 s1 p1) (gan_code.py:49), a 3x3 convolution of the 3-channel image gradient.
// Included by conv_gemm.hip (inside its anonymous namespace); same descriptor (vg_gg_desc), same packed weights.
//
// These layers are HBM-bound (<= 20 FLOP/B): K = taps x 8 is 72..128, the output is N = 32..64
// channels wide.  The generic gather-GEMM stages A through LDS with per-16-byte bounds tests and an LDS-staged C tile
// sized for 128-wide problems; here
//   * the whole weight operand lives in registers (N/16 x K/32 fragments, <= 64 VGPRs),
//   * an A fragment IS a gather of 16 pixels x 4 taps (lane = (pixel, tap)): loaded straight from L1/L2 into VGPRs,
//     16 independent loads in flight per lane, no LDS, no barrier in the main part,
//   * a workgroup owns 256 consecutive output pixels x all N channels: its C tile is one contiguous run of the NHWC
//     output (256 x N x 2 bytes), written with 16-byte stores after a transpose through LDS,
//   * bias, (Leaky)ReLU of a BatchNorm-less layer and the per-channel BatchNorm partial sums ride in the epilogue,
//     one slab row per workgroup exactly as gg_kernel emits them (bn_act.hip reads either).

#ifndef VG_NK_GP
#define VG_NK_GP 4
#endif
constexpr int NK_GP = VG_NK_GP;        // 16-pixel MFMA row groups per wave (4: 204 VGPRs at N=64, K=128 -> 2 waves per SIMD)
constexpr int NK_BM = 64 * NK_GP;

inline bool narrowk_enabled() { return vg_sw().edge != 0; }      // VG_EDGE (common.hpp: switches are read once at load)

// host: does the descriptor have the narrow-K form?
inline bool narrowk_ok(const vg_gg_desc* d, int dtype) {
    if (dtype != VG_BF16 || !narrowk_enabled()) return false;
    if (d->IC != 8 || d->nphase != 1 || d->mask_x != nullptr) return false;
    if (d->N % 16 != 0 || d->N < 16 || d->N > 64 || d->OC != d->N) return false;
    if (d->Kp % 32 != 0 || d->Kp > 128 || d->Kp < d->TH * d->TW * 8 || d->TW > 4) return false;
    if ((int64_t)d->B * d->GH * d->GW >= (1 << 24)) return false;                             // float-reciprocal divisions
    if (!(d->OSY == 1 && d->OSX == 1 && d->GH == d->OH && d->GW == d->OW)) return false;     // flat output
    return true;
}

__device__ __forceinline__ int nk_fdiv(int a, int b, float inv) {      // a / b for 0 <= a < 2^24, inv = 1.0f / b
    int q = (int)((float)a * inv);
    const int r = a - q * b;
    if (r < 0) --q;
    else if (r >= b) ++q;
    return q;
}

template <int NT, int KC>                       // N = 16 * NT output channels, Kp = 32 * KC
__global__ __launch_bounds__(256) void ggn_kernel(const vg_gg_desc d) {
    // per-wave C staging [64 pixels][N] bf16 (+16 B pad per pixel row) | stats scratch [4 waves][N][2]
    constexpr int N = NT * 16, CP = N * 2 + 16;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 16 * NK_GP * CP + 4 * N * 2 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int M = d.B * d.GH * d.GW, GHW = d.GH * d.GW;
    const int m0 = blockIdx.x * NK_BM + wave * 16 * NK_GP;
    const int ntap = d.TH * d.TW;
    // no integer divisions in the address generation (they, not the loads, were the critical path of the first cut):
    // m -> (b, gy, gx) through float reciprocals (m < 2^24), tap -> (ta, tb) through an 8-bit fixed-point reciprocal
    const float inv_ghw = 1.0f / (float)GHW, inv_gw = 1.0f / (float)d.GW;
    const int tw_inv = (256 + d.TW - 1) / d.TW;               // exact for taps < 16, TW <= 4

    constexpr int SEGS = NT * 16 * 2 / 16;                                    // 16-byte segments per pixel

    // ---- weights: all of them, in registers ----
    bf16x8 bw[NT][KC];
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(d.W);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
            bw[nt][kc] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
                Wb + ((int64_t)(nt * 16 + fr) * d.Kp + kc * 32 + fg * 8) * 2));

    // ---- A fragments: lane (fr, fg) = (pixel m0 + 16g + fr, tap 4kc + fg) -> one 16-byte input pixel ----
    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(d.X);
    u32x4 a[NK_GP][KC];
#pragma unroll
    for (int g = 0; g < NK_GP; ++g) {
        const int m = m0 + g * 16 + fr;
        int b = 0, gy = -(1 << 20), gx = 0;
        if (m < M) {
            b = nk_fdiv(m, GHW, inv_ghw);
            const int r = m - b * GHW;
            gy = nk_fdiv(r, d.GW, inv_gw);
            gx = r - gy * d.GW;
        }
        const int iy0 = gy * d.SY + d.y0[0], ix0 = gx * d.SX + d.x0[0];
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const int t = kc * 4 + fg;
            const int ta = (t * tw_inv) >> 8, tb = t - ta * d.TW;
            const int iy = iy0 + d.DY * ta, ix = ix0 + d.DX * tb;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (t < ntap && (unsigned)iy < (unsigned)d.IH && (unsigned)ix < (unsigned)d.IW)
                v = *reinterpret_cast<const u32x4*>(Xb + ((int64_t)(b * d.IH + iy) * d.IW + ix) * 16);
            a[g][kc] = v;
        }
    }
    f32x4 acc[NK_GP][NT];
#pragma unroll
    for (int g = 0; g < NK_GP; ++g)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[g][kc]), bw[nt][kc], c, 0, 0, 0);
            acc[g][nt] = c;
        }

    // ---- epilogue: lane holds rows (pixels) 16g + 4fg + r, column n = 16nt + fr ----
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float bv = d.bias != nullptr ? d.bias[nt * 16 + fr] : 0.f;
#pragma unroll
        for (int g = 0; g < NK_GP; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[g][nt][r] + bv;
                if (d.act != VG_ACT_NONE) v = act_fwd(v, d.act, d.act_slope);
                acc[g][nt][r] = v;
            }
    }
    if (d.stats != nullptr) {
        float* red = reinterpret_cast<float*>(smem + 4 * 16 * NK_GP * CP);  // [wave][N][2]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int g = 0; g < NK_GP; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (m0 + g * 16 + fg * 4 + r < M) {
                        const float v = acc[g][nt][r];
                        s1 += v;
                        s2 += v * v;
                    }
            s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
            if (fg == 0) {
                red[(wave * N + nt * 16 + fr) * 2 + 0] = s1;
                red[(wave * N + nt * 16 + fr) * 2 + 1] = s2;
            }
        }
        __syncthreads();
        if (tid < N) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s1 += red[(w * N + tid) * 2]; s2 += red[(w * N + tid) * 2 + 1]; }
            d.stats[((int64_t)blockIdx.x * 2 + 0) * d.N + tid] = s1;
            d.stats[((int64_t)blockIdx.x * 2 + 1) * d.N + tid] = s2;
        }
    }
    // C tile: each wave transposes its own 64 x N block through LDS and writes one contiguous run of the output
    unsigned char* cw = smem + wave * 16 * NK_GP * CP;
#pragma unroll
    for (int g = 0; g < NK_GP; ++g)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<uint16_t*>(cw + (g * 16 + fg * 4 + r) * CP + (nt * 16 + fr) * 2) =
                    ElemT<VG_BF16>::from_f32(acc[g][nt][r]);
    __syncthreads();
    unsigned char* Yb = reinterpret_cast<unsigned char*>(d.Y);
#pragma unroll
    for (int it = 0; it < 16 * NK_GP * SEGS / 64; ++it) {
        const int u = lane + 64 * it;
        const int row = u / SEGS, seg = u - row * SEGS;
        const int m = m0 + row;
        if (m < M) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(cw + row * CP + seg * 16);
            *reinterpret_cast<u32x4*>(Yb + (int64_t)m * (N * 2) + seg * 16) = v;
        }
    }
}

inline int launch_narrowk(const vg_gg_desc* d, hipStream_t s) {
    const int M = d->B * d->GH * d->GW;
    dim3 grid((M + NK_BM - 1) / NK_BM), block(256);
    const int NT = d->N / 16, KC = d->Kp / 32;
#define NK_LAUNCH(A, B) vg_launch_timed(2, (ggn_kernel<A, B>), grid, block, 0, s, *d)
    if (NT == 1) { if (KC == 1) NK_LAUNCH(1, 1); else if (KC == 2) NK_LAUNCH(1, 2); else if (KC == 3) NK_LAUNCH(1, 3); else NK_LAUNCH(1, 4); }
    else if (NT == 2) { if (KC == 1) NK_LAUNCH(2, 1); else if (KC == 2) NK_LAUNCH(2, 2); else if (KC == 3) NK_LAUNCH(2, 3); else NK_LAUNCH(2, 4); }
    else if (NT == 3) { if (KC == 1) NK_LAUNCH(3, 1); else if (KC == 2) NK_LAUNCH(3, 2); else if (KC == 3) NK_LAUNCH(3, 3); else NK_LAUNCH(3, 4); }
    else { if (KC == 1) NK_LAUNCH(4, 1); else if (KC == 2) NK_LAUNCH(4, 2); else if (KC == 3) NK_LAUNCH(4, 3); else NK_LAUNCH(4, 4); }
#undef NK_LAUNCH
    return VG_LAUNCH_RC();
}

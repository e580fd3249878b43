// Edge layers: the convolutions at the image boundary of the networks, where one side has 3 (padded 8) channels --
// the Generator's last ConvTranspose2d(C -> 3, k3 s1 p1) + Tanh (gan_code.py:49-50) and the image gradient below the
// Discriminator's first Conv2d(3 -> C, k4 s2 p1) (gan_code.py:61).  SURVEY.md section 8(d) prices them against the
// HBM roofline (<= 20 FLOP/B): the MFMA work is negligible, what matters is that every activation byte moves once.
//
// vg_tnconv -- narrow-N transposed convolution as GEMM + col2im:
//     Y[b][oy][ox][n] = sum_{c,kh,kw} X[b][iy][ix][c] * W[c][n][kh][kw],   oy = iy*S - P + kh,  ox = ix*S - P + kw
// The gather-GEMM form (conv_gemm.hip) contracts over (tap, c) and re-reads every input pixel once per tap (9x / 4x
// here, through LDS-DMA).  With N*K*K <= 64 the other factorisation is cheaper by an order of magnitude in on-chip
// traffic: ONE GEMM per input pixel over the channels only,
//     Pm[pix][j] = sum_c X[pix][c] * Wp[j][c],      j = (kh*K + kw)*N + n      (K*K*N <= 64 columns)
// with the A fragments loaded straight from HBM in whole pixel rows (each input byte is read once, plus the halo
// rows shared with the neighbouring tile), the f32 product tile parked in LDS, and the K*K-tap sum done there:
//     Y[oy][ox][n] = sum_{kh,kw} Pm[(oy + P - kh)/S][(ox + P - kw)/S][(kh,kw,n)]        (divisible, in range)
// Epilogue options: tanh -> NCHW f32 image (what vaegan_code.py:83 returns to the loss code) and, in the same pass,
// image + sigma*noise in the Discriminator's NHWC layout (vaegan_code.py:92), noise injected or drawn in-kernel.
#include "common.hpp"
#include "noise.hpp"

namespace {

// a / b for 0 <= a < 2^24 with inv = 1.0f / b: no integer division in the pixel loops
__device__ __forceinline__ int fdiv(int a, int b, float inv) {
    int q = (int)((float)a * inv);
    const int r = a - q * b;
    if (r < 0) --q;
    else if (r >= b) ++q;
    return q;
}

// NT 16-column tiles of j (K*K*N <= 16*NT), KC 32-channel chunks (C = 32*KC), NPIX input pixels (halo rows included)
// per workgroup, K x K taps at stride S (compile time: the col2im loop unrolls to its 9 | 4 live taps, no divisions)
template <int NT, int KC, int NPIX, int K, int S>
__global__ __launch_bounds__(256) void tnconv_kernel(const vg_tn_desc d, const int RO, const int tiles_y) {
    constexpr int PS = NPIX + 4;                              // LDS row pitch of the product tile (floats)
    __shared__ __attribute__((aligned(16))) float Pm[NT * 16 * PS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.x / tiles_y, ty = blockIdx.x - b * tiles_y;
    const int oy_a = ty * RO, oy_b = min(d.OH, oy_a + RO);
    // input rows that contribute to output rows [oy_a, oy_b)
    int iy_lo = oy_a + d.P - (K - 1);
    iy_lo = iy_lo <= 0 ? 0 : (iy_lo + S - 1) / S;
    int iy_hi = (oy_b - 1 + d.P) / S;                         // inclusive
    iy_hi = min(iy_hi, d.IH - 1);
    const int nrows = iy_hi - iy_lo + 1;
    const int npix = nrows * d.IW;                            // <= NPIX (host), multiple of 16 (IW % 16 == 0)
    const int NJ = K * K * d.N;

    // ---- B fragments: Wp[j][c] (bf16, row pitch Wpitch), zero rows beyond NJ ----
    bf16x8 bw[NT][KC];
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(d.Wp);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const int j = nt * 16 + fr;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (j < NJ) v = *reinterpret_cast<const u32x4*>(Wb + ((int64_t)j * d.Wpitch + kc * 32 + fg * 8) * 2);
            bw[nt][kc] = __builtin_bit_cast(bf16x8, v);
        }

    // ---- GEMM over the tile's pixels: 16 pixels per MFMA row group; ALL of a wave's groups (<= 8: 16 x 16-byte
    //      loads per lane) are in flight at once -- the only latency the workgroup exposes is one HBM round trip ----
    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(d.X) +
                              ((int64_t)(b * d.IH + iy_lo) * d.IW) * (int64_t)(KC * 64);
    const int ngroups = npix >> 4;
    constexpr int U = NPIX / 64;                              // groups per wave
    {
        u32x4 a[U][KC];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const int g = wave * U + u;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (g < ngroups) v = *reinterpret_cast<const u32x4*>(Xb + (int64_t)(g * 16 + fr) * (KC * 64) + kc * 64 + fg * 16);
                a[u][kc] = v;
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int g = wave * U + u;
            if (g < ngroups) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kc = 0; kc < KC; ++kc)
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[u][kc]), bw[nt][kc], acc, 0, 0, 0);
                    // lane holds Pm[pixel 16g + 4fg + r][j = 16nt + fr], r = 0..3: four consecutive pixels of one column
                    *reinterpret_cast<f32x4*>(&Pm[(nt * 16 + fr) * PS + g * 16 + fg * 4]) = acc;
                }
            }
        }
    }
    __syncthreads();

    // ---- col2im + epilogue: one output pixel per thread and pass, taps summed in fixed (kh, kw) order ----
    const int count = (oy_b - oy_a) * d.OW;
    const float inv_ow = 1.0f / (float)d.OW;
    const NoiseSrc nz{d.eps, reinterpret_cast<const unsigned long long*>(d.rng), (uint32_t)d.draw};
    const bool noisy = d.eps != nullptr || d.rng != nullptr;
    unsigned char* Yb = reinterpret_cast<unsigned char*>(d.Y);
    const int N = d.N;
    for (int q = tid; q < count; q += 256) {
        const int oyl = fdiv(q, d.OW, inv_ow);
        const int ox = q - oyl * d.OW, oy = oy_a + oyl;
        float v[4] = {0.f, 0.f, 0.f, 0.f};                    // N <= 4 output channels
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            const int t = oy + d.P - kh;
            const int iy = S == 2 ? (t >> 1) : t;
            const bool yok = t >= 0 && !(S == 2 && (t & 1)) && iy <= iy_hi;
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const int s = ox + d.P - kw;
                const int ix = S == 2 ? (s >> 1) : s;
                if (yok && s >= 0 && !(S == 2 && (s & 1)) && ix < d.IW) {
                    const float* src = &Pm[((kh * K + kw) * N) * PS + (iy - iy_lo) * d.IW + ix];
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        if (n < N) v[n] += src[n * PS];
                }
            }
        }
        const int64_t opix = ((int64_t)b * d.OH + oy) * d.OW + ox;
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (n < N) {
                float t = v[n];
                if (d.act == VG_ACT_TANH) t = tanhf(t);
                const int64_t cidx = (((int64_t)b * N + n) * d.OH + oy) * d.OW + ox;      // NCHW index (noise order too)
                if (d.Y_nchw) d.Y_nchw[cidx] = t;
                if (noisy) t = t + d.sigma * noise_at(nz, cidx);
                o[n] = t;
            }
        }
        if (Yb) {
            u32x4 pk;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                pk[k] = (uint32_t)ElemT<VG_BF16>::from_f32(o[2 * k]) | ((uint32_t)ElemT<VG_BF16>::from_f32(o[2 * k + 1]) << 16);
            *reinterpret_cast<u32x4*>(Yb + opix * 16) = pk;    // OC = 8 bf16 = one 16-byte pixel
        }
    }
}

struct TnPlan { int RO, tiles_y, NT, KC, NPIX; };

inline int tn_plan(const vg_tn_desc* d, TnPlan* p) {
    VG_CHECK_ARG(d != nullptr, VG_EINVAL);
    VG_CHECK_ARG(d->X && d->Wp && (d->Y || d->Y_nchw), VG_EINVAL);
    VG_CHECK_ARG(d->B > 0 && d->IH > 0 && d->IW > 0 && d->P >= 0, VG_EINVAL);
    VG_CHECK_ARG((d->K == 3 && d->S == 1) || (d->K == 4 && d->S == 2), VG_ENOSUP);      // the two edge-layer forms
    VG_CHECK_ARG(d->N >= 1 && d->N <= 4 && d->OC == 8, VG_ENOSUP);
    VG_CHECK_ARG(d->C == 32 || d->C == 64, VG_ENOSUP);
    VG_CHECK_ARG(d->Wpitch >= d->C && d->Wpitch % 8 == 0, VG_EALIGN);
    VG_CHECK_ARG(d->OH == (d->IH - 1) * d->S - 2 * d->P + d->K && d->OW == (d->IW - 1) * d->S - 2 * d->P + d->K, VG_EINVAL);
    VG_CHECK_ARG(vg_aligned16(d->X) && vg_aligned16(d->Wp) && vg_aligned16(d->Y), VG_EALIGN);
    VG_CHECK_ARG(d->act == VG_ACT_NONE || d->act == VG_ACT_TANH, VG_EINVAL);
    VG_CHECK_ARG(d->draw >= 0 && d->draw < 256, VG_EINVAL);
    const int NJ = d->K * d->K * d->N;
    VG_CHECK_ARG(NJ <= 64, VG_ENOSUP);
    p->NT = (NJ + 15) / 16;
    p->KC = d->C / 32;
    // product tile in LDS: NT*16 columns x NPIX pixels of f32; keep it <= ~66 KB so that two workgroups share a CU
    // (one's HBM loads under the other's col2im): 512 pixels for <= 32 columns, 256 beyond
    p->NPIX = p->NT <= 2 ? 512 : 256;
    VG_CHECK_ARG(d->IW % 16 == 0 && d->IW <= p->NPIX, VG_ENOSUP);
    // largest block of output rows (a multiple of S) whose input rows, halo included, fit the tile
    const int max_rows = p->NPIX / d->IW;
    int RO = 0;
    for (int r = d->S; r <= d->OH + d->S; r += d->S) {
        const int need = (r - 1 + d->K - 1) / d->S + 1;      // input rows touched by r consecutive output rows (worst phase)
        if (need > max_rows) break;
        RO = r;
    }
    VG_CHECK_ARG(RO > 0, VG_ENOSUP);
    if (RO > d->OH) RO = ((d->OH + d->S - 1) / d->S) * d->S;
    p->RO = RO;
    p->tiles_y = (d->OH + RO - 1) / RO;
    return 0;
}

template <int NT, int KC>
inline void tn_launch(const vg_tn_desc* d, const TnPlan& p, hipStream_t s) {
    dim3 grid(d->B * p.tiles_y), block(256);
    constexpr int NPIX = NT <= 2 ? 512 : 256;
    if (d->K == 3) vg_launch_timed(2, (tnconv_kernel<NT, KC, NPIX, 3, 1>), grid, block, 0, s, *d, p.RO, p.tiles_y);
    else vg_launch_timed(2, (tnconv_kernel<NT, KC, NPIX, 4, 2>), grid, block, 0, s, *d, p.RO, p.tiles_y);
}

}  // namespace

extern "C" int vg_tnconv_supported(const vg_tn_desc* d) {
    TnPlan p;
    return tn_plan(d, &p);
}

extern "C" int vg_tnconv(const vg_tn_desc* d, void* stream) {
    TnPlan p;
    int rc = tn_plan(d, &p);
    if (rc) return rc;
    hipStream_t s = vg_stream(stream);
    if (p.KC == 1) {
        if (p.NT == 1) tn_launch<1, 1>(d, p, s); else if (p.NT == 2) tn_launch<2, 1>(d, p, s);
        else if (p.NT == 3) tn_launch<3, 1>(d, p, s); else tn_launch<4, 1>(d, p, s);
    } else {
        if (p.NT == 1) tn_launch<1, 2>(d, p, s); else if (p.NT == 2) tn_launch<2, 2>(d, p, s);
        else if (p.NT == 3) tn_launch<3, 2>(d, p, s); else tn_launch<4, 2>(d, p, s);
    }
    return VG_LAUNCH_RC();
}

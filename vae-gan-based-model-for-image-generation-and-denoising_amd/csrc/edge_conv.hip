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

// component j of the Normal4 held by lane SRC of this lane's quad (quad-broadcast DPP moves: quad_perm [SRC, SRC, SRC, SRC])
template <int SRC>
__device__ __forceinline__ float quad_pick(const Normal4& blk, int j) {
    float c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        c[k] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(blk.v[k]), SRC * 0x55, 0xf, 0xf, false));
    return j == 0 ? c[0] : j == 1 ? c[1] : j == 2 ? c[2] : c[3];
}

// NT 16-column tiles of j (K*K*N <= 16*NT), KC 32-channel chunks (C = 32*KC), NPIX input pixels (halo rows included)
// per workgroup, K x K taps at stride S (compile time: the col2im loop unrolls to its 9 | 4 live taps, no divisions)
template <int NT, int KC, int NPIX, int K, int S>
__global__ __launch_bounds__(256) void tnconv_kernel(const vg_tn_desc d, const int RO, const int tiles_y, const int CO,
                                                     const int tiles_x, const int TC) {
    // Tile: output rows [oy_a, oy_b) x output columns [ox_a, ox_b); its input window is nrows x TC pixels (TC a multiple of
    // 16: whole MFMA row groups per window row) starting at column c0.  tiles_x == 1: TC = IW, whole rows (every map up to
    // 128 pixels wide); wider maps (S=256 members) are cut into column blocks with a K-1 halo on either side.
    constexpr int PS = NPIX + 4;                              // LDS row pitch of the product tile (floats)
    __shared__ __attribute__((aligned(16))) float Pm[NT * 16 * PS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int txy = tiles_x * tiles_y;
    const int b = blockIdx.x / txy, tr = blockIdx.x - b * txy;
    const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
    const int oy_a = ty * RO, oy_b = min(d.OH, oy_a + RO);
    const int ox_a = tx * CO, ox_b = min(d.OW, ox_a + CO);
    int c0 = ox_a + d.P - (K - 1);
    c0 = c0 <= 0 ? 0 : (c0 + S - 1) / S;                      // first input column that contributes
    // input rows that contribute to output rows [oy_a, oy_b)
    int iy_lo = oy_a + d.P - (K - 1);
    iy_lo = iy_lo <= 0 ? 0 : (iy_lo + S - 1) / S;
    int iy_hi = (oy_b - 1 + d.P) / S;                         // inclusive
    iy_hi = min(iy_hi, d.IH - 1);
    const int nrows = iy_hi - iy_lo + 1;
    const int npix = nrows * TC;                              // <= NPIX (host), multiple of 16 (TC % 16 == 0)
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
    // one pixel = C bf16 = pixb bytes (C = 16: half a 32-channel chunk -- the lanes of its upper half load nothing and
    // multiply zero weight rows: S >= 128 members of the size family, gan_code.py:46-49 / :61-66)
    const int pixb = d.C * 2;
    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(d.X) +
                              ((int64_t)(b * d.IH + iy_lo) * d.IW) * (int64_t)pixb;
    const int ngroups = npix >> 4;
    const float inv_tc = 1.0f / (float)TC;
    constexpr int U = NPIX / 64;                              // groups per wave
    {
        u32x4 a[U][KC];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const int g = wave * U + u;
                const int t = g * 16 + fr;                    // pixel of the window: row t / TC, column c0 + t % TC
                const int wr = fdiv(t, TC, inv_tc), gx = c0 + t - wr * TC;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (g < ngroups && gx < d.IW && kc * 32 + fg * 8 < d.C)
                    v = *reinterpret_cast<const u32x4*>(Xb + ((int64_t)wr * d.IW + gx) * pixb + kc * 64 + fg * 16);
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
    const int cw = ox_b - ox_a;                               // output columns of this tile
    const int count = (oy_b - oy_a) * cw;
    const float inv_cw = 1.0f / (float)cw;
    const int ix_end = min(d.IW, c0 + TC);                    // input columns [c0, ix_end) are in the window
    const NoiseSrc nz{d.eps, reinterpret_cast<const unsigned long long*>(d.rng), (uint32_t)d.draw};
    const bool noisy = d.eps != nullptr || d.rng != nullptr;
    // quads of lanes = four consecutive pixels of one row whose NCHW indices start at a multiple of 4 (every channel plane
    // and row is a multiple of 4 long, the tile's columns start and end on multiples of 4: count % 4 == 0 too, so a quad
    // is complete in every pass)
    const bool quad_noise = d.eps == nullptr && d.rng != nullptr && (cw & 3) == 0 && (ox_a & 3) == 0 && (d.OW & 3) == 0;
    unsigned char* Yb = reinterpret_cast<unsigned char*>(d.Y);
    const int N = d.N;
    for (int q = tid; q < count; q += 256) {
        const int oyl = fdiv(q, cw, inv_cw);
        const int ox = ox_a + q - oyl * cw, oy = oy_a + oyl;
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
                if (yok && s >= 0 && !(S == 2 && (s & 1)) && ix < ix_end) {
                    const float* src = &Pm[((kh * K + kw) * N) * PS + (iy - iy_lo) * TC + (ix - c0)];
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        if (n < N) v[n] += src[n * PS];
                }
            }
        }
        const int64_t opix = ((int64_t)b * d.OH + oy) * d.OW + ox;
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // In-kernel noise, shared by lane quads (round 4): the four lanes of a quad hold four consecutive pixels of one
        // output row, i.e. for each channel the four elements of ONE Philox block (element idx = component idx & 3 of
        // block idx >> 2).  Lane j < N of the quad generates channel j's block (four normals), quad-broadcast DPP moves hand
        // every lane its component: one Philox + two Box-Muller pairs per lane instead of N of each -- the same values.
        float nzv[4] = {0.f, 0.f, 0.f, 0.f};
        if (quad_noise) {
            const int j = lane & 3;
            const int jn = j < N ? j : 0;
            const int64_t cj = (((int64_t)b * N + jn) * d.OH + oy) * d.OW + (ox - j);      // first pixel of the quad, channel jn
            const Normal4 blk = philox_randn4(nz.rng[0], nz.rng[1], nz.draw, (unsigned long long)cj >> 2);
            // what lane j needs of channel n's block is component j: select it on the SOURCE side (every lane picks its
            // own v[j] ... no: the source lane n must supply v[j of the destination]) -> broadcast all four, pick locally
            nzv[0] = quad_pick<0>(blk, j);
            if (N > 1) nzv[1] = quad_pick<1>(blk, j);
            if (N > 2) nzv[2] = quad_pick<2>(blk, j);
            if (N > 3) nzv[3] = quad_pick<3>(blk, j);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (n < N) {
                float t = v[n];
                if (d.act == VG_ACT_TANH) t = tanhf(t);
                const int64_t cidx = (((int64_t)b * N + n) * d.OH + oy) * d.OW + ox;      // NCHW index (noise order too)
                if (d.Y_nchw) d.Y_nchw[cidx] = t;
                if (quad_noise) t = t + d.sigma * nzv[n];
                else if (noisy) t = t + d.sigma * noise_at(nz, cidx);
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

struct TnPlan { int RO, tiles_y, NT, KC, NPIX, CO, tiles_x, TC; };

inline int tn_plan(const vg_tn_desc* d, TnPlan* p) {
    VG_CHECK_ARG(d != nullptr, VG_EINVAL);
    VG_CHECK_ARG(d->X && d->Wp && (d->Y || d->Y_nchw), VG_EINVAL);
    VG_CHECK_ARG(d->B > 0 && d->IH > 0 && d->IW > 0 && d->P >= 0, VG_EINVAL);
    VG_CHECK_ARG((d->K == 3 && d->S == 1) || (d->K == 4 && d->S == 2), VG_ENOSUP);      // the two edge-layer forms
    VG_CHECK_ARG(d->N >= 1 && d->N <= 4 && d->OC == 8, VG_ENOSUP);
    VG_CHECK_ARG(d->C == 16 || d->C == 32 || d->C == 64, VG_ENOSUP);
    VG_CHECK_ARG(d->Wpitch >= ((d->C + 31) / 32) * 32 && d->Wpitch % 8 == 0, VG_EALIGN);     // whole 32-channel chunks (zero-padded)
    VG_CHECK_ARG(d->OH == (d->IH - 1) * d->S - 2 * d->P + d->K && d->OW == (d->IW - 1) * d->S - 2 * d->P + d->K, VG_EINVAL);
    VG_CHECK_ARG(vg_aligned16(d->X) && vg_aligned16(d->Wp) && vg_aligned16(d->Y), VG_EALIGN);
    VG_CHECK_ARG(d->act == VG_ACT_NONE || d->act == VG_ACT_TANH, VG_EINVAL);
    VG_CHECK_ARG(d->draw >= 0 && d->draw < 256, VG_EINVAL);
    const int NJ = d->K * d->K * d->N;
    VG_CHECK_ARG(NJ <= 64, VG_ENOSUP);
    p->NT = (NJ + 15) / 16;
    p->KC = (d->C + 31) / 32;
    // product tile in LDS: NT*16 columns x NPIX pixels of f32; keep it <= ~66 KB so that two workgroups share a CU
    // (one's HBM loads under the other's col2im): 512 pixels for <= 32 columns, 256 beyond
    p->NPIX = p->NT <= 2 ? 512 : 256;
    VG_CHECK_ARG(d->IW % 16 == 0, VG_ENOSUP);
    // Window of input pixels per workgroup: nrows x TC <= NPIX with TC a multiple of 16.  Whole rows (TC = IW) when at
    // least the rows behind one block of S output rows fit; otherwise (maps wider than ~128 pixels) column blocks.  Among
    // the candidates the one with the least halo overhead (window pixels per output pixel) wins.
    auto rows_for = [&](int tc) { return p->NPIX / tc; };
    auto ro_for = [&](int max_rows) {               // largest block of output rows (a multiple of S) whose input rows fit
        int RO = 0;
        for (int r = d->S; r <= d->OH + d->S; r += d->S) {
            const int need = (r - 1 + d->K - 1) / d->S + 1;  // input rows touched by r consecutive output rows (worst phase)
            if (need > max_rows) break;
            RO = r;
        }
        return RO;
    };
    auto co_for = [&](int tc) {                     // largest block of output columns (a multiple of S) behind tc input columns
        // (with in-kernel noise: a multiple of 4, so that lane quads own whole Philox blocks -- quad_noise in the kernel)
        const int cstep = (d->rng != nullptr && d->OW % 4 == 0) ? 4 : d->S;
        int CO = 0;
        for (int c = cstep; c <= d->OW + cstep; c += cstep) {
            const int need = (c - 1 + d->K - 1) / d->S + 1;
            if (need > tc) break;
            CO = c;
        }
        return CO;
    };
    int best_tc = 0, best_ro = 0, best_co = 0;
    double best_cost = 1e30;
    for (int tc = 16; tc <= d->IW && tc <= p->NPIX; tc += 16) {
        const bool full = tc == d->IW;
        if (!full && d->IW % tc != 0 && tc * 2 > d->IW) continue;   // (a block that is neither the row nor a sensible fraction)
        int ro = ro_for(rows_for(tc));
        int co = full ? d->OW : co_for(tc);
        if (ro <= 0 || co <= 0) continue;
        if (ro > d->OH) ro = ((d->OH + d->S - 1) / d->S) * d->S;
        if (co > d->OW) co = d->OW;
        const int nrows = (ro - 1 + d->K - 1) / d->S + 1;
        const double cost = (double)nrows * tc / ((double)ro * co) + (full ? 0.0 : 0.02);   // prefer whole rows on a tie
        if (cost < best_cost) { best_cost = cost; best_tc = tc; best_ro = ro; best_co = co; }
    }
    VG_CHECK_ARG(best_tc > 0, VG_ENOSUP);
    p->TC = best_tc;
    p->RO = best_ro;
    p->CO = best_co;
    p->tiles_y = (d->OH + p->RO - 1) / p->RO;
    p->tiles_x = (d->OW + p->CO - 1) / p->CO;
    return 0;
}

template <int NT, int KC>
inline void tn_launch(const vg_tn_desc* d, const TnPlan& p, hipStream_t s) {
    dim3 grid(d->B * p.tiles_y * p.tiles_x), block(256);
    constexpr int NPIX = NT <= 2 ? 512 : 256;
    if (d->K == 3) vg_launch_timed(2, (tnconv_kernel<NT, KC, NPIX, 3, 1>), grid, block, 0, s, *d, p.RO, p.tiles_y, p.CO, p.tiles_x, p.TC);
    else vg_launch_timed(2, (tnconv_kernel<NT, KC, NPIX, 4, 2>), grid, block, 0, s, *d, p.RO, p.tiles_y, p.CO, p.tiles_x, p.TC);
}


// ---------------------------------------------------------------------------------------------------------------------
// vg_edge_wgrad -- weight gradient of the edge layers (the Discriminator's / Encoder's first Conv2d, the Generator's
// last ConvTranspose2d):   G[j = (kh*K + kw)*4 + n][c] = sum_pix  Nr[b][py*S - P + kh][px*S - P + kw][n] * Wd[b][py][px][c]
// over the pixels (b, py, px) of the WIDE operand Wd (C = 32 | 64 channels: dY of the conv, the input of the convT);
// Nr is the 3-channel tensor on the other side (the image, or the image gradient), 8 bf16 per pixel, channel 3 is zero.
// The generic wgrad kernel (wgrad.hip) runs this as a 128 x 128 tile of which 1/4 is used, gathers the narrow operand
// in 16-byte granules through LDS-DMA and writes 17-33 MB of split slabs for a 3 K-element result.  Here:
//   * a workgroup loops over tiles of 256 wide pixels (whole grid rows); the wide tile goes HBM -> LDS by LDS-DMA
//     (each byte read once), the narrow receptive field of the tile (a (R*S + K - S) x ((W-1)*S + K) pixel patch)
//     is staged once, zero-filled outside the image;
//   * both MFMA operands have the pixel as reduction index, i.e. the strided dimension of both tensors: fragments
//     come from ds_read_b64_tr_b16 (cdna_hip_programming.md T10).  The narrow fragment is read STRAIGHT from the
//     patch: lane (k row q, column quad p) of a 16-lane group supplies the address of the 4 channels of tap p of pixel
//     q -- the im2col matrix is never materialised.  The wide tile's 32-byte column blocks are XOR-swizzled with
//     (pixel >> 1) & 3 (C = 64; (pixel >> 2) & 1 for C = 32) on the DMA source side: conflict-free transposed reads;
//   * wave w owns the 16 rows j = 16w .. 16w+15 (4 taps) x all C columns; accumulators live across the workgroup's
//     tiles; one [J][C] f32 partial per workgroup, summed in fixed order by edge_wgrad_reduce_kernel, which also
//     scatters into the reference layout dW[c*s_c + n*s_n + kh*K + kw].
constexpr int EW_TP = 256;                 // wide pixels per tile
// bytes of LDS for the narrow patch: 22 KB beside the 32 KB wide tile of 64-channel operands (two workgroups per CU); the
// 16- / 32-channel operands (S >= 128 members of the size family) have 8 / 16 KB wide tiles and room for a 26 KB patch -- at
// 256-pixel-wide images that is TWO wide rows per tile instead of one (a 6 x 258-pixel patch), i.e. half the tiles and half
// the re-read halo rows (round 4)
constexpr int ew_patch_max(int ct) { return ct == 4 ? 22 * 1024 : 26 * 1024; }

typedef __attribute__((ext_vector_type(4))) __bf16 ew_bf16x4;

template <int CT>                          // C = 16 * CT channels of the wide operand (CT = 1 | 2 | 4)
__global__ __launch_bounds__(256) void edge_wgrad_kernel(const vg_ew_desc d, const int R, const int tiles_per_img,
                                                         const int ntiles, const int JT) {
    constexpr int RB = CT * 32;                                   // bytes per wide pixel
    constexpr int PATCH_MAX = ew_patch_max(CT);
    // [wide tile 256 x RB][patch: slot 0 = zero pixel, then PR x PW pixels of 16 B][pixel table 256 x int]
    __shared__ __attribute__((aligned(16))) unsigned char smem[EW_TP * RB + PATCH_MAX + EW_TP * 4];
    unsigned char* const wide = smem;
    unsigned char* const patch = smem + EW_TP * RB;
    int* const ptab = reinterpret_cast<int*>(smem + EW_TP * RB + PATCH_MAX);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int PW = (d.WW - 1) * d.S + d.K, PR = (R - 1) * d.S + d.K;
    // this lane's tap (column quad p of j tile `wave`): byte offset inside the patch, or the zero pixel
    const int tap = wave * 4 + p;
    const int kh = tap / d.K, kw = tap - kh * d.K;
    const bool tap_ok = tap < d.K * d.K;
    const int tapoff = tap_ok ? (kh * PW + kw) * 16 : 0;

    f32x4 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(d.Wd);
    const unsigned char* Nb = reinterpret_cast<const unsigned char*>(d.Nr);
    const unsigned char* Zp = reinterpret_cast<const unsigned char*>(d.zeros);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img, ty = tile - b * tiles_per_img;
        const int py0 = ty * R;
        const int rows = min(R, d.WH - py0);                       // last tile of an image may be short
        const int npix = rows * d.WW;
        __syncthreads();                                           // previous tile's fragment reads are done
        // ---- wide tile: LDS-DMA, 1 KiB per wave instruction, source block swizzled by the pixel ----
        {
            constexpr int PPI = 1024 / RB;                         // pixels per wave instruction (8 | 16)
            constexpr int UPP = RB / 16;                           // 16-byte units per pixel (8 | 4)
            const unsigned char* base = Wb + ((int64_t)(b * d.WH + py0) * d.WW) * RB;
#pragma unroll
            for (int it = 0; it < EW_TP / PPI / 4; ++it) {
                const int i0 = (it * 4 + wave_u) * PPI;            // first pixel of this wave instruction
                const int pix = i0 + lane / UPP, u = lane % UPP;
                const int f = CT == 4 ? ((pix >> 1) & 3) : CT == 2 ? ((pix >> 2) & 1) : 0;     // (C = 16: one 32-byte block, no swizzle)
                const int su = (((u >> 1) ^ f) << 1) | (u & 1);    // source unit that belongs at LDS unit u
                const unsigned char* src = pix < npix ? base + (int64_t)pix * RB + su * 16 : Zp;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(wide + i0 * RB), 16, 0, 0);
            }
        }
        // ---- narrow patch (zero outside the image) and the pixel table ----
        const int ny0 = py0 * d.S - d.P;
        for (int e = tid; e < PR * PW; e += 256) {
            const int pr = e / PW, pc = e - pr * PW;
            const int ny = ny0 + pr, nx = pc - d.P;
            u32x4 v = {0u, 0u, 0u, 0u};
            if ((unsigned)ny < (unsigned)d.NH && (unsigned)nx < (unsigned)d.NW)
                v = *reinterpret_cast<const u32x4*>(Nb + ((int64_t)(b * d.NH + ny) * d.NW + nx) * 16);
            *reinterpret_cast<u32x4*>(patch + 16 + e * 16) = v;
        }
        if (tid == 0) *reinterpret_cast<u32x4*>(patch) = u32x4{0u, 0u, 0u, 0u};
        {
            const int pyl = tid / d.WW, px = tid - pyl * d.WW;
            ptab[tid] = tid < npix ? 16 + (pyl * d.S * PW + px * d.S) * 16 : 0;     // pad pixels: wide rows are zero anyway
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- MFMA over the tile's pixels: 32 per step ----
        if (wave_u < JT) {
            const int nsteps = (npix + 31) >> 5;
            for (int ks = 0; ks < nsteps; ++ks) {
                const int r0 = ks * 32 + 4 * g + q, r1 = r0 + 16;
                // narrow fragment: 4 channels of tap p of pixels r0 / r1 (the zero pixel for padded taps)
                const unsigned char* a0 = patch + (tap_ok ? ptab[r0] + tapoff : 0);
                const unsigned char* a1 = patch + (tap_ok ? ptab[r1] + tapoff : 0);
                const ew_bf16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) ew_bf16x4*)a0);
                const ew_bf16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) ew_bf16x4*)a1);
                const bf16x8 af = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
                const int f0 = CT == 4 ? ((r0 >> 1) & 3) : CT == 2 ? ((r0 >> 2) & 1) : 0;
                const int f1 = CT == 4 ? ((r1 >> 1) & 3) : CT == 2 ? ((r1 >> 2) & 1) : 0;
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const unsigned char* b0 = wide + r0 * RB + ((c ^ f0) << 5) + 8 * p;
                    const unsigned char* b1 = wide + r1 * RB + ((c ^ f1) << 5) + 8 * p;
                    const ew_bf16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) ew_bf16x4*)b0);
                    const ew_bf16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) ew_bf16x4*)b1);
                    const bf16x8 bf = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[c], 0, 0, 0);
                }
            }
        }
    }
    // ---- partial [J][C] of this workgroup: rows j = 16*wave + 4*(lane>>4) + r, column c = 16*ct + (lane & 15) ----
    if (wave_u < JT) {
        float* out = d.ws + (int64_t)blockIdx.x * (JT * 16) * (CT * 16);
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(wave * 16 + (lane >> 4) * 4 + r) * (CT * 16) + c * 16 + (lane & 15)] = acc[c][r];
    }
}

// Fixed-order sum of the workgroups' partials in ONE launch (a one-level loop over ~500 partials per thread is a chain of
// dependent HBM round trips: it cost more than the kernel it finishes; two launches -- groups of parts, then groups --
// cost two launch floors, 8.3 us per layer).  A 16-wave workgroup owns 64 consecutive outputs: wave g sums parts
// [g*per, (g+1)*per) for them, 8 loads in flight per lane, one 256-byte run per part row; the EW_G wave sums are added in
// wave order through LDS and scattered into dW[c*s_c + n*s_n + tap].  (Same grouping, same order, same bits as the
// two-launch form it replaces.)
constexpr int EW_G = 16;

__global__ __launch_bounds__(64 * EW_G) void edge_wgrad_reduce_kernel(const vg_ew_desc d, const float* __restrict__ ws,
                                                                      int nparts, int J, int C) {
    __shared__ float red[EW_G][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + lane;                       // = j * C + c  (J * C is a multiple of 64)
    const int JC = J * C;
    const int per = (nparts + EW_G - 1) / EW_G;
    const int k0 = grp * per, k1 = min(nparts, k0 + per);
    float s = 0.f;
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ws[(int64_t)(k + u) * JC + idx];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < k1; ++k) s += ws[(int64_t)k * JC + idx];
    red[grp][lane] = s;
    __syncthreads();
    if (grp != 0) return;
    const int j = idx / C, c = idx - j * C;
    const int n = j & 3, tap = j >> 2;
    if (n >= d.N || tap >= d.K * d.K) return;
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < EW_G; ++g) t += red[g][lane];
    float* dst = d.dW + (int64_t)c * d.s_c + (int64_t)n * d.s_n + tap;
    *dst = d.accumulate ? *dst + t : t;
}

struct EwPlan { int R, tiles_per_img, ntiles, JT, CT, grid; int64_t ws_bytes; };

inline int ew_plan(const vg_ew_desc* d, EwPlan* p) {
    VG_CHECK_ARG(d != nullptr, VG_EINVAL);
    VG_CHECK_ARG(d->B > 0 && d->WH > 0 && d->WW > 0 && d->NH > 0 && d->NW > 0 && d->P >= 0, VG_EINVAL);
    VG_CHECK_ARG(d->C == 16 || d->C == 32 || d->C == 64, VG_ENOSUP);
    VG_CHECK_ARG(d->N >= 1 && d->N <= 3, VG_ENOSUP);              // channel 3 of the narrow operand must be the zero pad
    VG_CHECK_ARG((d->K == 3 || d->K == 4) && (d->S == 1 || d->S == 2), VG_ENOSUP);
    VG_CHECK_ARG(d->WW <= EW_TP, VG_ENOSUP);
    p->JT = (d->K * d->K * 4 + 15) / 16;                           // 3 (k3) | 4 (k4)
    p->CT = d->C / 16;
    int R = EW_TP / d->WW;
    const int64_t pmax = ew_patch_max(p->CT);
    while (R > 1 && (int64_t)(((R - 1) * d->S + d->K) * ((d->WW - 1) * d->S + d->K) + 1) * 16 > pmax) --R;
    VG_CHECK_ARG((int64_t)(((R - 1) * d->S + d->K) * ((d->WW - 1) * d->S + d->K) + 1) * 16 <= pmax, VG_ENOSUP);
    if (R > d->WH) R = d->WH;
    p->R = R;
    p->tiles_per_img = (d->WH + R - 1) / R;
    p->ntiles = d->B * p->tiles_per_img;
    p->grid = p->ntiles < 512 ? p->ntiles : 512;                   // 2 workgroups per CU, each walks several tiles
    p->ws_bytes = (int64_t)p->grid * p->JT * 16 * d->C * 4;          // one [J][C] partial per workgroup
    return 0;
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

extern "C" int64_t vg_edge_wgrad_ws_bytes(const vg_ew_desc* d) {
    EwPlan p;
    int rc = ew_plan(d, &p);
    return rc ? (int64_t)rc : p.ws_bytes;
}

extern "C" int vg_edge_wgrad(const vg_ew_desc* d, void* stream) {
    EwPlan p;
    int rc = ew_plan(d, &p);
    if (rc) return rc;
    VG_CHECK_ARG(d->Wd && d->Nr && d->dW && d->ws && d->zeros, VG_EINVAL);
    VG_CHECK_ARG(vg_aligned16(d->Wd) && vg_aligned16(d->Nr) && vg_aligned16(d->ws), VG_EALIGN);
    VG_CHECK_ARG(d->ws_bytes >= p.ws_bytes, VG_EINVAL);
    hipStream_t s = vg_stream(stream);
    if (p.CT == 4) vg_launch_timed(1, edge_wgrad_kernel<4>, dim3(p.grid), dim3(256), 0, s, *d, p.R, p.tiles_per_img, p.ntiles, p.JT);
    else if (p.CT == 1) vg_launch_timed(1, edge_wgrad_kernel<1>, dim3(p.grid), dim3(256), 0, s, *d, p.R, p.tiles_per_img, p.ntiles, p.JT);
    else vg_launch_timed(1, edge_wgrad_kernel<2>, dim3(p.grid), dim3(256), 0, s, *d, p.R, p.tiles_per_img, p.ntiles, p.JT);
    rc = VG_LAUNCH_RC();
    if (rc) return rc;
    const int JC = p.JT * 16 * d->C;                                // multiple of 256 (C >= 32, J >= 48)
    hipLaunchKernelGGL(edge_wgrad_reduce_kernel, dim3(JC / 64), dim3(64 * EW_G), 0, s, *d, d->ws, p.grid, p.JT * 16, d->C);
    return VG_LAUNCH_RC();
}

// Four-phase transposed gather-GEMM for the NARROW layers of the S >= 128 stacks (<= 32 output channels): ALL FOUR sub-pixel
// phases of a tile in one workgroup.  Included by conv_gemm.hip (inside its anonymous namespace), after conv_patch.hpp.
//
// Why.  With 16 or 32 output channels the 4-phase form of ConvTranspose2d(k4, s2, p1) (gan_code.py:21-49 at img_size >= 128)
// and of the data gradient of Conv2d(k4, s2, p1) (gan_code.py:61-84) is a STREAM, not a GEMM: 24 KB of input patch per 256
// grid pixels and phase, 16 MFMAs per wave.  Run phase by phase (ggp_kernel, blockIdx.z = phase) every phase re-reads the
// patch and -- worse -- writes every second output pixel: 32 or 64 bytes per pixel with the neighbour belonging to another
// phase, i.e. quarter / half cache lines that leave for HBM half-written (ConvTranspose2d(32 -> 16) at S = 256, B = 32: 97 us
// on the generic tile, 60 us on the one-phase patch tile, against 100 MB of compulsory traffic = 20 us).  Here a workgroup owns
// 256 grid pixels = R whole grid rows (or whole images), fetches the UNION patch (R + 2) x (GW + 2) once per 32-channel chunk
// together with the 16 weight tiles (4 phases x 4 taps), accumulates the four phases side by side and writes 2R complete
// output rows: every output line leaves whole, the input is read once.
//
// Contract (checked by ggq_geometry): bf16, LDS-DMA; nphase = 4 with phase p = 2*py + px writing output pixel (2*gy + py,
// 2*gx + px) and its tap (a, b) reading grid pixel (gy + py - a, gx + px - b) (k4 s2 p1: y0 = py, DY = -1); N <= 32; IC % 32 == 0, Kp = 4*IC;
// GW, GH powers of two, 4 <= GW <= 128; M % 256 == 0; OH = 2*GH, OW = 2*GW.
// LDS: [patch NR x 4 KB | (re-used as) output staging][16 weight tiles x BN x 64 B]: 52 KB (BN = 16: three workgroups per CU),
// 72 KB (BN = 32: two).  One chunk at a time (issue everything, wait, multiply): the layers that come here have one or two
// chunks, and the other workgroups of the CU cover the wait.

struct Q4Geo {
    int R, IMGS, PW, PIMG, NPP, nct;   // grid rows per image in a tile, images per tile, patch width, patch pixels per image / per tile, 32-channel chunks
    int lg_rows, lg_gw;                // log2(grid pixels per image in a tile), log2(GW)
};

template <int BN, int NR>
__global__ __launch_bounds__(256, BN == 16 ? 3 : 2) void ggq_kernel(const vg_gg_desc d, const Q4Geo g) {
    constexpr int TM = 4, TN = BN / 16, NT = 256, BM = 256;
    constexpr int PBUF = NR * NT * 16;                       // patch image: 64 B per pixel (32 channels)
    constexpr int WTILE = BN * 64;                           // one (phase, tap) weight tile
    constexpr int WBUF = 16 * WTILE;
    constexpr int WR = WBUF / (NT * 16);                     // weight DMA rounds: 4 | 8
    constexpr int CP = BN * 2 + 16;                          // staging pitch of one (grid pixel, px) output pixel
    constexpr int STG = BM * 2 * CP;                         // the two output pixels (px = 0, 1) of every grid pixel for one py
    constexpr int REGION0 = PBUF > STG ? PBUF : STG;
    constexpr int SEGS = BN * 2 / 16;
    static_assert(BN == 16 || BN == 32, "narrow layers only");
    static_assert(4 * 4 * BN * 2 * 4 <= WBUF, "statistics scratch lives in the weight buffer");
    __shared__ __attribute__((aligned(16))) unsigned char smem[REGION0 + WBUF];
    unsigned char* const pbuf = smem;
    unsigned char* const wbuf = smem + REGION0;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int GHW = d.GH * d.GW;
    const int m_tiles = (d.B * GHW) / BM;
    // each XCD takes a contiguous run of tiles (neighbouring tiles share their halo rows in that XCD's L2)
    const int id = blockIdx.x;
    const int bx = (m_tiles & 7) == 0 ? (id & 7) * (m_tiles >> 3) + (id >> 3) : id;
    if (bx >= m_tiles) return;
    const int m0 = bx * BM;
    const int b0 = m0 / GHW;
    const int gy0 = (m0 - b0 * GHW) >> g.lg_gw;               // 0 for multi-image tiles

    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(d.X);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(d.W);
    const unsigned char* Zp = reinterpret_cast<const unsigned char*>(d.zeros);
    const uint32_t pix_bytes = (uint32_t)d.IC * 2u;

    // ---- DMA sources (computed once; a live lane advances by 64 B per chunk) ----
    // patch: LDS slot sidx = 256*r + tid -> patch pixel pp = sidx >> 2, unit (sidx & 3) ^ ((pp >> 1) & 3) (the read-side key
    // of ggp_kernel: a ds_read_b128 of 16 consecutive pixels is conflict-free at every shift)
    const unsigned char* a_src[NR];
    uint32_t a_live = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int sidx = NT * r + tid;
        const int pp = sidx >> 2;
        const int q = (sidx & 3) ^ ((pp >> 1) & 3);
        const int img = pp / g.PIMG;
        const int rem = pp - img * g.PIMG;
        const int pr = rem / g.PW;
        const int pc = rem - pr * g.PW;
        const int b = b0 + img;
        const int iy = gy0 + pr - 1, ix = pc - 1;
        const bool ok = pp < g.NPP && b < d.B && (unsigned)iy < (unsigned)d.IH && (unsigned)ix < (unsigned)d.IW;
        a_src[r] = ok ? Xb + ((uint32_t)((b * d.IH + iy) * d.IW + ix) * pix_bytes + (uint32_t)q * 16u) : Zp;
        a_live |= ok ? (1u << r) : 0u;
    }
    // weights: slot idx = 256*r + tid -> unit idx & 3 of row n = (idx >> 2) % BN of tile (phase, tap) = idx / (4 * BN);
    // source unit swizzled by the row (gg_kernel's key)
    const unsigned char* w_src[WR];
    uint32_t w_live = 0;
#pragma unroll
    for (int r = 0; r < WR; ++r) {
        const int idx = NT * r + tid;
        const int n = (idx >> 2) & (BN - 1);
        const int pt = idx / (4 * BN);
        const int p = pt >> 2, t = pt & 3;
        const int qb = (idx & 3) ^ ((-(n >> 2)) & 3);
        const bool ok = n < d.N;
        w_src[r] = ok ? Wb + (((int64_t)p * d.N + n) * d.Kp + (int64_t)t * d.IC) * 2 + qb * 16 : Zp;
        w_live |= ok ? (1u << r) : 0u;
    }

    // ---- fragment addressing: the wave's 64 grid pixels, patch position of each (centre of the 3 x 3 neighbourhood) ----
    int ppbase[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wave * 64 + i * 16 + fr;
        const int img = r >> g.lg_rows;
        const int rr = r & ((1 << g.lg_rows) - 1);
        const int ry = rr >> g.lg_gw, rx = rr & (d.GW - 1);
        ppbase[i] = img * g.PIMG + (ry + 1) * g.PW + (rx + 1);
    }

    f32x4 acc[4][TM][TN];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[p][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < g.nct; ++c) {
        if (c > 0) __syncthreads();                              // the previous chunk's fragment reads are done
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_src[r],
                                             (__attribute__((address_space(3))) void*)(pbuf + (NT * r + 64 * wave) * 16), 16, 0, 0);
            a_src[r] += (a_live >> r & 1u) * 64u;
        }
#pragma unroll
        for (int r = 0; r < WR; ++r) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)w_src[r],
                                             (__attribute__((address_space(3))) void*)(wbuf + (NT * r + 64 * wave) * 16), 16, 0, 0);
            w_src[r] += (w_live >> r & 1u) * 64u;
        }
        VG_WAITCNT_VM(0);
        __builtin_amdgcn_s_barrier();
        // The 16 (phase, tap) products read the patch at 9 shifts only: phase (py, px), tap (a, b) reads grid pixel
        // (gy + py - a, gx + px - b) (k4 s2 p1: y0 = py, DY = -1).  One set of A fragments per shift, used by every
        // (phase, tap) that reads there (centre: 4, edges: 2, corners: 1): 36 + 16*TN fragment reads for 64*TN MFMAs.
        // (the patch width behind an opaque copy: the 36 shifted fragment addresses are cheap to form where they are used,
        // and hoisted out of the chunk loop they cost 36 registers for the whole kernel -- scratch)
        int pw = g.PW;
        asm volatile("" : "+s"(pw));
#pragma unroll
        for (int sy = 0; sy < 3; ++sy) {
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) {
                const int off = (sy - 1) * pw + (sx - 1);
                u32x4 fa[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int pp = ppbase[i] + off;
                    fa[i] = *reinterpret_cast<const u32x4*>(pbuf + pp * 64 + ((fg ^ ((pp >> 1) & 3)) << 4));
                }
#pragma unroll
                for (int py = 0; py < 2; ++py)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int px = 0; px < 2; ++px)
#pragma unroll
                            for (int b = 0; b < 2; ++b) {
                                if (py - a != sy - 1 || px - b != sx - 1) continue;        // folded: all six are unrolled constants
                                const int p = py * 2 + px, t = a * 2 + b;
                                u32x4 fb[TN];
#pragma unroll
                                for (int j = 0; j < TN; ++j) {
                                    const int r = j * 16 + fr;
                                    fb[j] = *reinterpret_cast<const u32x4*>(wbuf + (p * 4 + t) * WTILE + r * 64 + ((fg ^ ((-(r >> 2)) & 3)) << 4));
                                }
#pragma unroll
                                for (int i = 0; i < TM; ++i)
#pragma unroll
                                    for (int j = 0; j < TN; ++j)
                                        acc[p][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                                                               __builtin_bit_cast(bf16x8, fb[j]), acc[p][i][j], 0, 0, 0);
                            }
                // one shift at a time: left alone, the scheduler hoists the fragment reads of all nine above the first MFMA
                // (36 + 16*TN fragments: scratch); the CU's other workgroups cover the wait
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---------------- epilogue: bias / activation, BatchNorm partial sums per phase, 2R complete output rows ----------------
    typedef ElemT<VG_BF16> E;
    float biasv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nc = j * 16 + fr;
        biasv[j] = (d.bias != nullptr && nc < d.N) ? d.bias[nc] : 0.f;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[p][i][j][r] + biasv[j];
                    if (d.act != VG_ACT_NONE) v = act_fwd(v, d.act, d.act_slope);
                    acc[p][i][j][r] = v;
                }
    __syncthreads();                                             // every wave is through with the patch and the weights

    if (d.stats != nullptr) {
        float* red = reinterpret_cast<float*>(wbuf);             // [wave][phase][BN][2]
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[p][i][j][r];
                        a += v;
                        b += v * v;
                    }
                a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
                b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
                if (fg == 0) {
                    const int c = j * 16 + fr;
                    red[((wave * 4 + p) * BN + c) * 2 + 0] = a;
                    red[((wave * 4 + p) * BN + c) * 2 + 1] = b;
                }
            }
        __syncthreads();
        if (tid < 4 * BN) {
            const int p = tid / BN, c = tid - p * BN;
            if (c < d.N) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) { a += red[((w * 4 + p) * BN + c) * 2]; b += red[((w * 4 + p) * BN + c) * 2 + 1]; }
                const int64_t part = (int64_t)p * m_tiles + bx;
                d.stats[(part * 2 + 0) * d.N + c] = a;
                d.stats[(part * 2 + 1) * d.N + c] = b;
            }
        }
    }

    unsigned char* Yb = reinterpret_cast<unsigned char*>(d.Y);
    const int oc_bytes = d.OC * 2;
#pragma unroll
    for (int py = 0; py < 2; ++py) {
        if (py) __syncthreads();                                 // the first half's stores have read their staging
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = wave * 64 + i * 16 + fg * 4 + r;
                        const int col = j * 16 + fr;
                        *reinterpret_cast<typename E::type*>(smem + (row * 2 + px) * CP + col * 2) = E::from_f32(acc[py * 2 + px][i][j][r]);
                    }
        __syncthreads();
        // consecutive lanes: 16-byte segments of one output pixel, then px, then gx -- whole output rows, in address order
#pragma unroll
        for (int it = 0; it < (BM * 2 * SEGS) / NT; ++it) {
            const int u = tid + it * NT;
            const int seg = u % SEGS;
            const int rp = u / SEGS;                             // row * 2 + px
            const int px = rp & 1, row = rp >> 1;
            const int img = row >> g.lg_rows;
            const int rr = row & ((1 << g.lg_rows) - 1);
            const int ry = rr >> g.lg_gw, rx = rr & (d.GW - 1);
            const int oy = (gy0 + ry) * 2 + py, ox = rx * 2 + px;
            const int64_t op = ((int64_t)(b0 + img) * d.OH + oy) * d.OW + ox;
            const int cb = seg * 16;
            if (cb < oc_bytes) {
                u32x4 v = *reinterpret_cast<const u32x4*>(smem + rp * CP + cb);
                if (d.mask_x != nullptr)
                    v = mask_segment<VG_BF16>(v, *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(d.mask_x) +
                                                                                op * oc_bytes + cb), d.mask_act, d.mask_slope);
                *reinterpret_cast<u32x4*>(Yb + op * oc_bytes + cb) = v;
            }
        }
    }
}

inline bool use_phase4() { return vg_sw().gg_phase4 != 0; }     // VG_GG_PHASE4 (common.hpp: switches are read once at load)

// Does the descriptor have the 4-phase k4 s2 p1 transposed form this kernel takes?  -> geometry, and the DMA rounds (7 | 9).
inline bool ggq_geometry(const vg_gg_desc* d, Q4Geo* g, int* nr) {
    if (!use_phase4() || d->zeros == nullptr || d->nphase != 4 || d->N > 32) return false;
    if (d->TH != 2 || d->TW != 2 || d->SY != 1 || d->SX != 1 || (d->DY != 1 && d->DY != -1) || (d->DX != 1 && d->DX != -1)) return false;
    if (d->OSY != 2 || d->OSX != 2 || d->OH != 2 * d->GH || d->OW != 2 * d->GW) return false;
    if (d->IC % 32 != 0 || d->Kp != 4 * d->IC) return false;
    if (d->DY != -1 || d->DX != -1) return false;
    for (int p = 0; p < 4; ++p)                                 // phase p = 2*py + px; tap (a, b) reads grid pixel (gy + py - a, gx + px - b)
        if (d->ooy[p] != (p >> 1) || d->oox[p] != (p & 1) || d->y0[p] != (p >> 1) || d->x0[p] != (p & 1)) return false;
    const int GW = d->GW, GH = d->GH;
    if (GW < 4 || GW > 128 || (GW & (GW - 1)) || (GH & (GH - 1))) return false;
    const int64_t M = (int64_t)d->B * GH * GW;
    if (M % 256 != 0) return false;
    // two or three workgroups per CU, each a chain of waits: below two tiles per CU the one-phase tiles (four times the
    // workgroups) finish sooner (S = 256: Conv2d(32 -> 64)'s data gradient at B = 32, 128 tiles: 11.1 -> 16.1 us)
    if (M / 256 < vg_sw().gg_phase4_min) return false;
    int R, IMGS;
    if (GH * GW >= 256) { IMGS = 1; R = 256 / GW; if (R < 1 || GH % R) return false; }
    else { IMGS = 256 / (GH * GW); R = GH; }
    g->R = R; g->IMGS = IMGS; g->PW = GW + 2; g->PIMG = (R + 2) * (GW + 2); g->NPP = IMGS * g->PIMG;
    g->nct = d->IC / 32;
    int lg = 0; while ((1 << lg) < R * GW) ++lg;
    g->lg_rows = lg;
    lg = 0; while ((1 << lg) < GW) ++lg;
    g->lg_gw = lg;
    if (g->NPP <= 7 * 64) *nr = 7;
    else if (g->NPP <= 9 * 64) *nr = 9;
    else return false;
    return true;
}

inline int launch_phase4(const vg_gg_desc* d, const Q4Geo& g, int nr, hipStream_t s) {
    const int m_tiles = (d->B * d->GH * d->GW) / 256;
    const dim3 grid(m_tiles), block(256);
    if (d->N <= 16) {
        if (nr == 7) vg_launch_timed(0, (ggq_kernel<16, 7>), grid, block, 0, s, *d, g);
        else vg_launch_timed(0, (ggq_kernel<16, 9>), grid, block, 0, s, *d, g);
    } else {
        if (nr == 7) vg_launch_timed(0, (ggq_kernel<32, 7>), grid, block, 0, s, *d, g);
        else vg_launch_timed(0, (ggq_kernel<32, 9>), grid, block, 0, s, *d, g);
    }
    return VG_LAUNCH_RC();
}

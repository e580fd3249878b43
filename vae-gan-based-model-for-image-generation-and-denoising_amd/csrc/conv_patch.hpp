// Patch gather-GEMM: the 2x2-tap / stride-2 4x4-tap convolution forms with the input patch of a tile resident in LDS.
// Included by conv_gemm.hip (inside its anonymous namespace).  bf16, LDS-DMA, 128 x 128 (4 waves) or 256 x 128
// (8 waves) output tile.
//
// Why.  gg_kernel re-gathers the A operand once per filter tap: a 128-row tile moves 4 x 128 pixels x 64 B per
// 32-channel chunk for a 2x2-tap form although those rows are the SAME input pixels shifted by one.  The kernel
// sits on the per-CU L2->LDS fill rate, so bytes per FLOP are what bounds it.  Here a tile's
// input patch -- (R+1) x (GW+1) pixels for R whole grid rows -- is fetched ONCE per 32-channel chunk (<= 192 pixels
// instead of 512) and the four taps read it at four shifted positions; only the weight tiles still stream per tap.
// LDS fill per (chunk, 4 taps): <= 12 KB patch + 32 KB weights, against 32 + 32 KB.
//
// Forms (vg_gg_desc):
//   * one phase of the 4-phase transposed form: TH = TW = 2, SY = SX = 1, DY, DX = +-1 (blockIdx.z = phase);
//   * the direct stride-2 4x4 convolution: TH = TW = 4, SY = SX = 2, DY = DX = 1, split into its 4 input-parity
//     classes, each of which IS a 2x2-tap stride-1 form on the parity-subsampled input.
// Tile = 128 consecutive grid pixels = IMGS images x R whole grid rows (GW a power of two, 4 ... 64).
//
// Pipeline per workgroup: chunks j = (class, 32-channel chunk); patch(j) lives in pbuf[j & 1] and is fetched while
// chunk j-1 is multiplied (3 DMA rounds, one per tap stage); weight tiles ride a 3-deep ring, 2 stages ahead, exactly
// as in gg_kernel.  One s_barrier per tap stage, counted s_waitcnt vmcnt (all DMA of a wave completes in order).

struct PatchGeo {
    int R, IMGS, PW, PIMG, NPP;     // grid rows per image in a tile, images per tile, patch width, pixels per image, total
    int ncy, ncx, nct;              // parity classes per dimension, 32-channel chunks per (class, tap)
    int n_major;                    // XCD-major over n tiles (conv_gemm.hip n_major())
};

// Taps per barrier stage / weight-ring slots: 1 tap x 3 slots (48.5 KB, 3 workgroups per CU) or 2 taps x 2 slots
// (56.5 KB, 2 workgroups per CU, half the barriers per FLOP -- the shape that won for gg_kernel and wgrad).
#ifndef VG_GP_TPS
#define VG_GP_TPS 2
#endif
constexpr int GP_BN = 128, GP_BST = GP_BN * 64, GP_TPS = VG_GP_TPS, GP_NB = GP_TPS == 2 ? 2 : 3;
constexpr int GP_BSTAGE = GP_TPS * GP_BST;

__device__ __forceinline__ void gp_wait(int n) {
    if (n >= 3) VG_WAITCNT_VM(3);
    else if (n == 2) VG_WAITCNT_VM(2);
    else if (n == 1) VG_WAITCNT_VM(1);
    else VG_WAITCNT_VM(0);
}

// Waves per SIMD the register allocation must leave room for: the narrow variants whose stage buffers are exactly 40 KB
// (128 rows) / 80 KB (256 rows) run FOUR / TWO workgroups per CU = 4 waves per SIMD (<= 128 registers).
constexpr int gp_min_waves(int WM, int BN, int NR_) {
#ifdef VG_GP_OCC3           // A/B build: the occupancy of the kernels before the overlaid pixel table (3 workgroups per CU)
    return WM == 4 ? 2 : 1;
#endif
    if (GP_TPS != 2) return WM == 4 ? 2 : 1;
    if (WM == 4) return BN <= 64 ? 4 : 2;
    return (2 * (NR_ ? NR_ : 4) * 4096 + 256 * BN <= 40960) ? 4 : 1;
}

// WM = 2: 128-row tile, 4 waves, patch <= 256 pixels (4 DMA rounds) or <= 192 (NR_ = 3); LDS = 2 patch buffers + 2 weight
//         slots of 2 taps: 64.5 KB at BN = 128 (2 workgroups per CU), 48 KB at BN = 64, exactly 40 KB for BN = 64 with 3
//         rounds and for BN = 32 (FOUR workgroups per CU, <= 128 registers: gp_min_waves).
// WM = 4, BN = 64: 256 x 64, 8 waves, 80 KB, TWO workgroups per CU (<= 128 registers), for launches with >= 512 such tiles.
// WM = 4: 256-row tile, 8 waves sharing every weight tile (half the weight traffic per FLOP), patch <= 384 pixels,
//         all four taps of a chunk per barrier, 113 KB LDS: ONE workgroup (8 waves, up to 256 registers each) per CU.
//         History: with a 128-register cap (two workgroups per CU) it spilled and lost 30 %; at one workgroup per CU
//         with two taps per barrier it was on par; with four taps per barrier it wins where the launch has >= 384
//         tiles (G3 fprop 742 -> 780, G4 dgrad 689 -> 750 TFLOP/s) and loses at 256 tiles (G2 fprop 835 -> 806).
#ifdef VG_DBG_STAMPS     // diagnostic build only (tools/probes/clock_probe.py): shader-clock / 100 MHz stamps of workgroup phases
__device__ unsigned long long vg_dbg_stamps[8192 * 8];
#define VG_STAMP(slot) do { if (threadIdx.x == 0) { const unsigned w_ = (blockIdx.x + gridDim.x * blockIdx.z) & 8191u; \
        vg_dbg_stamps[w_ * 8 + (slot) * 2] = __builtin_amdgcn_s_memtime(); \
        vg_dbg_stamps[w_ * 8 + (slot) * 2 + 1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define VG_STAMP(slot) do { } while (0)
#endif

template <int WM, int BN = GP_BN, int NR_ = 0>       // NR_: patch DMA rounds (0 = by tile shape); 3 where the patch has <= 192 pixels
__global__ __launch_bounds__(128 * WM, gp_min_waves(WM, BN, NR_)) void ggp_kernel(const vg_gg_desc d, const PatchGeo g) {
    // BN = 128: wave tile 64 x 64; BN = 64 (layers with <= 64 output channels): wave tile 64 x 32; BN = 32: 64 x 16
    constexpr int BM = 64 * WM, WN = 2, TM = 4, TN = BN / 32, NT = 128 * WM, BST = BN * 64, WNC = BN / 2;
    constexpr int NR = NR_ ? NR_ : ((WM == 4 || GP_TPS != 2) ? 3 : 4);   /* WM == 4: 3 rounds of 512 lanes = 384 pixels */   // patch DMA rounds: 192 (3) or 256 (4) pixels per 128 rows
    constexpr int GP_PBUF = NR * NT * 16;           // NR rounds of NT lanes x 16 B
    constexpr int TPS = (WM == 4 && GP_TPS == 2) ? 4 : GP_TPS;   // 8 waves, one workgroup per CU: all 4 taps per barrier
    constexpr int NB = TPS >= 2 ? 2 : 3, BSTAGE = TPS * BST;   // (the TPS >= 2 loop alternates two slots)
    constexpr int BJ = (BN * 4 + NT - 1) / NT;      // weight-tile DMA instructions per wave and tap (2 / 1)
    // 256 x 64: the 64-row weight tile needs only 256 of the 512 lanes -> waves 4-7 skip the weight DMA (legal here:
    // that variant never uses counted vmcnt waits, every stage drains with vmcnt(0))
    constexpr bool B_HALF = (BN * 4 < NT);
    static_assert(BN == 128 || BN == 64 || BN == 32, "supported shapes");
    static_assert(!B_HALF || TPS >= 2, "partial weight issue needs the drain-every-stage loop");
    // [patch buffers 2 x 12|24 KB][weight ring 3 x 8 KB][output-pixel table]
    // OVL (narrow tiles; 64 wide with 3 patch rounds and 32 wide: exactly 40 KB of stage buffers): the epilogue's output-pixel table lives in
    // the weight-ring slot the LAST stage does not read, so that FOUR workgroups fit a CU's 160 KB (the kernel is bound
    // by the LDS-DMA bytes in flight per CU: short-K tiles wait ~1 us for every stage they issue, section 9).
#ifdef VG_GP_OCC3
    constexpr bool OVL = false;
#else
    constexpr bool OVL = (TPS >= 2 && BN <= 64);
#endif
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * GP_PBUF + NB * BSTAGE + (OVL ? 0 : BM * 4)];
    unsigned char* const pbuf = smem;
    unsigned char* const bring = smem + 2 * GP_PBUF;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / WN, wn = wave % WN;
    const int phase = blockIdx.z;
    VG_STAMP(0);
    int bx, by;
    {
        const int n_tiles = (d.N + BN - 1) / BN;
        const int id = blockIdx.x;
        const int xcd = id & 7, slot = id >> 3;
        if (g.n_major) {
            const int mt = (int)gridDim.x / n_tiles;
            bx = slot % mt;
            by = (slot / mt) * 8 + xcd;
        } else {
            by = slot % n_tiles;
            bx = (slot / n_tiles) * 8 + xcd;
        }
    }
    const int M = d.B * d.GH * d.GW;
    const int GHW = d.GH * d.GW;
    const int m_tiles_ = M / BM;                               // eligibility: M % 128 == 0
    if (bx >= m_tiles_) return;
    const int m0 = bx * BM, n0 = by * BN;
    const int b0 = m0 / GHW;
    const int gy0 = (m0 - b0 * GHW) / d.GW;                    // 0 for multi-image tiles

    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(d.X);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(d.W);
    const unsigned char* Zp = reinterpret_cast<const unsigned char*>(d.zeros);
    const uint32_t pix_bytes = (uint32_t)d.IC * 2u;
    const uint32_t wrow_bytes = (uint32_t)d.Kp * 2u;
    const int ncls = g.ncy * g.ncx;
    const int J = ncls * g.nct;                                 // chunks
    const int S = J * 4;                                        // tap stages
    // stage s of the TPS >= 2 loop reads ring slot s & 1; J * (4 / TPS) stages -> the other slot is free from the last barrier on
    int* const opix_tab = reinterpret_cast<int*>(OVL ? bring + ((J * (4 / TPS)) & 1) * BSTAGE : smem + 2 * GP_PBUF + NB * BSTAGE);

    // ---- weight-tile DMA lanes: rows lrow, lrow + 64; source unit swizzled by the row (as gg_kernel) ----
    const int lrow = tid >> 2;
    const int qb = (tid & 3) ^ ((-(lrow >> 2)) & 3);
    const unsigned char* b_base[BJ];
    uint32_t b_live[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int n = n0 + lrow + (NT / 4) * j;
        const bool ok = n < d.N;
        b_base[j] = ok ? Wb + ((int64_t)phase * d.N + n) * wrow_bytes + qb * 16 : Zp;
        b_live[j] = ok ? 1u : 0u;
    }
    // ---- patch DMA lanes: LDS slot sidx = NT*r + tid -> pixel pp = sidx >> 2, stored unit sidx & 3.  The lane's
    // pixel coordinates are recomputed at every class change (<= 4 times per kernel) rather than kept in registers.
    // Unit swizzle of the patch image: unit u of pixel pp sits in slot u ^ ((pp >> 1) & 3).  The four taps read the
    // patch at offsets 0, 1, PW, PW + 1 (PW odd), so a fragment's 16 pixels start at ANY alignment; with this key a
    // ds_read_b128 of 16 consecutive pixels is conflict-free at every offset (4 LDS cycles; the row-tile key
    // (-(pp >> 2)) & 3 of gg_kernel cost 7.5-7.75 here: SQ_LDS_BANK_CONFLICT was 37-53 % of SQ_LDS_IDX_ACTIVE).
    const unsigned char* a_cur[NR];
    uint32_t a_live = 0;                                         // bit r: round r reads real data (else the zero page)
    auto patch_sources = [&](int cl) {
        const int py = cl / g.ncx, px = cl - py * g.ncx;
        // class (py, px): input pixel of patch position (pr, pc) is ((gy0 + pr)*SY + cy, pc*SX + cx)
        const int cy = d.y0[phase] + d.DY * py - (d.DY < 0 ? d.SY : 0);
        const int cx = d.x0[phase] + d.DX * px - (d.DX < 0 ? d.SX : 0);
        a_live = 0;
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
            const int iy = (gy0 + pr) * d.SY + cy;
            const int ix = pc * d.SX + cx;
            const bool ok = pp < g.NPP && b < d.B && (unsigned)iy < (unsigned)d.IH && (unsigned)ix < (unsigned)d.IW;
#ifdef VG_ABL_A_L2          // timing-only build: every patch read hits a 256 KB window (L2-resident)
            a_cur[r] = ok ? Xb + (((uint32_t)((b * d.IH + iy) * d.IW + ix) * pix_bytes) & 0x3ffc0u) + (uint32_t)q * 16u : Zp;
            a_live |= 0u;
#else
            a_cur[r] = ok ? Xb + ((uint32_t)((b * d.IH + iy) * d.IW + ix) * pix_bytes + (uint32_t)q * 16u) : Zp;
            a_live |= ok ? (1u << r) : 0u;
#endif
        }
    };
    // issue-side cursors
    int pj_cl = 0, pj_c = 0;                                    // (class, chunk) of the NEXT patch to fetch
    auto issue_patch_round = [&](int buf, int r) {              // r is a compile-time constant at every call site
#ifdef VG_ABL_NO_A
        return;
#endif
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_cur[r],
                                         (__attribute__((address_space(3))) void*)(pbuf + buf * GP_PBUF + (NT * r + 64 * wave_u) * 16),
                                         16, 0, 0);
        a_cur[r] += (a_live >> r & 1u) * 64u;
    };
    auto patch_advance = [&]() {                                // after the 3 rounds of one patch
        if (++pj_c == g.nct) { pj_c = 0; ++pj_cl; if (pj_cl < ncls) patch_sources(pj_cl); }
    };
    int bs_cl = 0, bs_c = 0, bs_k = 0;                          // (class, chunk, tap-in-class) of the NEXT weight stage
    auto issue_b = [&](int buf) {                                // the TPS taps of one stage
#ifdef VG_ABL_NO_B
        return;
#endif
#pragma unroll
        for (int tp = 0; tp < TPS; ++tp) {
            const int py = bs_cl / g.ncx, px = bs_cl - py * g.ncx;
            const int t = ((bs_k >> 1) * g.ncy + py) * d.TW + ((bs_k & 1) * g.ncx + px);
            const uint32_t koff = ((uint32_t)t * (uint32_t)d.IC + (uint32_t)bs_c * 32u) * 2u;
            if (!(B_HALF && wave_u >= (BN * 4) / 64)) {
#pragma unroll
                for (int j = 0; j < BJ; ++j) {
                    const unsigned char* src = b_base[j] + (b_live[j] ? koff : 0u);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(bring + buf * BSTAGE + tp * BST + ((NT / 4) * j + 16 * wave_u) * 64),
                                                     16, 0, 0);
                }
            }
            if (++bs_k == 4) { bs_k = 0; if (++bs_c == g.nct) { bs_c = 0; ++bs_cl; } }
        }
    };

    // ---- fragment addressing ----
    const int fr = lane & 15, fg = lane >> 4;
    int ppbase[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * 64 + i * 16 + fr;
        const int per_img = g.R * d.GW;
        const int img = r / per_img;
        const int rr = r - img * per_img;
        const int ry = rr / d.GW;
        ppbase[i] = img * g.PIMG + ry * g.PW + (rr - ry * d.GW);
    }
    const int sh_y1 = d.DY > 0 ? 1 : 0, sh_x1 = d.DX > 0 ? 1 : 0;   // patch shift of local tap a = 1 (a = 0: the other)

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int pb, int bb, int k) {
#ifdef VG_ABLATE_COMPUTE
        return;
#endif
        const int a = k >> 1, b = k & 1;
        const int shy = a ? sh_y1 : 1 - sh_y1, shx = b ? sh_x1 : 1 - sh_x1;
        const int tapoff = shy * g.PW + shx;
        const unsigned char* sa = pbuf + pb * GP_PBUF;
        const unsigned char* sb = bring + bb * BSTAGE + (k % TPS) * BST;
        u32x4 fa[TM], fb[TN];
        auto ld_a = [&](int i) {
            const int pp = ppbase[i] + tapoff;
#ifdef VG_ABL_NO_FRAG           // timing-only: no LDS fragment reads (the matrix pipe runs on register garbage)
            fa[i] = u32x4{(uint32_t)pp, (uint32_t)k, (uint32_t)lane, 0x3f803f80u};
            (void)sa;
#else
            fa[i] = *reinterpret_cast<const u32x4*>(sa + pp * 64 + ((fg ^ ((pp >> 1) & 3)) << 4));
#endif
        };
        auto ld_b = [&](int j) {
            const int r = wn * WNC + j * 16 + fr;
#ifdef VG_ABL_NO_FRAG
            fb[j] = u32x4{(uint32_t)r, (uint32_t)bb, (uint32_t)lane, 0x3f803f80u};
            (void)sb;
#else
            fb[j] = *reinterpret_cast<const u32x4*>(sb + r * 64 + ((fg ^ ((-(r >> 2)) & 3)) << 4));
#endif
        };
#ifdef VG_ABL_NO_MFMA           // timing-only: fragment reads stay, the MFMAs go
#define GP_MFMA(i, j) asm volatile("" ::"v"(fa[i]), "v"(fb[j]))
#else
#define GP_MFMA(i, j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), \
                                                                  __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0)
#endif
        if constexpr (TN == 4) {
            ld_a(0); ld_b(0); ld_b(1); ld_a(1); ld_b(2); ld_a(2); ld_b(3); ld_a(3);
            // MFMAs in the order their fragments arrive (fa0 fb0 | fb1 | fa1 | fb2 | fa2 | fb3 | fa3) ...
            GP_MFMA(0, 0);
            GP_MFMA(0, 1);
            GP_MFMA(1, 0); GP_MFMA(1, 1);
            GP_MFMA(0, 2); GP_MFMA(1, 2);
            GP_MFMA(2, 0); GP_MFMA(2, 1); GP_MFMA(2, 2);
            GP_MFMA(0, 3); GP_MFMA(1, 3); GP_MFMA(2, 3);
            GP_MFMA(3, 0); GP_MFMA(3, 1); GP_MFMA(3, 2); GP_MFMA(3, 3);
#ifndef VG_NO_SCHED
            // ... and a scheduling pipeline that interleaves the ds_read_b128s with them instead of "all reads, wait
            // for everything, all MFMAs" (0x100 = DS read, 0x008 = MFMA): two reads stay in flight ahead of the MFMA
            // that needs them
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 10, 0);
#endif
        } else if constexpr (TN == 1) {
            ld_a(0); ld_b(0); ld_a(1); ld_a(2); ld_a(3);
            GP_MFMA(0, 0); GP_MFMA(1, 0); GP_MFMA(2, 0); GP_MFMA(3, 0);
        } else {
            ld_a(0); ld_b(0); ld_b(1); ld_a(1); ld_a(2); ld_a(3);
            GP_MFMA(0, 0); GP_MFMA(0, 1);
            GP_MFMA(1, 0); GP_MFMA(1, 1);
            GP_MFMA(2, 0); GP_MFMA(2, 1);
            GP_MFMA(3, 0); GP_MFMA(3, 1);
#ifndef VG_NO_SCHED
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#endif
        }
#undef GP_MFMA
    };

    // ---- prologue: patch(0) (3 rounds), then the first weight stage(s) ----
    patch_sources(0);
#pragma unroll
    for (int r = 0; r < NR; ++r) issue_patch_round(0, r);
    patch_advance();
    VG_STAMP(1);
    if constexpr (TPS >= 2) {
        // ---- TPS taps per barrier, two weight slots: everything issued during a stage is drained at the next ----
        constexpr int SPC = 4 / TPS;                            // stages per chunk
        issue_b(0);
        const int S2 = J * SPC;
        int bb = 0, s = 0;
        for (int j = 0; j < J; ++j) {
            const int pb = j & 1;
            const bool more_p = j + 1 < J;
#pragma unroll
            for (int h = 0; h < SPC; ++h, ++s) {
#if defined(VG_ABL_DMA_FREE)          // timing-only: stages issued back to back (the counter's 63-deep limit is the only brake)
#elif defined(VG_ABL_DMA_AHEAD)       // timing-only: the newest VG_ABL_DMA_AHEAD DMA instructions may stay in flight (racy)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VG_ABL_DMA_AHEAD) : "memory");
                __builtin_amdgcn_s_barrier();
#else
                VG_WAITCNT_VM(0);
                __builtin_amdgcn_s_barrier();
#endif
                if (more_p) {
#pragma unroll
                    for (int r = 0; r < NR; ++r)
                        if (r >= (h * NR) / SPC && r < ((h + 1) * NR) / SPC) issue_patch_round(pb ^ 1, r);
                    if (h == SPC - 1) patch_advance();
                }
                if (s + 1 < S2) issue_b(bb ^ 1);
#pragma unroll
                for (int t = 0; t < TPS; ++t) compute(pb, bb, h * TPS + t);
                bb ^= 1;
            }
        }
    } else {
    issue_b(0);
    issue_b(1);
    // ---- main loop over chunks; the 4 tap stages of a chunk are unrolled (static DMA counts per stage) ----
    int bb = 0, wb = 2, s = 0;
    for (int j = 0; j < J; ++j) {
        const int pb = j & 1;
        const bool more_p = j + 1 < J;
#pragma unroll
        for (int k = 0; k < 4; ++k, ++s) {
            // in flight behind weight stage s: what stage s-1 issued = [one patch round (k = 1, 2, 3)] + [stage s+1]
            const bool tail = s + 1 >= S;
            gp_wait(tail ? 0 : (k == 0 ? BJ : (more_p ? BJ + 1 : BJ)));
            __builtin_amdgcn_s_barrier();
            if (k < 3 && more_p) {
                if (k == 0) issue_patch_round(pb ^ 1, 0);
                else if (k == 1) issue_patch_round(pb ^ 1, 1);
                else { issue_patch_round(pb ^ 1, 2); patch_advance(); }
            }
            if (s + 2 < S) issue_b(wb);
            compute(pb, bb, k);
            bb = bb == NB - 1 ? 0 : bb + 1;
            wb = wb == NB - 1 ? 0 : wb + 1;
        }
    }
    }

    VG_STAMP(2);
    // ---------------- epilogue (as gg_kernel<bf16, 128, 128>) ----------------
#ifdef VG_ABL_EPI_LGKM            // timing-only: workgroup barriers that do not wait for outstanding global stores
#define GP_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
#else
#define GP_SYNC() __syncthreads()
#endif
#ifdef VG_ABL_NO_EPI
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
#endif
#ifdef VG_ABL_EPI_RUNTIME_SKIP    // timing-only: the epilogue is compiled (same registers, same main loop) but never executed
    if (d.B > 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
#endif
    typedef ElemT<VG_BF16> E;
    constexpr int ESZ = 2;
    const bool flat = (d.nphase == 1 && d.OSY == 1 && d.OSX == 1 && d.GH == d.OH && d.GW == d.OW);
    for (int r = tid; r < BM; r += NT) {
        const int m = m0 + r;
        int op = -1;
        if (m < M) {
#ifdef VG_ABL_EPI_FLATOP
            if (true) {
#else
            if (flat) {
#endif
                op = m;
            } else {
                const int b = m / GHW;
                const int rem = m - b * GHW;
                const int gy = rem / d.GW;
                const int gx = rem - gy * d.GW;
                const int oy = gy * d.OSY + d.ooy[phase];
                const int ox = gx * d.OSX + d.oox[phase];
                if (oy < d.OH && ox < d.OW) op = (b * d.OH + oy) * d.OW + ox;
            }
        }
        opix_tab[r] = op;
    }
    float biasv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nc = n0 + wn * WNC + j * 16 + fr;
        biasv[j] = (d.bias != nullptr && nc < d.N) ? d.bias[nc] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] += biasv[j];
    if (d.act != VG_ACT_NONE) {                        // activation of a layer without BatchNorm, fused
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = act_fwd(acc[i][j][r], d.act, d.act_slope);
    }
    GP_SYNC();                                   // opix_tab visible; main-loop LDS reads are done

#ifdef VG_ABL_EPI_NOSTATS
    if (false) {
#else
    if (d.stats != nullptr) {
#endif
        float* red = reinterpret_cast<float*>(smem);   // [WM][BN][2]
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm * 64 + i * 16 + fg * 4 + r;
                    if (opix_tab[row] >= 0) {
                        const float v = acc[i][j][r];
                        a += v;
                        b += v * v;
                    }
                }
            a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
            b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
            if (fg == 0) {
                const int c = wn * WNC + j * 16 + fr;
                red[(wm * BN + c) * 2 + 0] = a;
                red[(wm * BN + c) * 2 + 1] = b;
            }
        }
        GP_SYNC();
        if (tid < BN && n0 + tid < d.N) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) { a += red[(w * BN + tid) * 2]; b += red[(w * BN + tid) * 2 + 1]; }
            const int64_t part = (int64_t)phase * m_tiles_ + bx;
            d.stats[(part * 2 + 0) * d.N + n0 + tid] = a;
            d.stats[(part * 2 + 1) * d.N + n0 + tid] = b;
        }
        GP_SYNC();
    }

    constexpr int CPITCH = BN * ESZ + 16;              // BM rows x 272 B = 34 | 68 KB <= the 48 | 72 KB of stage buffers
    static_assert(BM * CPITCH <= 2 * GP_PBUF + NB * BSTAGE, "C tile does not fit in LDS");
    static_assert(!OVL || (BM * CPITCH <= 2 * GP_PBUF && WM * BN * 8 <= 2 * GP_PBUF), "C tile / statistics scratch would overwrite the overlaid pixel table");
    constexpr int SEGS = BN * ESZ / 16;
    unsigned char* Yb = reinterpret_cast<unsigned char*>(d.Y);
    const int oc_bytes = d.OC * ESZ;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * 64 + i * 16 + fg * 4 + r;
                const int col = wn * WNC + j * 16 + fr;
                typename E::type* dst = reinterpret_cast<typename E::type*>(smem + row * CPITCH) + col;
#ifdef VG_ABL_EPI_NOSTAGE
                asm volatile("" ::"v"(acc[i][j][r]), "v"(dst));
#else
                *dst = E::from_f32(acc[i][j][r]);
#endif
            }
    constexpr int YIT = (BM * SEGS) / NT;
    static_assert((BM * SEGS) % NT == 0, "whole store iterations");
    GP_SYNC();
#pragma unroll
    for (int it = 0; it < YIT; ++it) {
        const int u = tid + it * NT;
        const int row = u / SEGS, seg = u - row * SEGS;
        const int op = opix_tab[row];
        const int cb = n0 * ESZ + seg * 16;
        if (op >= 0 && cb < oc_bytes) {
            u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * CPITCH + seg * 16);
            if (d.mask_x != nullptr)
                v = mask_segment<VG_BF16>(v, *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(d.mask_x) +
                                                                            (int64_t)op * oc_bytes + cb), d.mask_act, d.mask_slope);
#ifdef VG_ABL_EPI_NOSTORE     // timing-only build: everything but the global store (v kept live)
            asm volatile("" ::"v"(v));
#else
            *reinterpret_cast<u32x4*>(Yb + (int64_t)op * oc_bytes + cb) = v;
#endif
        }
    }
    VG_STAMP(3);
}

#undef GP_SYNC

inline bool use_patch() { return vg_sw().gg_patch != 0; }      // VG_GG_PATCH (common.hpp: switches are read once at load)

// Does the descriptor have one of the two patch forms, and does a bm-row tiling (128 or 256) line up?
inline bool patch_geometry(const vg_gg_desc* d, int bm, PatchGeo* g) {
    const bool transposed = d->TH == 2 && d->TW == 2 && d->SY == 1 && d->SX == 1 && (d->DY == 1 || d->DY == -1) &&
                            (d->DX == 1 || d->DX == -1);
    const bool direct2 = d->TH == 4 && d->TW == 4 && d->SY == 2 && d->SX == 2 && d->DY == 1 && d->DX == 1 &&
                         d->nphase == 1;
    if (!transposed && !direct2) return false;
    if (d->IC % 32 != 0 || d->Kp != d->TH * d->TW * d->IC) return false;
    const int GW = d->GW, GH = d->GH;
    if (GW < 4 || GW > 64 || (GW & (GW - 1)) || (GH & (GH - 1))) return false;
    const int64_t M = (int64_t)d->B * GH * GW;
    if (M % bm != 0) return false;
    int R, IMGS;
    if (GH * GW >= bm) { IMGS = 1; R = bm / GW; if (R < 1 || GH % R) return false; }
    else { IMGS = bm / (GH * GW); R = GH; }
    g->R = R; g->IMGS = IMGS; g->PW = GW + 1; g->PIMG = (R + 1) * (GW + 1); g->NPP = IMGS * g->PIMG;
    if (g->NPP > (bm == 256 ? 384 : (GP_TPS == 2 ? 256 : 192))) return false;
    g->ncy = d->TH / 2; g->ncx = d->TW / 2; g->nct = d->IC / 32;
    return true;
}

// Weight-stationary, persistent gather-GEMM for the SHORT-K transposed forms: one phase of a k4 s2 p1
// ConvTranspose2d forward / Conv2d data gradient whose whole weight matrix is small -- N = 64 output channels,
// K = 4 taps x IC <= 512 (the Generator's ConvTranspose2d(128 -> 64) at gan_code.py:42, the data gradient of the
// Discriminator's Conv2d(64 -> 128) at gan_code.py:66).  Included by conv_gemm.hip (inside its anonymous namespace);
// same descriptor, same packed weights, same patch geometry (conv_patch.hpp) as ggp_kernel.
//
// Why.  These layers have 4096+ output tiles of 16 K steps each.  In ggp_kernel every tile is its own workgroup: it
// streams the phase's 64 KB of weights again (more than its 42 KB of input patch), waits out one DMA round trip per
// tap stage with 16-32 MFMAs to hide it, and its epilogue overlaps nothing.  G4 forward: 67 us for 13.7 us of MFMA and
// a 20 us HBM floor (67 MB of output).  Here a workgroup
//   * loads its phase's weights ONCE into LDS (all 4 x IC/32 stage tiles, in the swizzled [n][64 B] image the
//     fragment reads of ggp_kernel expect) and keeps them for a contiguous run of M tiles;
//   * streams only input patches, through a ring of NSLOT slots with NSLOT-1 patches (one per 32-channel chunk) in
//     flight: counted s_waitcnt vmcnt, one raw s_barrier per chunk (4 taps = 32 MFMAs per wave), the DMA of the next
//     tiles runs under the epilogue of the current one;
//   * accumulates the BatchNorm partial sums of all its tiles in registers: ONE slab row per workgroup.
// One workgroup (4 waves) per CU: 64 KB weights + 6 x 12 KB | 4 x 16 KB patch ring + 18 KB C tile.

constexpr int GS_BM = 128, GS_NT = 256;

struct StatPlan { int wgs_per_phase, tiles_per_wg, nr; };

// OPT-IN (VG_GG_STATIONARY=1).  Measured on MI355X (S=64, B=128; tools/ab_stationary.sh, tools/gs_stamps.py): parity-exact
// (bit-identical outputs to ggp_kernel) but SLOWER -- G4 forward 82 us against 69 us, D1 data gradient (2B) 43 against 33.
// s_memtime stamps of one wave, per tile of 11 070 cycles: MFMA chunks 2 800 (ideal 2 048), patch DMA issue 1 830 (~150
// cycles per global_load_lds, guide: 100-185), counted waits + barriers 1 290, epilogue 4 880 (bias / statistics / bf16
// conversion VALU, C staging, stores and the LDS latencies between them).  With the 64 KB of weights only ONE 4-wave
// workgroup fits a CU, so nothing runs under a wave's issue and epilogue phases, while ggp_kernel's three workgroups
// per CU hide each other's.  What this structure needs next is wave specialisation (a second wave per SIMD that owns
// DMA issue and the epilogue, fed through LDS) -- DESIGN.md section 10.
inline bool stationary_enabled() {
    const char* e = getenv("VG_GG_STATIONARY");      // 1 = this kernel; 2 (default) = register-resident weights, conv_regweights.hpp
    return e ? atoi(e) == 1 : false;
}

// wait until at most `patches` x NR of this wave's vector-memory operations are outstanding (literal immediates)
template <int NR>
__device__ __forceinline__ void gs_wait_patches(int patches) {
    switch (patches) {
        case 0: VG_WAITCNT_VM(0); break;
        case 1: if constexpr (NR == 3) VG_WAITCNT_VM(3); else VG_WAITCNT_VM(4); break;
        case 2: if constexpr (NR == 3) VG_WAITCNT_VM(6); else VG_WAITCNT_VM(8); break;
        case 3: if constexpr (NR == 3) VG_WAITCNT_VM(9); else VG_WAITCNT_VM(12); break;
        case 4: if constexpr (NR == 3) VG_WAITCNT_VM(12); else VG_WAITCNT_VM(16); break;
        default: if constexpr (NR == 3) VG_WAITCNT_VM(15); else VG_WAITCNT_VM(20); break;
    }
}

// LDS writes of this wave visible to the workgroup after the barrier; does NOT drain vector memory (the patch ring)
__device__ __forceinline__ void gs_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int BN, int SMAX, int NR, int NSLOT>
__global__ __launch_bounds__(GS_NT) void ggs_kernel(const vg_gg_desc d, const PatchGeo g, const int wgs_per_phase,
                                                    const int tiles_per_wg) {
    constexpr int BM = GS_BM, NT = GS_NT, TM = 4, TN = BN / 32, WNC = BN / 2, BST = BN * 64;
    constexpr int PBUF = NR * NT * 16, CPITCH = BN * 2 + 16, SEGS = BN * 2 / 16;
    constexpr int OFF_P = SMAX * BST, OFF_C = OFF_P + NSLOT * PBUF, OFF_T = OFF_C + BM * CPITCH, OFF_R = OFF_T + BM * 4;
    constexpr int TOTAL = OFF_R + 2 * BN * 2 * 4;
    static_assert(BN == 64, "wave grid 2 x 2, wave tile 64 x 32");
    static_assert(TOTAL <= 160 * 1024, "LDS");
    static_assert(NSLOT >= 3 && NSLOT <= 7, "gs_wait_patches covers <= 5 patches in flight behind the one awaited");
    // ONE shared object (a second one next to an LDS-DMA target can make hipcc drain vmcnt before every ds_read)
    __shared__ __attribute__((aligned(16))) unsigned char smem[TOTAL];
    unsigned char* const wres = smem;
    unsigned char* const pring = smem + OFF_P;
    unsigned char* const cbuf = smem + OFF_C;
    int* const opix_tab = reinterpret_cast<int*>(smem + OFF_T);
    float* const red = reinterpret_cast<float*>(smem + OFF_R);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave >> 1, wn = wave & 1;
    const int phase = (int)blockIdx.x / wgs_per_phase;
    const int widx = (int)blockIdx.x - phase * wgs_per_phase;
    const int GHW = d.GH * d.GW;
    const int J = g.nct;                                       // 32-channel chunks = patches per tile
    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(d.X);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(d.W);
    const unsigned char* Zp = reinterpret_cast<const unsigned char*>(d.zeros);
    const uint32_t pix_bytes = (uint32_t)d.IC * 2u;

    // bias first, and USED before any DMA is issued: a later first use would make the compiler wait for this load
    // with vmcnt(0) inside the tile loop and drain the patch ring every tile.
    // Accumulator layout (the MFMA takes the WEIGHT fragment as its row operand): lane (fr, fg) holds pixel
    // 16 i + fr and the four consecutive channels 16 j + 4 fg + r -- one 8-byte LDS write per accumulator.
    const int fr = lane & 15, fg = lane >> 4;
    float biasv[BN / 32][4];
#pragma unroll
    for (int j = 0; j < BN / 32; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            biasv[j][r] = d.bias != nullptr ? d.bias[wn * (BN / 2) + j * 16 + fg * 4 + r] : 0.f;
            asm volatile("" ::"v"(biasv[j][r]));
        }

    // ---- the phase's weights: stage s = 4 * chunk + tap is one DMA round (64 rows x 4 units = 256 lanes) ----
    {
        const int lrow = tid >> 2;
        const int qb = (tid & 3) ^ ((-(lrow >> 2)) & 3);       // swizzled SOURCE unit (the LDS slot is lane-linear)
        const unsigned char* wrow = Wb + ((int64_t)phase * d.N + lrow) * (int64_t)d.Kp * 2 + qb * 16;
        for (int c = 0; c < J; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) {                       // tap t = k of the 2 x 2 form
                const uint32_t koff = ((uint32_t)k * (uint32_t)d.IC + (uint32_t)c * 32u) * 2u;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wrow + koff),
                                                 (__attribute__((address_space(3))) void*)(wres + (c * 4 + k) * BST + 1024 * wave_u),
                                                 16, 0, 0);
            }
    }

    // ---- tile-invariant lane state: which patch pixel a DMA lane fetches, where a fragment row sits in the patch,
    //      which output pixel a tile row is ----
    int p_img[NR], p_pr[NR], p_pc[NR];
    uint32_t p_q16[NR];
    bool p_ok[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int sidx = NT * r + tid;
        const int pp = sidx >> 2;
        p_q16[r] = (uint32_t)(((sidx & 3) ^ ((pp >> 1) & 3)) * 16);
        p_img[r] = pp / g.PIMG;
        const int rem = pp - p_img[r] * g.PIMG;
        p_pr[r] = rem / g.PW;
        p_pc[r] = rem - p_pr[r] * g.PW;
        p_ok[r] = pp < g.NPP;
    }
    const int cy = d.y0[phase] - (d.DY < 0 ? d.SY : 0), cx = d.x0[phase] - (d.DX < 0 ? d.SX : 0);
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
    int o_img = 0, o_ry = 0, o_rx = 0;                         // tile row tid (< 128) -> image, grid row, grid column
    if (tid < BM) {
        const int per_img = g.R * d.GW;
        o_img = tid / per_img;
        const int rr = tid - o_img * per_img;
        o_ry = rr / d.GW;
        o_rx = rr - o_ry * d.GW;
    }
    const int sh_y1 = d.DY > 0 ? 1 : 0, sh_x1 = d.DX > 0 ? 1 : 0;

    // ---- patch issue cursor: patches are numbered tile-major (tile, chunk); patch n lives in ring slot n % NSLOT ----
    const int t_first = widx * tiles_per_wg;
    const int n_patches = tiles_per_wg * J;
    const unsigned char* a_cur[NR];
    uint32_t a_live = 0;
    auto patch_sources = [&](int tile) {
        const int m0 = tile * BM;
        const int b0 = m0 / GHW;
        const int gy0 = (m0 - b0 * GHW) / d.GW;
        a_live = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int b = b0 + p_img[r];
            const int iy = (gy0 + p_pr[r]) * d.SY + cy;
            const int ix = p_pc[r] * d.SX + cx;
            const bool ok = p_ok[r] && b < d.B && (unsigned)iy < (unsigned)d.IH && (unsigned)ix < (unsigned)d.IW;
            a_cur[r] = ok ? Xb + ((uint32_t)((b * d.IH + iy) * d.IW + ix) * pix_bytes + p_q16[r]) : Zp;
            a_live |= ok ? (1u << r) : 0u;
        }
    };
    int issued = 0, is_tile = t_first, is_c = 0;
    auto issue_patch = [&]() {
        unsigned char* dst = pring + (issued % NSLOT) * PBUF;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
#ifndef VG_ABL_NO_A
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_cur[r],
                                             (__attribute__((address_space(3))) void*)(dst + (NT * r + 64 * wave_u) * 16),
                                             16, 0, 0);
#endif
            a_cur[r] += (a_live >> r & 1u) * 64u;
        }
        ++issued;
        if (++is_c == J) { is_c = 0; ++is_tile; if (issued < n_patches) patch_sources(is_tile); }
    };

    f32x4 acc[TM][TN];
    // one chunk = 4 taps x (TM + TN fragment reads, TM x TN MFMAs); the reads of tap k+1 are issued before the MFMAs of
    // tap k (two fragment sets): with ONE wave per SIMD nothing else hides the LDS latency
    auto load_frags = [&](const unsigned char* sa, const unsigned char* sb, int k, u32x4 (&fa)[TM], u32x4 (&fb)[TN]) {
        const int a = k >> 1, b = k & 1;
        const int shy = a ? sh_y1 : 1 - sh_y1, shx = b ? sh_x1 : 1 - sh_x1;
        const int tapoff = shy * g.PW + shx;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int pp = ppbase[i] + tapoff;
            fa[i] = *reinterpret_cast<const u32x4*>(sa + pp * 64 + ((fg ^ ((pp >> 1) & 3)) << 4));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int r = wn * WNC + j * 16 + fr;
            fb[j] = *reinterpret_cast<const u32x4*>(sb + r * 64 + ((fg ^ ((-(r >> 2)) & 3)) << 4));
        }
    };
    auto mfma_frags = [&](const u32x4 (&fa)[TM], const u32x4 (&fb)[TN]) {
#ifndef VG_ABLATE_COMPUTE
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[j]),
                                                                   __builtin_bit_cast(bf16x8, fa[i]), acc[i][j], 0, 0, 0);
#endif
    };
    auto compute_chunk = [&](const unsigned char* sa, const unsigned char* sw) {
        u32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
        load_frags(sa, sw, 0, fa0, fb0);
        load_frags(sa, sw + BST, 1, fa1, fb1);
        mfma_frags(fa0, fb0);
        load_frags(sa, sw + 2 * BST, 2, fa0, fb0);
        mfma_frags(fa1, fb1);
        load_frags(sa, sw + 3 * BST, 3, fa1, fb1);
        mfma_frags(fa0, fb0);
        mfma_frags(fa1, fb1);
#if !defined(VG_ABLATE_COMPUTE) && !defined(VG_NO_SCHED)
        // the order the scheduler has to keep (0x100 = DS read, 0x008 = MFMA): tap 0's fragments, then every tap's
        // MFMAs with the next tap's TM + TN reads spread between them -- left alone hipcc re-serialises this into
        // "two reads, s_waitcnt lgkmcnt(0), two MFMAs" and the chunk takes 1900 cycles instead of ~700
        __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int h = 0; h < TM + TN; ++h) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
#endif
    };

    float st1[TN][4], st2[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            st1[j][r] = 0.f;
            st2[j][r] = 0.f;
        }
    unsigned char* Yb = reinterpret_cast<unsigned char*>(d.Y);
    const int oc_bytes = d.OC * 2;

    // ---- prologue: NSLOT-1 patches in flight ----
    patch_sources(t_first);
    for (int p = 0; p < NSLOT - 1 && p < n_patches; ++p) issue_patch();

#ifdef VG_GS_STAMPS
    unsigned long long tacc[5] = {0, 0, 0, 0, 0}, tprev;
#define GS_STAMP(i) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                      __builtin_amdgcn_sched_barrier(0); tacc[i] += t_ - tprev; tprev = t_; }
    { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev) :: "memory"); }
#else
#define GS_STAMP(i)
#endif
    int q = 0;                                                  // patches consumed
    for (int tl = 0; tl < tiles_per_wg; ++tl) {
        const int tile = t_first + tl;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < J; ++c, ++q) {
            // patch q has landed once at most the patches issued after it are outstanding (all DMA of a wave completes
            // in order; the weights and every older patch are older still; epilogue stores of earlier tiles are
            // YOUNGER than those patches, so counting patches only over-waits by a few operations: safe)
            GS_STAMP(4)
            gs_wait_patches<NR>(issued - 1 - q);
            __builtin_amdgcn_s_barrier();                       // ... for every wave; all waves are done with slot q-1
            GS_STAMP(0)
            if (c == 0 && tid < BM) {                           // (the previous tile's stores have read the old table)
                const int m0 = tile * BM;
                const int b0 = m0 / GHW;
                const int gy0 = (m0 - b0 * GHW) / d.GW;
                const int oy = (gy0 + o_ry) * d.OSY + d.ooy[phase];
                const int ox = o_rx * d.OSX + d.oox[phase];
                opix_tab[tid] = ((b0 + o_img) * d.OH + oy) * d.OW + ox;
            }
            if (issued < n_patches) issue_patch();              // into the slot patch q-1 has just left
            GS_STAMP(1)
            const unsigned char* sa = pring + (q % NSLOT) * PBUF;
            compute_chunk(sa, wres + c * 4 * BST);
            GS_STAMP(2)
        }

        // ---------------- epilogue of the tile (the ring keeps filling underneath) ----------------
#ifdef VG_ABL_NO_EPI
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[i][j]));
        continue;
#endif
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                uint32_t pk[2];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[i][j][r] + biasv[j][r];
                    if (d.act != VG_ACT_NONE) v = act_fwd(v, d.act, d.act_slope);
                    if (d.stats != nullptr) {
                        st1[j][r] += v;
                        st2[j][r] += v * v;
                    }
                    const uint32_t h = (uint32_t)ElemT<VG_BF16>::from_f32(v);
                    pk[r >> 1] = (r & 1) ? (pk[r >> 1] | (h << 16)) : h;
                }
                const int row = wm * 64 + i * 16 + fr;
                const int col = wn * WNC + j * 16 + fg * 4;
                *reinterpret_cast<uint2*>(cbuf + row * CPITCH + col * 2) = uint2{pk[0], pk[1]};
            }
        gs_lds_barrier();                                       // C tile and output-pixel table visible
#pragma unroll
        for (int u = tid; u < BM * SEGS; u += NT) {
            const int row = u / SEGS, seg = u - row * SEGS;
            const int op = opix_tab[row];
            *reinterpret_cast<u32x4*>(Yb + (int64_t)op * oc_bytes + seg * 16) =
                *reinterpret_cast<const u32x4*>(cbuf + row * CPITCH + seg * 16);
        }
    }

#ifdef VG_GS_STAMPS
    GS_STAMP(3)
    if (tid == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(d.Y) + (int64_t)blockIdx.x * 8;
        for (int i = 0; i < 5; ++i) dbg[i] = tacc[i];
    }
#endif
    // ---- BatchNorm partial sums of all this workgroup's tiles: one slab row ----
    if (d.stats != nullptr) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = st1[j][r], b = st2[j][r];              // this lane: pixels fr (mod 16) of all its tiles
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
                if (fr == 0) {
                    const int cidx = wn * WNC + j * 16 + fg * 4 + r;
                    red[(wm * BN + cidx) * 2 + 0] = a;
                    red[(wm * BN + cidx) * 2 + 1] = b;
                }
            }
        gs_lds_barrier();
        if (tid < BN) {
            const float a = red[tid * 2] + red[(BN + tid) * 2];
            const float b = red[tid * 2 + 1] + red[(BN + tid) * 2 + 1];
            const int64_t part = (int64_t)blockIdx.x;
            d.stats[(part * 2 + 0) * d.N + tid] = a;
            d.stats[(part * 2 + 1) * d.N + tid] = b;
        }
    }
}

// host: does the descriptor have the stationary form, and how is it cut over the workgroups?
inline bool stationary_plan(const vg_gg_desc* d, int dtype, PatchGeo* g, StatPlan* sp) {
    if (dtype != VG_BF16 || !stationary_enabled() || !use_patch() || !use_dma() || d->zeros == nullptr) return false;
    const bool transposed = d->TH == 2 && d->TW == 2 && d->SY == 1 && d->SX == 1 && (d->DY == 1 || d->DY == -1) &&
                            (d->DX == 1 || d->DX == -1);
    if (!transposed || d->mask_x != nullptr) return false;
    if (d->N != 64 || d->OC < d->N || d->IC % 32 != 0 || d->IC > 128 || d->Kp != 4 * d->IC) return false;
    if (!patch_geometry(d, GS_BM, g) || g->ncy != 1 || g->ncx != 1) return false;
    for (int p = 0; p < d->nphase; ++p)                        // every tile row writes an output pixel (no skips)
        if ((d->GH - 1) * d->OSY + d->ooy[p] >= d->OH || (d->GW - 1) * d->OSX + d->oox[p] >= d->OW) return false;
    const int T = (int)(((int64_t)d->B * d->GH * d->GW) / GS_BM);      // tiles per phase
    int W = 256 / d->nphase;                                             // one workgroup per CU
    if (W < 1) W = 1;
    while (W > 1 && T % W != 0) --W;
    if (T / W < 4) return false;                                        // too few tiles to amortise the weight load
    sp->wgs_per_phase = W;
    sp->tiles_per_wg = T / W;
    sp->nr = g->NPP <= 192 ? 3 : 4;
    return true;
}

inline int launch_stationary(const vg_gg_desc* d, const PatchGeo& g, const StatPlan& sp, hipStream_t s) {
    dim3 grid((unsigned)(sp.wgs_per_phase * d->nphase)), block(GS_NT);
    if (sp.nr == 3) vg_launch_timed(0, (ggs_kernel<64, 16, 3, 6>), grid, block, 0, s, *d, g, sp.wgs_per_phase, sp.tiles_per_wg);
    else vg_launch_timed(0, (ggs_kernel<64, 16, 4, 4>), grid, block, 0, s, *d, g, sp.wgs_per_phase, sp.tiles_per_wg);
    return VG_LAUNCH_RC();
}

// Register-resident weights: persistent gather-GEMM for the short-K, 64-channel transposed forms -- one phase of a
// k4 s2 p1 ConvTranspose2d forward / Conv2d data gradient with N = 64 output channels and K = 4 taps x 128 channels
// (the Generator's ConvTranspose2d(128 -> 64) at gan_code.py:42, the data gradient of the Discriminator's
// Conv2d(64 -> 128) at gan_code.py:66).  Included by conv_gemm.hip (inside its anonymous namespace); same descriptor,
// same packed weights, same patch geometry (conv_patch.hpp) and the same accumulation order as ggp_kernel / ggs_kernel.
//
// Why.  These launches are the least efficient MFMA work of the step: per 128 x 64 tile ggp_kernel
// streams 64 KB of weights behind 48 KB of input patch through a chain of 8 barrier stages that each wait out one
// LDS-DMA round trip, and its epilogue runs with nothing in flight.  ggs_kernel (weights resident in LDS) removed the
// weight stream but its 64 KB of weights left room for ONE 4-wave workgroup per CU, so nothing ran under its epilogues.
// Here a wave keeps ITS slice of the phase's weights -- 32 output channels x K = 512: 32 fragments = 128 registers --
// in registers for the whole launch:
//   * LDS holds only the patch ring (NSLOT x 12 | 16 KB) and one 16.5 KB tile buffer: TWO workgroups per CU, which
//     run each other's epilogues under their main loops;
//   * the only ingest is the input patch (48 KB per tile instead of 112): NSLOT-1 chunks in flight per workgroup behind
//     counted s_waitcnt vmcnt, one raw s_barrier per 32-channel chunk;
//   * fragment reads drop by a third (no weight fragments): 4 ds_read_b128 per 8 MFMAs;
//   * BatchNorm partial sums of all tiles of a workgroup stay in registers: one slab row per workgroup;
//   * the activation-backward mask of a data-gradient launch (vg_gg_desc::mask_x) arrives by LDS-DMA into the tile
//     buffer, is applied in place when the accumulators are staged, and costs no register across the main loop.

constexpr int GR_BM = 128, GR_NT = 256;

struct RegwPlan { int wgs_per_phase, tiles_per_wg, nr; };

// OPT-IN (VG_GG_STATIONARY=2).  Measured on MI355X (S=64, B=128, tools/layer_bench.py): bit-identical to ggp_kernel
// (tests/test_gpu_kernels.py::test_register_weights_gather_gemm_equals_reference_and_patch_path) but SLOWER -- G4 forward
// 72 us against 64, D1 data gradient (2B) 39 against 29.  History: the first build needed 128 registers for the weights
// plus ~150 for accumulators, fragments, cursors and the ~40 loop-invariant addresses hipcc hoists; at the 256-register
// cap of two waves per SIMD it spilled 76-300 bytes per lane, and every scratch_load in the tile loop comes with an
// s_waitcnt vmcnt(0) that drains the patch ring (152 / 44 us).  The addresses are now recomputed behind opaque zeros
// (tz, cz) and the instantiations in use compile without scratch -- but the recomputation is ~100 VALU instructions
// per 32 MFMAs, and with ONE wave of a workgroup per SIMD nothing hides it: ablation builds (tools/ablate.sh): no DMA
// and no epilogue 38.6 us (MFMA bound 13.7), no epilogue 46.7, no DMA 64.0.  What it needs: the fragment addresses as
// ds_read immediates (a patch image whose swizzle commutes with the tap shifts) or the weights pinned in AGPRs by hand
// so that 16 address registers fit.  Not done in round 2.
// 0 (default) off, 1 = weights resident in LDS (ggs_kernel, conv_stationary.hpp), 2 = weights resident in registers
inline int stationary_mode() {
    const char* e = getenv("VG_GG_STATIONARY");
    return e ? atoi(e) : 0;
}

// tile row -> address inside the tile buffer: 128-byte rows, 16 bytes of padding after every 8 rows (the LDS-DMA of the
// mask writes 8 whole rows = 1 KB per wave instruction; the padding spreads the accumulators' 8-byte column accesses)
__device__ __forceinline__ int gr_row_off(int row) { return (row >> 3) * 1040 + (row & 7) * 128; }

template <int NR, int NSLOT, bool STATS, bool MASK>
__global__ __launch_bounds__(GR_NT, 2) void ggr_kernel(const vg_gg_desc d, const PatchGeo g, const int wgs_per_phase,
                                                       const int tiles_per_wg) {
    constexpr int BM = GR_BM, NT = GR_NT, BN = 64, TM = 4, TN = 2, WNC = 32, J = 4;
    constexpr int PBUF = NR * NT * 16;
    constexpr int OFF_C = NSLOT * PBUF, CBYTES = (BM / 8) * 1040, OFF_R = OFF_C + CBYTES, TOTAL = OFF_R + 2 * BN * 2 * 4;
    static_assert(2 * TOTAL <= 160 * 1024, "two workgroups per CU");
    static_assert(NSLOT >= 3 && NSLOT <= 7, "gs_wait_patches covers <= 5 patches in flight behind the one awaited");
    // ONE shared object (a second one next to an LDS-DMA target can make hipcc drain vmcnt before every ds_read)
    __shared__ __attribute__((aligned(16))) unsigned char smem[TOTAL];
    unsigned char* const pring = smem;
    unsigned char* const cbuf = smem + OFF_C;
    float* const red = reinterpret_cast<float*>(smem + OFF_R);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave >> 1, wn = wave & 1;
    const int phase = (int)blockIdx.x / wgs_per_phase;
    const int widx = (int)blockIdx.x - phase * wgs_per_phase;
    const int GHW = d.GH * d.GW;
    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(d.X);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(d.W);
    const unsigned char* Zp = reinterpret_cast<const unsigned char*>(d.zeros);
    const unsigned char* Mb = reinterpret_cast<const unsigned char*>(d.mask_x);
    constexpr bool masked = MASK;
    const uint32_t pix_bytes = (uint32_t)d.IC * 2u;
    const int fr = lane & 15, fg = lane >> 4;

    // ---- this wave's weights: fragment (chunk c, tap k, column tile j) = 16 channels x 32 k, the 16 bytes a lane would
    // read from ggp_kernel's LDS image (row n = wn*32 + j*16 + fr, unit fg).  Loaded ONCE, and USED before any DMA is
    // issued: a later first use would make the compiler wait for these loads with vmcnt(0) inside the tile loop.
    u32x4 wf[J][4][TN];
    {
        const unsigned char* wrow = Wb + ((int64_t)phase * d.N + wn * WNC + fr) * (int64_t)d.Kp * 2 + fg * 16;
#pragma unroll
        for (int c = 0; c < J; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    wf[c][k][j] = *reinterpret_cast<const u32x4*>(wrow + (int64_t)j * 16 * d.Kp * 2 +
                                                                   ((uint32_t)k * (uint32_t)d.IC + (uint32_t)c * 32u) * 2u);
#pragma unroll
        for (int c = 0; c < J; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(wf[c][k][j]));
    }

    // ---- tile-invariant lane state ----
    // patch DMA lane of round r: image / patch row / patch column / source unit / "inside the patch", packed in one register
    uint32_t p_st[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int sidx = NT * r + tid;
        const int pp = sidx >> 2;
        const int q = (sidx & 3) ^ ((pp >> 1) & 3);
        const int img = pp / g.PIMG;
        const int rem = pp - img * g.PIMG;
        const int pr = rem / g.PW;
        const int pc = rem - pr * g.PW;
        p_st[r] = (uint32_t)img | ((uint32_t)pr << 8) | ((uint32_t)pc << 16) | ((uint32_t)q << 24) | (pp < g.NPP ? 1u << 26 : 0u);
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
    const int sh_y1 = d.DY > 0 ? 1 : 0, sh_x1 = d.DX > 0 ? 1 : 0;

    // ---- patch issue cursor: patches are numbered tile-major (tile, chunk); patch n lives in ring slot n % NSLOT ----
    const int t_first = widx * tiles_per_wg;
    const int n_patches = tiles_per_wg * J;
    uint32_t a_off[NR];                                         // byte offset of the lane's next 16 bytes in X (rounds that read real data)
    uint32_t a_live = 0;
    auto patch_sources = [&](int tile) {
        const int m0 = tile * BM;
        const int b0 = m0 / GHW;
        const int gy0 = (m0 - b0 * GHW) / d.GW;
        a_live = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int b = b0 + (int)(p_st[r] & 255u);
            const int iy = (gy0 + (int)(p_st[r] >> 8 & 255u)) * d.SY + cy;
            const int ix = (int)(p_st[r] >> 16 & 255u) * d.SX + cx;
            const bool ok = (p_st[r] >> 26 & 1u) && b < d.B && (unsigned)iy < (unsigned)d.IH && (unsigned)ix < (unsigned)d.IW;
            a_off[r] = (uint32_t)((b * d.IH + iy) * d.IW + ix) * pix_bytes + (p_st[r] >> 24 & 3u) * 16u;
            a_live |= ok ? (1u << r) : 0u;
        }
    };
    int issued = 0, is_tile = t_first, is_c = 0;
    auto issue_patch = [&]() {
        unsigned char* dst = pring + (issued % NSLOT) * PBUF;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
#ifndef VG_ABL_NO_A
            const unsigned char* src = (a_live >> r & 1u) ? Xb + a_off[r] : Zp;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + (NT * r + 64 * wave_u) * 16),
                                             16, 0, 0);
#endif
            a_off[r] += 64u;
        }
        ++issued;
        if (++is_c == J) { is_c = 0; ++is_tile; if (issued < n_patches) patch_sources(is_tile); }
    };

    f32x4 acc[TM][TN];
    // one chunk = 4 taps x (TM fragment reads, TM x TN MFMAs against the register-resident weights); the MFMA takes the
    // WEIGHT fragment as its row operand: lane (fr, fg) then holds pixel 16 i + fr, channels 16 j + 4 fg + 0..3
    auto load_frags = [&](const unsigned char* sa, int k, int z, u32x4 (&fa)[TM]) {
        const int a = k >> 1, b = k & 1;
        const int shy = a ? sh_y1 : 1 - sh_y1, shx = b ? sh_x1 : 1 - sh_x1;
        const int tapoff = shy * g.PW + shx;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int pp = ppbase[i] + tapoff + z;              // z: an opaque 0 -- keeps the 16 fragment addresses out of registers
            fa[i] = *reinterpret_cast<const u32x4*>(sa + pp * 64 + (((lane >> 4) ^ ((pp >> 1) & 3)) << 4));
        }
    };
    auto mfma_frags = [&](const u32x4 (&fa)[TM], const u32x4 (&w)[TN]) {
#ifndef VG_ABLATE_COMPUTE
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[j]),
                                                                   __builtin_bit_cast(bf16x8, fa[i]), acc[i][j], 0, 0, 0);
#endif
    };
    // the reads of tap k+1 are issued before the MFMAs of tap k (two fragment sets): with one wave of this workgroup per
    // SIMD the LDS latency is otherwise paid four times per chunk
    auto compute_chunk = [&](const unsigned char* sa, const u32x4 (&w)[4][TN], const int z) {
        u32x4 fa0[TM], fa1[TM];
        load_frags(sa, 0, z, fa0);
        load_frags(sa, 1, z, fa1);
        mfma_frags(fa0, w[0]);
        load_frags(sa, 2, z, fa0);
        mfma_frags(fa1, w[1]);
        load_frags(sa, 3, z, fa1);
        mfma_frags(fa0, w[2]);
        mfma_frags(fa1, w[3]);
#if !defined(VG_ABLATE_COMPUTE) && !defined(VG_NO_SCHED)
        // the order the scheduler has to keep (0x100 = DS read, 0x008 = MFMA)
        __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int h = 0; h < TM; ++h) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - TM, 0);
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

    int q = 0;                                                  // patches consumed
    for (int tl = 0; tl < tiles_per_wg; ++tl) {
        const int tile = t_first + tl;
        const int m0 = tile * BM;
        const int b0 = m0 / GHW;
        const int gy0 = (m0 - b0 * GHW) / d.GW;
        const int obase = (b0 * d.OH + gy0 * d.OSY) * d.OW;     // output pixel of a tile row = obase + o_inv
        int after_mask = 0;                                     // patches issued behind this tile's mask DMA
        // tile rows of this thread in the mask DMA and in the store loop: row(it) = 32 it + (tid >> 3), segment tid & 7;
        // output pixel = obase + o_inv[it].  Recomputed per tile behind an opaque 0 (tz) so that neither these nor the
        // epilogue's LDS addresses stay in registers across the main loop (the weights leave no room for them).
        int tz;
        asm volatile("s_mov_b32 %0, 0" : "=s"(tz));
        const int tidz = tid + tz;
        int o_inv[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 32 + (tidz >> 3);
            const int per_img = g.R * d.GW;
            const int img = row / per_img;
            const int rr = row - img * per_img;
            const int ry = rr / d.GW;
            const int rx = rr - ry * d.GW;
            o_inv[it] = (img * d.OH + ry * d.OSY + d.ooy[phase]) * d.OW + rx * d.OSX + d.oox[phase];
        }
        const int seg16 = (tidz & 7) * 16;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < J; ++c, ++q) {
            // patch q has landed once at most the patches issued after it are outstanding (all DMA of a wave completes
            // in order; mask rows and epilogue stores issued in between only make this wait stricter: safe)
            gs_wait_patches<NR>(issued - 1 - q);
            __builtin_amdgcn_s_barrier();                       // ... for every wave; all waves are done with slot q-1
            if (c == 0 && masked) {
                // the activated output of the layer below at this tile's output pixels -> tile buffer (the previous
                // tile's store loop, which read the buffer, lies before the barrier above): 8 rows per wave instruction
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(Mb + (int64_t)(obase + o_inv[it]) * oc_bytes + seg16),
                        (__attribute__((address_space(3))) void*)(cbuf + (it * 4 + wave_u) * 1040), 16, 0, 0);
            }
            if (issued < n_patches) { issue_patch(); ++after_mask; }   // into the slot patch q-1 has just left
            int cz;
            asm volatile("s_mov_b32 %0, 0" : "=s"(cz));
            compute_chunk(pring + (q % NSLOT) * PBUF, wf[c], cz);
        }

        // ---------------- epilogue of the tile (the ring keeps filling underneath) ----------------
#ifdef VG_ABL_NO_EPI
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[i][j]));
        continue;
#endif
        if (masked) {                                           // this wave's mask rows have landed ... and everybody's
            gs_wait_patches<NR>(after_mask);
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wm * 64 + i * 16 + (tidz & 15);
                const int col = wn * WNC + j * 16 + (tidz >> 4 & 3) * 4;
                uint2* slot = reinterpret_cast<uint2*>(cbuf + gr_row_off(row) + col * 2);
                uint2 mx = uint2{0u, 0u};
                if (masked) mx = *slot;
                uint32_t pk[2];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[i][j][r];
                    if (d.act != VG_ACT_NONE) v = act_fwd(v, d.act, d.act_slope);
                    if (STATS) {
                        st1[j][r] += v;
                        st2[j][r] += v * v;
                    }
                    uint32_t h = (uint32_t)ElemT<VG_BF16>::from_f32(v);
                    if (masked) {                               // v * act'(x) on the ROUNDED value, rounded again (as mask_segment)
                        const uint32_t xw = (r >> 1) ? mx.y : mx.x;
                        const float xv = __uint_as_float((r & 1) ? (xw & 0xffff0000u) : (xw << 16));
                        h = (uint32_t)ElemT<VG_BF16>::from_f32(act_bwd(xv, __uint_as_float(h << 16), d.mask_act, d.mask_slope));
                    }
                    pk[r >> 1] = (r & 1) ? (pk[r >> 1] | (h << 16)) : h;
                }
                *slot = uint2{pk[0], pk[1]};
            }
        gs_lds_barrier();                                       // C tile visible
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 32 + (tidz >> 3);
            *reinterpret_cast<u32x4*>(Yb + (int64_t)(obase + o_inv[it]) * oc_bytes + seg16) =
                *reinterpret_cast<const u32x4*>(cbuf + gr_row_off(row) + seg16);
        }
    }

    // ---- BatchNorm partial sums of all this workgroup's tiles: one slab row ----
    if (STATS) {
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

// host: does the descriptor have the form, and how is it cut over the workgroups (two per CU)?
inline bool regw_plan(const vg_gg_desc* d, int dtype, PatchGeo* g, RegwPlan* rp) {
    if (dtype != VG_BF16 || stationary_mode() != 2 || !use_patch() || !use_dma() || d->zeros == nullptr) return false;
    const bool transposed = d->TH == 2 && d->TW == 2 && d->SY == 1 && d->SX == 1 && (d->DY == 1 || d->DY == -1) &&
                            (d->DX == 1 || d->DX == -1);
    if (!transposed || d->bias != nullptr || d->bnb_y != nullptr) return false;
    if (d->N != 64 || d->OC != d->N || d->IC != 128 || d->Kp != 4 * d->IC) return false;
    if (d->mask_x != nullptr && !vg_aligned16(d->mask_x)) return false;
    if (!patch_geometry(d, GR_BM, g) || g->ncy != 1 || g->ncx != 1) return false;
    for (int p = 0; p < d->nphase; ++p)                        // every tile row writes an output pixel (no skips)
        if ((d->GH - 1) * d->OSY + d->ooy[p] >= d->OH || (d->GW - 1) * d->OSX + d->oox[p] >= d->OW) return false;
    const int T = (int)(((int64_t)d->B * d->GH * d->GW) / GR_BM);      // tiles per phase
    int W = 512 / d->nphase;                                             // two workgroups per CU
    if (W < 1) W = 1;
    if (W > T / 2) W = T / 2;                                           // at least two tiles per workgroup
    if (W < 1) return false;
    while (W > 1 && T % W != 0) --W;
    if (T / W < 2) return false;                                        // too few tiles to amortise the weight load
    rp->wgs_per_phase = W;
    rp->tiles_per_wg = T / W;
    rp->nr = g->NPP <= 192 ? 3 : 4;
    // only the instantiations that compile without scratch (a spill in the tile loop drains the patch ring)
    if (rp->nr != 3 || (d->stats != nullptr && d->mask_x != nullptr)) return false;
    return true;
}

inline int launch_regw(const vg_gg_desc* d, const PatchGeo& g, const RegwPlan& rp, hipStream_t s) {
    dim3 grid((unsigned)(rp.wgs_per_phase * d->nphase)), block(GR_NT);
    const bool st = d->stats != nullptr, mk = d->mask_x != nullptr;
#define GR_LAUNCH(NR_, NS_, ST_, MK_) vg_launch_timed(0, (ggr_kernel<NR_, NS_, ST_, MK_>), grid, block, 0, s, *d, g, rp.wgs_per_phase, rp.tiles_per_wg)
    if (rp.nr == 3) {
        if (st && mk) GR_LAUNCH(3, 5, true, true); else if (st) GR_LAUNCH(3, 5, true, false);
        else if (mk) GR_LAUNCH(3, 5, false, true); else GR_LAUNCH(3, 5, false, false);
    } else {
        if (st && mk) GR_LAUNCH(4, 3, true, true); else if (st) GR_LAUNCH(4, 3, true, false);
        else if (mk) GR_LAUNCH(4, 3, false, true); else GR_LAUNCH(4, 3, false, false);
    }
#undef GR_LAUNCH
    return VG_LAUNCH_RC();
}

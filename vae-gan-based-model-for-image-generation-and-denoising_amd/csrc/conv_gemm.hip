// Gather-GEMM: implicit-GEMM convolution / transposed convolution / linear on MFMA (gfx950).
//
// Replaces the ATen kernels behind nn.Conv2d (main_vae.py:23, gan_code.py:61-84),
// nn.ConvTranspose2d (gan_code.py:21-49), nn.Linear (main_vae.py:47-48) and their data
// gradients.  Semantics: vg_gg_desc in include/vaegan_hip.h.
//
// Structure (cdna_hip_programming.md section 5, register-staged 2-buffer loop):
//   * one workgroup = 4 waves = BM x BN output tile; wave tile = (BM/WM) x (BN/WN) made of
//     16x16 MFMA tiles (v_mfma_f32_16x16x4_f32 for f32 -- an exact f32 FMA chain --,
//     v_mfma_f32_16x16x32_bf16 for bf16 storage with f32 accumulate);
//   * K is walked in 64-byte "chunks" per row (16 f32 / 32 bf16): each thread gathers one
//     16-byte unit per row-pass straight from the NHWC activation (a unit never straddles a
//     filter tap because IC*esize % 16 == 0) and zero-fills out-of-image taps, so no
//     zero-inserted or padded MACs are read from HBM;
//   * LDS image: [row][64 B] with the 16-byte slot XOR-swizzled by row (slot ^= (-(row>>2))&3)
//     -> conflict-free ds_read_b128 fragment reads and conflict-free ds_write_b128 staging;
//   * loads for chunk k+1 are issued before the MFMAs of chunk k and written to the other LDS
//     buffer after them (T14 issue-early / write-late), one barrier per chunk;
//   * epilogue: + bias, scatter to the NHWC output (phase/sub-pixel offsets), and optional
//     per-channel (sum, sum of squares) partials for train-mode BatchNorm, reduced
//     wave -> LDS -> one slab row per workgroup (deterministic, no atomics).
#include "common.hpp"
#include <type_traits>
#include <cstdlib>

namespace {

// activation backward fused into a data-gradient epilogue (vg_gg_desc::mask_x): v = dL/d(activated output) segment
// (16 bytes, storage dtype), x = the layer's activated output at the same place; returns v * act'(x), rounded again
template <int DT>
__device__ __forceinline__ u32x4 mask_segment(u32x4 v, const u32x4& x, int act, float slope) {
    if constexpr (DT == VG_BF16) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xl = __uint_as_float(x[k] << 16), xh = __uint_as_float(x[k] & 0xffff0000u);
            const float gl = __uint_as_float(v[k] << 16), gh = __uint_as_float(v[k] & 0xffff0000u);
            const uint32_t lo = ElemT<VG_BF16>::from_f32(act_bwd(xl, gl, act, slope));
            const uint32_t hi = ElemT<VG_BF16>::from_f32(act_bwd(xh, gh, act, slope));
            v[k] = lo | (hi << 16);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __float_as_uint(act_bwd(__uint_as_float(x[k]), __uint_as_float(v[k]), act, slope));
    }
    return v;
}

#define VG_WAITCNT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

// wait until at most n vector-memory operations of this wave are outstanding (n = DMA instructions that may stay
// in flight); the immediate must be a literal, so the few values that occur are enumerated (wave-uniform switch)
__device__ __forceinline__ void wait_vmcnt_le(int n) {
    switch (n) {
        case 2: VG_WAITCNT_VM(2); break;
        case 3: VG_WAITCNT_VM(3); break;
        case 4: VG_WAITCNT_VM(4); break;
        case 6: VG_WAITCNT_VM(6); break;
        case 8: VG_WAITCNT_VM(8); break;
        case 9: VG_WAITCNT_VM(9); break;
        case 12: VG_WAITCNT_VM(12); break;
        case 16: VG_WAITCNT_VM(16); break;
        case 18: VG_WAITCNT_VM(18); break;
        case 24: VG_WAITCNT_VM(24); break;
        default: VG_WAITCNT_VM(0); break;
    }
}

// LDS-DMA ring shape (measured, S=64 B=128 layer sweeps).  First sweep (one chunk per stage, 3-6 slots) favoured
// 1 chunk x 3 slots for the 128x128 tile over 2 x 3 (96 KB, one workgroup per CU).  What that sweep missed: TWO
// chunks per stage in only TWO slots -- half the barriers per FLOP, 64 KB (128x128) / 48 KB (128x64) of LDS, i.e.
// still 2-3 workgroups per CU: G1 fprop 589 -> 661, G2 703 -> 761 / 595 -> 672, G4 fprop 380 -> 454 TFLOP/s.
// The 64x64 tile (4 MFMAs per chunk and wave) wants FOUR chunks per stage in two slots (64 KB): D3 fprop 283 -> 318,
// D3 dgrad 427 -> 475, G1 dgrad 500 -> 583 (2 chunks x 3 slots: 282; 2 x 2: 239; 4 x 3: one workgroup per CU, slower).
// VG_DMA_KCH / VG_DMA_NBUF override both for sweeps.
template <int BM, int BN>
struct DmaRing {
#if defined(VG_DMA_KCH) && defined(VG_DMA_NBUF)
    static constexpr int KCH = VG_DMA_KCH, NBUF = VG_DMA_NBUF;
#else
#ifndef VG_RING64_KCH
#define VG_RING64_KCH 4
#endif
#ifndef VG_RING64_NBUF
#define VG_RING64_NBUF 2
#endif
#ifndef VG_RING128x64_KCH
#define VG_RING128x64_KCH 2
#endif
    static constexpr int KCH = (BM * BN >= 128 * 128) ? 2 : (BM * BN >= 128 * 64) ? VG_RING128x64_KCH : VG_RING64_KCH;
    static constexpr int NBUF = (BM * BN >= 128 * 64) ? 2 : VG_RING64_NBUF;
#endif
};

constexpr int GG_N_MAJOR = 1 << 16;     // ksplit_arg flag of a non-split launch: XCD-major over n tiles

template <int DT, int BM, int BN, int WM, int WN, bool SPLITK, bool DMA>
__global__ __launch_bounds__(256) void gg_kernel(const vg_gg_desc d, const int ksplit_arg, const int stages_per_split) {
    const int ksplit = SPLITK ? ksplit_arg : 1;        // compile-time 1 on the common path (keeps its registers lean)
    typedef ElemT<DT> E;
    constexpr int ESZ = E::size;
    // output element: the operand type, except fp8 operands (VG_FP8), which produce bf16 like the bf16 engine
    constexpr int DTO = (DT == VG_FP8) ? VG_BF16 : DT;
    typedef ElemT<DTO> EO;
    constexpr int ESZO = EO::size;
    constexpr int TM = BM / WM / 16;
    constexpr int TN = BN / WN / 16;
    constexpr int AP = BM / 64;                    // A row passes per chunk
    constexpr int BP = (BN + 63) / 64;             // B row passes per chunk
    // 64-byte K chunks per barrier (bf16 MFMAs are 16x shorter than f32 ones)
    constexpr int KCH = DMA ? DmaRing<BM, BN>::KCH : ((DT == VG_F32) ? 1 : 2);    // fp8: 2 chunks = the MFMA's K of 128
    constexpr int CHB = (BM + BN) * 64;            // bytes of one chunk image
    constexpr int STAGE = CHB * KCH;               // bytes per LDS buffer
    constexpr int NBUF = DMA ? DmaRing<BM, BN>::NBUF : 2;    // DMA path: LDS ring, NBUF-1 stages in flight
    static_assert(WM * WN == 4, "4 waves");
    static_assert(BM % 64 == 0, "BM multiple of 64");
    static_assert(!DMA || BN % 64 == 0, "DMA path needs whole 16-row wave slices of the B tile");

    // ONE shared object (a second one next to an LDS-DMA target can make hipcc drain vmcnt before every ds_read):
    // [NBUF stage buffers][output-pixel table of the epilogue]
    __shared__ __attribute__((aligned(16))) unsigned char smem[NBUF * STAGE + BM * 4];
    int* const opix_tab = reinterpret_cast<int*>(smem + NBUF * STAGE);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // split-K launches: z = phase * ksplit + K slice (round 4: the sub-pixel phases of the transposed forms split too)
    const int phase = ksplit > 1 ? (int)blockIdx.z / ksplit : (int)blockIdx.z;
    const int kz = ksplit > 1 ? (int)blockIdx.z - phase * ksplit : 0;
    // XCD-aware tile order (cdna_hip_programming.md T1): workgroups are dealt round-robin over the 8 XCDs, each
    // with a private L2.  The launch is 1-D in (m tile, n tile); id -> (xcd = id % 8, slot = id / 8) and the slots
    // of one XCD walk ALL n tiles of an m tile before moving on, so the gathered A rows of an m tile are fetched
    // into one XCD's L2 once and re-used by its other n tiles (speed only; any placement is correct).
    int bx, by;
    {
        const int n_tiles = (d.N + BN - 1) / BN;
        const int id = blockIdx.x;
        const int xcd = id & 7, slot = id >> 3;
        if (!SPLITK && (ksplit_arg & GG_N_MAJOR)) {
            // weights larger than the input (n_major() below): an XCD owns n tiles (n % 8 == xcd) and walks all m
            // tiles of one before the next, so it streams 1/8 of the WEIGHTS and all of the (smaller) input
            const int mt = (int)gridDim.x / n_tiles;
            bx = slot % mt;
            by = (slot / mt) * 8 + xcd;
        } else {
            by = slot % n_tiles;
            bx = (slot / n_tiles) * 8 + xcd;
        }
    }
    const int m_tiles_ = (d.B * d.GH * d.GW + BM - 1) / BM;
    if (bx >= m_tiles_) return;                             // padding blocks of the last group of 8 m tiles
    const int m0 = bx * BM;
    const int n0 = by * BN;
    const int M = d.B * d.GH * d.GW;
    const int GHW = d.GH * d.GW;

    const int lrow = tid >> 2;                     // 0..63
    // 16-byte unit within the chunk.  Register path: the LDS slot is swizzled when written.  DMA path: a lane's
    // LDS slot is fixed (base + lane*16), so the swizzle moves to the SOURCE unit it fetches (same involution).
    const int q = DMA ? ((tid & 3) ^ ((-(lrow >> 2)) & 3)) : (tid & 3);
    const int upt = (d.IC * ESZ) >> 4;             // 16-byte units per tap
    const int ntap = d.TH * d.TW;
    const int nchunks = (d.Kp * ESZ) >> 6;
    const int nstages_all = (nchunks + KCH - 1) / KCH;
    const int s_begin = kz * stages_per_split;
    const int nstages = min(nstages_all, s_begin + stages_per_split) - s_begin;     // stages of THIS K slice
    const uint32_t pix_bytes = (uint32_t)d.IC * ESZ;

    // ---- per-thread gather bases for its A rows: pixel index of tap (0,0) and its (y, x) for the bounds test ----
    int a_pix[AP], a_iy[AP], a_ix[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = m0 + lrow + 64 * i;
        if (m < M) {
            const int b = m / GHW;
            const int r = m - b * GHW;
            const int gy = r / d.GW;
            const int gx = r - gy * d.GW;
            a_iy[i] = gy * d.SY + d.y0[phase];
            a_ix[i] = gx * d.SX + d.x0[phase];
            a_pix[i] = (b * d.IH + a_iy[i]) * d.IW + a_ix[i];
        } else {
            a_iy[i] = -(1 << 28);                  // fails every bounds test -> zeros
            a_ix[i] = 0;
            a_pix[i] = 0;
        }
    }
    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(d.X);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(d.W);
    const uint32_t wrow_bytes = (uint32_t)d.Kp * ESZ;
    const unsigned char* b_ptr[BP];
    bool b_ok[BP];
#pragma unroll
    for (int j = 0; j < BP; ++j) {
        const int n = n0 + lrow + 64 * j;
        b_ok[j] = (lrow + 64 * j < BN) && (n < d.N);
        b_ptr[j] = Wb + ((int64_t)phase * d.N + (b_ok[j] ? n : 0)) * wrow_bytes + q * 16;
    }

    // running K position of the NEXT chunk this thread loads: tap (ta, tb), 16-byte unit cu inside the tap
    const int u0 = s_begin * KCH * 4 + q;          // first 16-byte unit of this thread in this K slice
    int ld_t = u0 / upt, ld_cu = u0 - ld_t * upt;
    int ld_ta = ld_t / d.TW, ld_tb = ld_t - ld_ta * d.TW;
    uint32_t ld_koff = (uint32_t)s_begin * KCH * 64;   // byte offset of the next chunk inside a packed weight row

    // Two register sets: the loads of stages s+1 AND s+2 are in flight while stage s is multiplied (each stage
    // otherwise costs one full L2 round trip: the MFMAs of a stage are far shorter than the memory latency).
    u32x4 ra[2][KCH][AP], rb[2][KCH][BP];
    auto load_stage = [&](auto SET) {
        constexpr int rs = decltype(SET)::value;
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            const bool uok = ld_t < ntap && ld_koff < wrow_bytes;
            const int dy = d.DY * ld_ta, dx = d.DX * ld_tb;
            const int dpix = dy * d.IW + dx;
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const bool ok = uok && (unsigned)(a_iy[i] + dy) < (unsigned)d.IH &&
                                (unsigned)(a_ix[i] + dx) < (unsigned)d.IW;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (ok) v = *reinterpret_cast<const u32x4*>(Xb + (uint32_t)(a_pix[i] + dpix) * pix_bytes + (uint32_t)ld_cu * 16u);
                ra[rs][c][i] = v;
            }
#pragma unroll
            for (int j = 0; j < BP; ++j) {
                u32x4 v = {0u, 0u, 0u, 0u};
                if (b_ok[j] && ld_koff < wrow_bytes) v = *reinterpret_cast<const u32x4*>(b_ptr[j] + ld_koff);
                rb[rs][c][j] = v;
            }
            // advance by one chunk = 4 units
            ld_koff += 64;
            ld_cu += 4;
            while (ld_cu >= upt) {
                ld_cu -= upt;
                ++ld_t;
                if (++ld_tb == d.TW) { ld_tb = 0; ++ld_ta; }
            }
        }
    };
    auto store_stage = [&](int buf, auto SET) {
        constexpr int rs = decltype(SET)::value;
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            unsigned char* sa = smem + buf * STAGE + c * CHB;
            unsigned char* sb = sa + BM * 64;
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const int r = lrow + 64 * i;
                *reinterpret_cast<u32x4*>(sa + r * 64 + ((q ^ ((-(r >> 2)) & 3)) << 4)) = ra[rs][c][i];
            }
#pragma unroll
            for (int j = 0; j < BP; ++j) {
                const int r = lrow + 64 * j;
                if (r < BN) *reinterpret_cast<u32x4*>(sb + r * 64 + ((q ^ ((-(r >> 2)) & 3)) << 4)) = rb[rs][c][j];
            }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15;       // row within the 16x16 operand tile
    const int fg = lane >> 4;       // 16-byte unit (k group)

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    // (an explicit ds_read / MFMA scheduling pipeline -- __builtin_amdgcn_sched_group_barrier, next chunk's reads
    // between this chunk's MFMAs -- was measured here and changes nothing: the loop waits on DMA arrival, not on LDS)
    auto compute = [&](int buf) {
#ifdef VG_ABLATE_COMPUTE
        return;
#endif
        if constexpr (DT == VG_FP8) {
            // v_mfma_scale_f32_16x16x128_f8f6f4: lane (row|col = lane & 15, k group = lane >> 4) holds 32 e4m3 values.
            // Any assignment of the 128 k indices to (lane group, byte) is valid as long as A and B use the same one
            // (checked with exact integer data, tools/probes/mfma_fp8_probe.hip): lane group fg takes 16-byte unit fg
            // of BOTH 64-byte chunks of the stage -- the fragment reads of the bf16 path, concatenated.
            typedef __attribute__((ext_vector_type(8))) int v8i;
            static_assert(DT != VG_FP8 || KCH % 2 == 0, "fp8: whole K=128 steps per stage");
#pragma unroll
            for (int c0 = 0; c0 < KCH; c0 += 2) {
            u32x4 fa[2][TM], fb[2][TN];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const unsigned char* sa = smem + buf * STAGE + (c0 + c) * CHB;
                const unsigned char* sb = sa + BM * 64;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int r = wm * (BM / WM) + i * 16 + fr;
                    fa[c][i] = *reinterpret_cast<const u32x4*>(sa + r * 64 + ((fg ^ ((-(r >> 2)) & 3)) << 4));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int r = wn * (BN / WN) + j * 16 + fr;
                    fb[c][j] = *reinterpret_cast<const u32x4*>(sb + r * 64 + ((fg ^ ((-(r >> 2)) & 3)) << 4));
                }
            }
            constexpr int SA = 127 * 0x01010101;                          // E8M0 1.0 for the activations
            constexpr int SB = (127 - VG_FP8_WSHIFT) * 0x01010101;        // 2^-shift: weights are stored * 2^shift
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const v8i av = {(int)fa[0][i][0], (int)fa[0][i][1], (int)fa[0][i][2], (int)fa[0][i][3],
                                (int)fa[1][i][0], (int)fa[1][i][1], (int)fa[1][i][2], (int)fa[1][i][3]};
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const v8i bv = {(int)fb[0][j][0], (int)fb[0][j][1], (int)fb[0][j][2], (int)fb[0][j][3],
                                    (int)fb[1][j][0], (int)fb[1][j][1], (int)fb[1][j][2], (int)fb[1][j][3]};
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc[i][j], 0, 0, 0, SA, 0, SB);
                }
            }
            }
            return;
        }
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            const unsigned char* sa = smem + buf * STAGE + c * CHB;
            const unsigned char* sb = sa + BM * 64;
            u32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm * (BM / WM) + i * 16 + fr;
                fa[i] = *reinterpret_cast<const u32x4*>(sa + r * 64 + ((fg ^ ((-(r >> 2)) & 3)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn * (BN / WN) + j * 16 + fr;
                fb[j] = *reinterpret_cast<const u32x4*>(sb + r * 64 + ((fg ^ ((-(r >> 2)) & 3)) << 4));
            }
#ifdef VG_ABLATE_MFMA
#pragma unroll
            for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa[i]));
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fb[j]));
            continue;
#endif
            if constexpr (DT == VG_F32) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                __uint_as_float(fa[i][kk]), __uint_as_float(fb[j][kk]), acc[i][j], 0, 0, 0);
            } else if constexpr (DT == VG_BF16) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
            }
        }
    };

    if constexpr (DMA) {
        // ---- LDS-DMA main loop: global_load_lds_dwordx4 straight into a 3-deep LDS ring -------------------------
        // No staging registers and no ds_write (the 128x128 bf16 tile was LDS-bound on its ds_write_b128 traffic).
        // Out-of-image taps / rows beyond M or N / the K tail fetch from a 64-byte zero page instead.
        // Ordering (cdna_hip_programming.md "Pipelining across barriers"): counted s_waitcnt vmcnt(L) leaves the
        // next stage in flight, then a raw s_barrier; the buffer refilled at iteration s was last read at s-1.
        constexpr int LDMA = KCH * (AP + BP);                         // DMA instructions per wave per stage
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        // zero page addressed relative to the same base as the data, so a lane's source is ONE selected offset
        // (a pointer select makes hipcc emit two exec-masked DMA instructions per load)
        const int64_t zoffA = reinterpret_cast<const unsigned char*>(d.zeros) - Xb;
        const int64_t zoffB = reinterpret_cast<const unsigned char*>(d.zeros) - Wb;
        auto issue_stage = [&](int buf) {
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                const bool uok = ld_t < ntap && ld_koff < wrow_bytes;
                const int dy = d.DY * ld_ta, dx = d.DX * ld_tb;
                const int dpix = dy * d.IW + dx;
                unsigned char* sa = smem + buf * STAGE + c * CHB;
                unsigned char* sb = sa + BM * 64;
#pragma unroll
                for (int i = 0; i < AP; ++i) {
                    const bool ok = uok && (unsigned)(a_iy[i] + dy) < (unsigned)d.IH &&
                                    (unsigned)(a_ix[i] + dx) < (unsigned)d.IW;
                    const int64_t off = ok ? (int64_t)((uint32_t)(a_pix[i] + dpix) * pix_bytes + (uint32_t)ld_cu * 16u) : zoffA;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Xb + off),
                                                     (__attribute__((address_space(3))) void*)(sa + (64 * i + 16 * wave_u) * 64),
                                                     16, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < BP; ++j) {
                    const int64_t off = (b_ok[j] && ld_koff < wrow_bytes) ? (int64_t)(b_ptr[j] - Wb) + ld_koff : zoffB;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wb + off),
                                                     (__attribute__((address_space(3))) void*)(sb + (64 * j + 16 * wave_u) * 64),
                                                     16, 0, 0);
                }
                ld_koff += 64;
                ld_cu += 4;
                while (ld_cu >= upt) {
                    ld_cu -= upt;
                    ++ld_t;
                    if (++ld_tb == d.TW) { ld_tb = 0; ++ld_ta; }
                }
            }
        };
        // Fast path (every heavy layer: >= 32 bf16 channels, i.e. whole 64-byte chunks per filter tap): the pixel, its
        // bounds test and the 64-bit source pointer are computed ONCE PER TAP; inside a tap each lane's pointer just
        // advances by 64 bytes per chunk (0 for lanes parked on the zero page).  The ablated build showed the
        // per-stage address generation, not the MFMAs, was the critical path of this kernel.
        // (round 4: also for a K slice that starts inside the operand -- split-K launches used to fall back to the
        // per-stage address generation AND to register staging; the slice's first tap is entered mid-way: tap_chunk0)
        const bool fast = (upt % 4) == 0;
        const int nct = upt >> 2;                                      // chunks per tap
        int tap_chunk = fast ? (ld_cu >> 2) : 0;                       // wave-uniform (ld_cu - q is a multiple of 4)
        bool tap_entered = false;                                      // the first tap of the slice still needs its addresses
        const unsigned char* a_cur[AP];
        uint32_t a_stride[AP];
        const unsigned char* b_cur[BP];
        uint32_t b_stride[BP];
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            b_cur[j] = b_ok[j] ? b_ptr[j] + ld_koff : reinterpret_cast<const unsigned char*>(d.zeros);
            b_stride[j] = b_ok[j] ? 64u : 0u;
        }
        auto issue_stage_fast = [&](int buf) {
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                unsigned char* sa = smem + buf * STAGE + c * CHB;
                unsigned char* sb = sa + BM * 64;
                if (tap_chunk == 0 || !tap_entered) {                  // new filter tap: the only expensive part
                    tap_entered = true;
                    const int dy = d.DY * ld_ta, dx = d.DX * ld_tb;
                    const int dpix = dy * d.IW + dx;
                    const bool tap_ok = ld_t < ntap;
#pragma unroll
                    for (int i = 0; i < AP; ++i) {
                        const bool ok = tap_ok && (unsigned)(a_iy[i] + dy) < (unsigned)d.IH &&
                                        (unsigned)(a_ix[i] + dx) < (unsigned)d.IW;
                        a_cur[i] = ok ? Xb + ((uint32_t)(a_pix[i] + dpix) * pix_bytes + (uint32_t)q * 16u + (uint32_t)tap_chunk * 64u)
                                      : reinterpret_cast<const unsigned char*>(d.zeros);
                        a_stride[i] = ok ? 64u : 0u;
                    }
                }
                const bool kin = ld_koff < wrow_bytes;                 // wave-uniform K tail (odd chunk count, KCH = 2)
#pragma unroll
                for (int i = 0; i < AP; ++i) {
                    const unsigned char* g = kin ? a_cur[i] : reinterpret_cast<const unsigned char*>(d.zeros);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)(sa + (64 * i + 16 * wave_u) * 64),
                                                     16, 0, 0);
                    a_cur[i] += a_stride[i];
                }
#pragma unroll
                for (int j = 0; j < BP; ++j) {
                    const unsigned char* g = kin ? b_cur[j] : reinterpret_cast<const unsigned char*>(d.zeros);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)(sb + (64 * j + 16 * wave_u) * 64),
                                                     16, 0, 0);
                    b_cur[j] += b_stride[j];
                }
                ld_koff += 64;
                if (++tap_chunk == nct) {
                    tap_chunk = 0;
                    ++ld_t;
                    if (++ld_tb == d.TW) { ld_tb = 0; ++ld_ta; }
                }
            }
        };
        auto issue = [&](int buf) {
            if (fast) issue_stage_fast(buf);
            else issue_stage(buf);
        };
#pragma unroll
        for (int p = 0; p < NBUF - 1; ++p)
            if (p < nstages) issue(p);
        int rb = 0, wb = NBUF - 1;                                     // ring: read s % NBUF, write (s + NBUF-1) % NBUF
        for (int ks = 0; ks < nstages; ++ks) {
            // stages ks+1 .. ks+NBUF-2 may stay in flight; stage ks must have landed
            const int ahead = min(NBUF - 2, nstages - 1 - ks);
            wait_vmcnt_le(ahead * LDMA);
            __builtin_amdgcn_s_barrier();
#ifndef VG_ABLATE_LOAD
            if (ks + NBUF - 1 < nstages) issue(wb);
#endif
            compute(rb);
            rb = rb == NBUF - 1 ? 0 : rb + 1;
            wb = wb == NBUF - 1 ? 0 : wb + 1;
        }
    } else {
    // prologue: stage 0 -> LDS buffer 0; stages 1 and 2 in flight in register sets 1 and 0
        load_stage(S0{});
        store_stage(0, S0{});
        if (nstages > 1) load_stage(S1{});
        if (nstages > 2) load_stage(S0{});
        __syncthreads();
        // steady state, unrolled by two so that register-set indices are compile-time constants:
        //   compute(s) ; store(s+1) from its set ; refill that set with stage s+3 ; barrier
        for (int ks = 0; ks < nstages; ks += 2) {
            compute(0);
            if (ks + 1 < nstages) store_stage(1, S1{});
            if (ks + 3 < nstages) load_stage(S1{});
            __syncthreads();
            if (ks + 1 >= nstages) break;
            compute(1);
            if (ks + 2 < nstages) store_stage(0, S0{});
            if (ks + 4 < nstages) load_stage(S0{});
            __syncthreads();
        }
    }

    // ---------------- epilogue ----------------
    if constexpr (SPLITK) {
        // split-K: raw f32 partial tile -> workspace slab [kz][Mpad][Npad]; bias/convert happen in the reduce kernel
        const int Npad = ((d.N + BN - 1) / BN) * BN;
        float* slab = d.ws + ((int64_t)(phase * ksplit + kz) * m_tiles_ * BM + m0) * Npad + n0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm * (BM / WM) + i * 16 + fg * 4 + r;
                    const int col = wn * (BN / WN) + j * 16 + fr;
                    slab[(int64_t)row * Npad + col] = acc[i][j][r];
                }
        return;
    }
    // (1) BatchNorm partial statistics straight from the accumulators (valid rows / real channels only);
    // (2) the C tile goes through LDS so that every lane writes one 16-byte run of channels of one output pixel
    //     (NHWC) instead of 2/4-byte scatters; output pixel offsets come from an LDS table filled once per
    //     workgroup (no per-element integer division for the sub-pixel scatter).
    const bool flat = (d.nphase == 1 && d.OSY == 1 && d.OSX == 1 && d.GH == d.OH && d.GW == d.OW);
    for (int r = tid; r < BM; r += 256) {
        const int m = m0 + r;
        int op = -1;
        if (m < M) {
            if (flat) {
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
    int ncol[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        ncol[j] = n0 + wn * (BN / WN) + j * 16 + fr;
        biasv[j] = (d.bias != nullptr && ncol[j] < d.N) ? d.bias[ncol[j]] : 0.f;
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
    __syncthreads();                                   // opix_tab visible; main-loop LDS reads are done

    if (d.stats != nullptr) {
        float* red = reinterpret_cast<float*>(smem);   // [WM][BN][2]
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm * (BM / WM) + i * 16 + fg * 4 + r;
                    if (opix_tab[row] >= 0) {
                        const float v = acc[i][j][r];
                        a += v;
                        b += v * v;
                    }
                }
            a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
            b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
            if (fg == 0) {
                const int c = wn * (BN / WN) + j * 16 + fr;
                red[(wm * BN + c) * 2 + 0] = a;
                red[(wm * BN + c) * 2 + 1] = b;
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < d.N) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) { a += red[(w * BN + tid) * 2]; b += red[(w * BN + tid) * 2 + 1]; }
            const int64_t part = (int64_t)phase * m_tiles_ + bx;
            d.stats[(part * 2 + 0) * d.N + n0 + tid] = a;
            d.stats[(part * 2 + 1) * d.N + n0 + tid] = b;
        }
        __syncthreads();
    }

    // C tile staging: pitch padded by 16 B; the tile is written in NPASS row blocks when it exceeds the LDS buffers
    constexpr int NPASS = (BM * (BN * ESZO + 16) <= NBUF * STAGE) ? 1 : WM;
    constexpr int PROWS = BM / NPASS;
    constexpr int CPITCH = BN * ESZO + ((PROWS * (BN * ESZO + 16) <= NBUF * STAGE) ? 16 : 0);
    static_assert(PROWS * CPITCH <= NBUF * STAGE, "C tile pass does not fit in LDS");
    constexpr int SEGS = BN * ESZO / 16;                // 16-byte segments per tile row
    unsigned char* Yb = reinterpret_cast<unsigned char*>(d.Y);
    const int oc_bytes = d.OC * ESZO;
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        const bool mine = (NPASS == 1) || (wm == pass);
        if (mine) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = wm * (BM / WM) + i * 16 + fg * 4 + r - pass * PROWS;
                        const int col = wn * (BN / WN) + j * 16 + fr;
                        typename EO::type* dst = reinterpret_cast<typename EO::type*>(smem + row * CPITCH) + col;
                        *dst = EO::from_f32(acc[i][j][r]);
                    }
        }
        __syncthreads();
        for (int u = tid; u < PROWS * SEGS; u += 256) {
            const int row = u / SEGS, seg = u - row * SEGS;
            const int op = opix_tab[pass * PROWS + row];
            const int cb = n0 * ESZO + seg * 16;        // byte offset of this segment inside the output pixel
            if (op >= 0 && cb < oc_bytes) {
                u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * CPITCH + seg * 16);
                if (d.mask_x != nullptr)
                    v = mask_segment<DTO>(v, *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(d.mask_x) +
                                                                           (int64_t)op * oc_bytes + cb), d.mask_act, d.mask_slope);
                *reinterpret_cast<u32x4*>(Yb + (int64_t)op * oc_bytes + cb) = v;
            }
        }
        if (pass + 1 < NPASS) __syncthreads();
    }
}

template <int DT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int ksplit, int M, int N,
                                                            int OC, int Mpad, int Npad, const float* __restrict__ bias,
                                                            void* __restrict__ Y) {
    const int64_t total = (int64_t)M * OC;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / OC), n = (int)(i - (int64_t)m * OC);
        float s = 0.f;
        if (n < N) {
            // eight slab loads in flight at a time (up to 64 slices: one dependent load per slice was 12 us for a
            // 13k-element output); the summation order k = 0, 1, 2, ... is unchanged
            const float* src = ws + (int64_t)m * Npad + n;
            const int64_t kstride = (int64_t)Mpad * Npad;
            int k = 0;
            for (; k + 8 <= ksplit; k += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(k + u) * kstride];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; k < ksplit; ++k) s += src[(int64_t)k * kstride];
            if (bias) s += bias[n];
        }
        store1<DT>(Y, i, s);
    }
}

// Slab reduce of the general split-K launch (round 4): one workgroup per (tile of bm rows, 64 columns, phase).  Sums the K
// slices in fixed order, adds the bias, forms the tile's BatchNorm partial sums (from the f32 values, valid output pixels only:
// what the un-split epilogue does) into the same slab row stats[(phase * m_tiles + tile) * 2 + {0,1}][N], and writes the
// output pixel through the phase's sub-pixel mapping as 16-byte channel runs.  ws: [phase][slice][Mpad][Npad] f32.
template <int DT>
__global__ __launch_bounds__(256) void splitk_reduce2_kernel(const vg_gg_desc d, const int ksplit, const int bm, const int Mpad,
                                                             const int Npad) {
    constexpr int VE = 16 / ElemT<DT>::size;                  // elements per 16-byte run: 4 (f32) / 8 (bf16)
    constexpr int LPR = 64 / VE;                              // lanes per 64-column row: 16 / 8
    constexpr int RPP = 256 / LPR;                            // rows per pass: 16 / 32
    __shared__ float red[2][RPP][64];
    const int tid = threadIdx.x;
    const int phase = blockIdx.z;
    const int m0 = blockIdx.x * bm, n0 = blockIdx.y * 64;
    const int M = d.B * d.GH * d.GW, GHW = d.GH * d.GW;
    const int rl = tid / LPR, cv = (tid % LPR) * VE;
    const bool flat = (d.nphase == 1 && d.OSY == 1 && d.OSX == 1 && d.GH == d.OH && d.GW == d.OW);
    const int64_t kstride = (int64_t)Mpad * Npad;
    const float* base = d.ws + (int64_t)phase * ksplit * kstride;
    float s1[VE], s2[VE], bv[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        s1[e] = 0.f; s2[e] = 0.f;
        const int n = n0 + cv + e;
        bv[e] = (d.bias != nullptr && n < d.N) ? d.bias[n] : 0.f;
    }
    for (int r = rl; r < bm; r += RPP) {
        const int m = m0 + r;
        int op = -1;
        if (m < M) {
            if (flat) {
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
        if (op < 0 || n0 + cv >= d.OC) continue;
        float v[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) v[e] = 0.f;
        if (n0 + cv < Npad) {
            const float* src = base + (int64_t)m * Npad + n0 + cv;
            for (int k = 0; k < ksplit; ++k) {                // fixed order k = 0, 1, 2, ...
#pragma unroll
                for (int q4 = 0; q4 < VE / 4; ++q4) {
                    const float4 t = *reinterpret_cast<const float4*>(src + (int64_t)k * kstride + 4 * q4);
                    v[4 * q4] += t.x; v[4 * q4 + 1] += t.y; v[4 * q4 + 2] += t.z; v[4 * q4 + 3] += t.w;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            const bool real = n0 + cv + e < d.N;
            v[e] = real ? v[e] + bv[e] : 0.f;
            s1[e] += v[e];
            s2[e] += v[e] * v[e];
        }
        if constexpr (DT == VG_F32) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.Y) + (int64_t)op * d.OC + n0 + cv) = float4{v[0], v[1], v[2], v[3]};
        } else {
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                o[k] = (uint32_t)ElemT<VG_BF16>::from_f32(v[2 * k]) | ((uint32_t)ElemT<VG_BF16>::from_f32(v[2 * k + 1]) << 16);
            *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(d.Y) + (int64_t)op * d.OC + n0 + cv) = o;
        }
    }
    if (d.stats == nullptr) return;
#pragma unroll
    for (int e = 0; e < VE; ++e) { red[0][rl][cv + e] = s1[e]; red[1][rl][cv + e] = s2[e]; }
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, c = tid & 63;
        float a = 0.f;
        for (int t = 0; t < RPP; ++t) a += red[which][t][c];
        if (n0 + c < d.N) {
            const int m_tiles = (M + bm - 1) / bm;
            d.stats[(((int64_t)phase * m_tiles + blockIdx.x) * 2 + which) * d.N + n0 + c] = a;
        }
    }
}

#include "conv_patch.hpp"
#include "conv_phase4.hpp"
#include "conv_narrowk.hpp"

struct TileCfg { int bm, bn; };
struct SplitK { int ksplit, sps; int64_t ws_bytes; };

inline int tiles_of(int M, int N, int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn); }

// Pick the output tile: small-N layers get tall-skinny tiles (they are HBM-bound); otherwise the
// largest tile that still fills the 256 CUs with >= ~2 workgroups each.
inline bool use_dma() { return vg_sw().gg_dma != 0; }            // VG_GG_DMA (common.hpp: switches are read once at load)

inline int min_wgs() { return vg_sw().tile_min_wgs; }            // VG_TILE_MIN_WGS

// (a 256x128 tile -- wave tile 128x64, 152 VGPRs, one wave per SIMD -- was measured and loses on every layer)
inline bool patch64() { return vg_sw().gg_patch64 != 0; }        // patch variant for the 128 x 64 tile (VG_GG_PATCH64)
inline bool patch32() { return vg_sw().gg_patch32 != 0; }        // ... for 32-output-channel layers (S >= 128; VG_GG_PATCH32)
inline bool patch_nr3() { return vg_sw().gg_patch_nr3 != 0; }    // 128 x 64 patch kernel with 3 patch rounds (40 KB LDS: 4 workgroups per CU)
// 256 x 128 tiles where every CU gets at least one (step sweep: 128 / 192 / 256 / 384 -> 38.3 / 38.6 / 38.7 / 38.7k img/s)
inline int patch256_min() { return vg_sw().patch256_min; }
// 256 x 64: G4 forward 63.9 -> 63.0 us, D1 data gradient 30.9 -> 28.4 (2B) and 17.8 -> 15.9 (B)
inline int patch256x64_min() { return vg_sw().patch256x64_min; }

// Few rows, wide N, long K, nothing in the epilogue that needs final values (the data gradient of the Generator's second
// ConvTranspose2d: M = B * 16 = 2048 rows, N = 1024, K = 8192 at S = 64): on 64 x 64 tiles every workgroup re-reads 64 weight
// rows for only 64 data rows (32 FLOP per operand byte: 75 us, the least efficient launch of the step); 128 x 128 tiles halve
// the operand bytes but offer 128 workgroups -- with the K dimension cut into slices on the LDS-DMA ring they offer 512.
inline bool bigk_split_ok(const vg_gg_desc* d, bool bf16) {
    if (!bf16 || vg_sw().splitk_bigk == 0 || !use_dma() || d->zeros == nullptr) return false;
    const int M = d->B * d->GH * d->GW;
    const bool flat = d->nphase == 1 && d->OSY == 1 && d->OSX == 1 && d->GH == d->OH && d->GW == d->OW;
    if (!flat || d->stats != nullptr || d->act != VG_ACT_NONE || d->mask_x != nullptr) return false;
    const int t128 = tiles_of(M, d->N, 128, 128);
    const int nstages = (d->Kp * 2) / 128;                        // 2 chunks of 64 bytes per stage
    return d->N >= 256 && M % 128 == 0 && t128 >= 64 && t128 <= 256 && nstages >= 32;      // (32 tiles -- S = 256, B = 32 -- measured slower: 3.62 -> 3.64 ms)
}

inline TileCfg pick_tile(const vg_gg_desc* d, bool bf16 = false, bool fp8 = false) {
    const int M = d->B * d->GH * d->GW;
    const int N = d->N;
    const int ph = d->nphase;
    const int need = min_wgs();
    if (fp8) {                                                    // plain register-staged tiles only
        if (N > 64 && tiles_of(M, N, 128, 128) * ph >= need) return {128, 128};
        if (tiles_of(M, N, 128, 64) * ph >= need) return {128, 64};
        return {64, 64};
    }
    if (bf16 && narrowk_ok(d, VG_BF16)) return {NK_BM, N};       // 3-channel-input edge layers: ggn_kernel, all N per workgroup
    if (bf16 && N <= 32 && use_dma()) {                           // narrow transposed layers: all four phases per workgroup
        Q4Geo qg; int nr;
        if (ggq_geometry(d, &qg, &nr)) return {256, N <= 16 ? 16 : 32};
    }
    if (N <= 16) return {256, 16};
    if (N <= 32) return {128, 32};
    // 256 x 128 exists only as the patch kernel (bf16, LDS-DMA): 8 waves share every weight tile
    PatchGeo pg;
    // 256 x 64 for 33..64 output channels (ggp_kernel<4, 64>: 80 KB of stage buffers, two 8-wave workgroups per CU,
    // every weight tile shared by 256 rows)
    if (bf16 && N > 32 && N <= 64 && use_patch() && use_dma() && d->zeros != nullptr &&
        tiles_of(M, N, 256, 64) * ph >= patch256x64_min() && patch_geometry(d, 256, &pg))
        return {256, 64};
    if (bf16 && N > 64 && use_patch() && use_dma() && d->zeros != nullptr &&
        tiles_of(M, N, 256, 128) * ph >= patch256_min() && patch_geometry(d, 256, &pg))
        return {256, 128};
    if (N > 64 && tiles_of(M, N, 128, 128) * ph >= need) return {128, 128};
    if (bigk_split_ok(d, bf16)) return {128, 128};               // few rows, long K: 128 x 128 tiles + split-K (plan_splitk)
    if (tiles_of(M, N, 128, 64) * ph >= need) return {128, 64};
    return {64, 64};
}

inline int validate(const vg_gg_desc* d, int dtype) {
    VG_CHECK_ARG(d != nullptr, VG_EINVAL);
    VG_CHECK_ARG(dtype == VG_F32 || dtype == VG_BF16 || dtype == VG_FP8, VG_ENOSUP);
    const int esz = dtype == VG_F32 ? 4 : (dtype == VG_BF16 ? 2 : 1);
    const int eszo = dtype == VG_F32 ? 4 : 2;                  // fp8 operands write bf16
    VG_CHECK_ARG(dtype != VG_FP8 || d->mask_x == nullptr, VG_ENOSUP);
    VG_CHECK_ARG(d->X && d->W && d->Y, VG_EINVAL);
    VG_CHECK_ARG(d->B > 0 && d->GH > 0 && d->GW > 0 && d->IH > 0 && d->IW > 0 && d->IC > 0, VG_EINVAL);
    VG_CHECK_ARG(d->N > 0 && d->OC >= d->N && d->OH > 0 && d->OW > 0, VG_EINVAL);
    VG_CHECK_ARG(d->TH > 0 && d->TW > 0 && d->nphase >= 1 && d->nphase <= VG_MAX_PHASE, VG_EINVAL);
    VG_CHECK_ARG((d->IC * esz) % 16 == 0, VG_EALIGN);
    VG_CHECK_ARG((d->Kp * esz) % 64 == 0 && d->Kp >= d->TH * d->TW * d->IC, VG_EALIGN);
    VG_CHECK_ARG(vg_aligned16(d->X) && vg_aligned16(d->W) && vg_aligned16(d->Y), VG_EALIGN);
    VG_CHECK_ARG((d->OC * eszo) % 16 == 0, VG_EALIGN);
    VG_CHECK_ARG(d->act == VG_ACT_NONE || ((d->act == VG_ACT_RELU || d->act == VG_ACT_LRELU) && d->stats == nullptr), VG_EINVAL);
    VG_CHECK_ARG(d->mask_x == nullptr || (vg_aligned16(d->mask_x) && (d->mask_act == VG_ACT_RELU || d->mask_act == VG_ACT_LRELU)), VG_EINVAL);
    VG_CHECK_ARG((int64_t)d->B * d->GH * d->GW < (1ll << 31), VG_EINVAL);
    VG_CHECK_ARG((int64_t)d->B * d->IH * d->IW < (1ll << 31), VG_EINVAL);
    VG_CHECK_ARG((int64_t)d->B * d->IH * d->IW * d->IC * esz < (1ll << 32), VG_ENOSUP);   // 32-bit gather offsets
    // every output pixel written by some phase must be inside the tensor (skips are allowed)
    for (int p = 0; p < d->nphase; ++p) VG_CHECK_ARG(d->ooy[p] >= 0 && d->oox[p] >= 0, VG_EINVAL);
    return 0;
}

// Split K for skinny problems -- few output tiles, long K.  Round 4: also with sub-pixel phases and with BatchNorm statistics
// (formed by the slab reduce, splitk_reduce2_kernel), on the LDS-DMA ring; never with a fused activation or mask (their
// epilogues need final values).  `dma`: the split launch will run on the DMA kernel (stage = DmaRing<bm, bn>::KCH chunks).
inline bool split_dma_ok(const vg_gg_desc* d, int dtype, TileCfg t) {
    return dtype == VG_BF16 && (t.bn % 64) == 0 && (t.bm == 64 || t.bm == 128) && use_dma() && d->zeros != nullptr;
}
inline SplitK plan_splitk(const vg_gg_desc* d, int dtype, TileCfg t) {
    SplitK r{1, 0, 0};
    const int M = d->B * d->GH * d->GW;
    const bool flat = d->nphase == 1 && d->OSY == 1 && d->OSX == 1 && d->GH == d->OH && d->GW == d->OW;
    const bool general = vg_sw().splitk_general != 0 && dtype == VG_BF16;     // phases / statistics allowed
    if (d->act != VG_ACT_NONE || d->mask_x != nullptr || dtype == VG_FP8) return r;
    if ((!flat || d->stats != nullptr) && !general) return r;
    if (t.bm == 256 || t.bn < 64) return r;                     // 256-row / narrow tiles: patch, 4-phase and edge kernels only
    const int esz = dtype == VG_F32 ? 4 : 2;
    const bool dma = split_dma_ok(d, dtype, t);
    int kch = dtype == VG_BF16 ? 2 : 1;
    if (dma) kch = t.bm == 128 ? (t.bn == 128 ? 2 : VG_RING128x64_KCH) : VG_RING64_KCH;
    const int nstages = ((d->Kp * esz) / 64 + kch - 1) / kch;
    const int gx = (M + t.bm - 1) / t.bm, gy = (d->N + t.bn - 1) / t.bn;
    const int tiles = gx * gy * d->nphase;
    const int max_tiles = vg_sw().splitk_max_tiles;
    const bool bigk = t.bm == 128 && t.bn == 128 && bigk_split_ok(d, dtype == VG_BF16);
    // the round-4 cases: 64 x 64-tile launches (the small-M layers) of up to one workgroup per CU unsplit; the 128-row tiles keep
    // their patch kernels (faster than the generic tile a split launch would run on)
    const bool wide = general && (!flat || d->stats != nullptr);
    if (wide && t.bm != 64) return r;
    if ((tiles > (wide ? 256 : max_tiles) && !bigk) || nstages < (dma && t.bm == 64 ? 8 : 16)) return r;
    // aim at ~4 workgroups per CU, at most 64 splits (the Encoder's Linear layers -- 4 tiles, K = 50 176 at S = 256 -- are at
    // their best there; more splits only add partial traffic), at least 4 stages per split (2 of the 4-chunk stages of the
    // 64 x 64 DMA tile).  The big-K case aims at 2 workgroups per CU: its f32 partials are the price (128 tiles x 64 KB per slice)
    int ks = (bigk ? 512 : vg_sw().splitk_wgs) / tiles;
    if (ks > 64) ks = 64;
    const int min_sps = (dma && t.bm == 64) ? 2 : 4;
    if (ks > nstages / min_sps) ks = nstages / min_sps;
    if (ks < 2) return r;
    r.sps = (nstages + ks - 1) / ks;
    r.ksplit = (nstages + r.sps - 1) / r.sps;
    r.ws_bytes = (int64_t)d->nphase * r.ksplit * gx * t.bm * (int64_t)(gy * t.bn) * 4;
    return r;
}

// Which operand an XCD keeps to itself.  Default: m tiles are dealt over the XCDs (each streams 1/8 of the gathered
// input and ALL weights).  Where the weights are the larger operand (Generator stage 1: 16.8 MB of weights against
// 4-8 MB of activations at B = 128) dealing the n tiles instead moves 8x less weight traffic out of the Infinity
// Cache.  VG_GG_NMAJOR=0 turns it off.
inline bool n_major(const vg_gg_desc* d, int n_tiles, int esz) {
    const int mode = vg_sw().gg_nmajor;
    if (mode == 0 || n_tiles % 8 != 0) return false;
    const int64_t wbytes = (int64_t)d->nphase * d->N * d->Kp * esz;
    const int64_t abytes = (int64_t)d->B * d->IH * d->IW * d->IC * esz;
    return mode == 2 || wbytes > abytes;
}

template <int DT, int BM, int BN, int WM, int WN>
int launch(const vg_gg_desc* d, hipStream_t s, SplitK sk) {
    const int M = d->B * d->GH * d->GW;
    const bool split = sk.ksplit > 1 && d->ws != nullptr && d->ws_bytes >= sk.ws_bytes;
    const int m_tiles = (M + BM - 1) / BM, n_tiles = (d->N + BN - 1) / BN;
    dim3 grid(((m_tiles + 7) / 8) * 8 * n_tiles, 1, d->nphase);
    const int nstages_all = 1 << 30;
    const int one = 1 | (n_major(d, n_tiles, ElemT<DT>::size) ? GG_N_MAJOR : 0);
    constexpr bool CAN_DMA = (DT == VG_BF16 || DT == VG_FP8) && (BN % 64 == 0);
    const bool dma = CAN_DMA && use_dma() && d->zeros != nullptr;
    if constexpr (DT == VG_FP8) {
        // LDS-DMA ring as the bf16 kernel (the staging code is byte-generic: a 64-byte chunk holds 64 e4m3 channels)
        if (dma) vg_launch_timed(3, gg_kernel<DT, BM, BN, WM, WN, false, true>, grid, dim3(256), 0, s, *d, one, nstages_all);
        else vg_launch_timed(3, gg_kernel<DT, BM, BN, WM, WN, false, false>, grid, dim3(256), 0, s, *d, one, nstages_all);
        return VG_LAUNCH_RC();
    } else if (split) {
        grid.z = d->nphase * sk.ksplit;
        bool launched = false;
        if constexpr (CAN_DMA && (BM == 64 || BM == 128)) {     // round 4: K slices on the LDS-DMA ring too
            if (dma && split_dma_ok(d, DT, TileCfg{BM, BN})) {
                vg_launch_timed(0, gg_kernel<DT, BM, BN, WM, WN, true, true>, grid, dim3(256), 0, s, *d, sk.ksplit, sk.sps);
                launched = true;
            }
        }
        if (!launched)
            vg_launch_timed(0, gg_kernel<DT, BM, BN, WM, WN, true, false>, grid, dim3(256), 0, s, *d, sk.ksplit, sk.sps);
    } else if constexpr (CAN_DMA) {
        if (dma) vg_launch_timed(0, gg_kernel<DT, BM, BN, WM, WN, false, true>, grid, dim3(256), 0, s, *d, one, nstages_all);
        else vg_launch_timed(0, gg_kernel<DT, BM, BN, WM, WN, false, false>, grid, dim3(256), 0, s, *d, one, nstages_all);
    } else {
        vg_launch_timed(DT == VG_FP8 ? 3 : 0, gg_kernel<DT, BM, BN, WM, WN, false, false>, grid, dim3(256), 0, s, *d, one, nstages_all);
    }
    int rc = VG_LAUNCH_RC();
    if (rc || !split) return rc;
    if constexpr (DT != VG_FP8) {
        const bool flat = d->nphase == 1 && d->OSY == 1 && d->OSX == 1 && d->GH == d->OH && d->GW == d->OW;
        if (flat && d->stats == nullptr) {                    // the round-1 reduce: elementwise over the flat output
            const int64_t total = (int64_t)M * d->OC;
            int blocks = (int)((total + 255) / 256);
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(splitk_reduce_kernel<DT>, dim3(blocks), dim3(256), 0, s, d->ws, sk.ksplit, M, d->N, d->OC,
                               m_tiles * BM, n_tiles * BN, d->bias, d->Y);
        } else {
            // (the output's padding channels [N, OC) of a pixel are written as zeros by the 64-column blocks that cover them)
            hipLaunchKernelGGL(splitk_reduce2_kernel<DT>, dim3(m_tiles, (d->OC + 63) / 64, d->nphase), dim3(256), 0, s, *d,
                               sk.ksplit, BM, m_tiles * BM, n_tiles * BN);
        }
    }
    return VG_LAUNCH_RC();
}

template <int DT>
int dispatch(const vg_gg_desc* d, TileCfg t, hipStream_t s, SplitK sk) {
    if constexpr (DT != VG_FP8) {
        if (t.bm == 256 && t.bn == 16) return launch<DT, 256, 16, 4, 1>(d, s, sk);
        if (t.bm == 128 && t.bn == 32) return launch<DT, 128, 32, 4, 1>(d, s, sk);
    }
    if (t.bm == 128 && t.bn == 128) return launch<DT, 128, 128, 2, 2>(d, s, sk);
    if (t.bm == 128 && t.bn == 64) return launch<DT, 128, 64, 2, 2>(d, s, sk);
    return launch<DT, 64, 64, 2, 2>(d, s, sk);
}

}  // namespace

#ifdef VG_DBG_STAMPS
extern "C" int vg_debug_stamps(unsigned long long* host_out, int n) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(vg_dbg_stamps), (size_t)n * 8);
}
#endif

extern "C" int vg_gather_gemm_nparts(const vg_gg_desc* d, int dtype) {
    int rc = validate(d, dtype);
    if (rc) return rc;
    TileCfg t = pick_tile(d, dtype == VG_BF16, dtype == VG_FP8);
    const int M = d->B * d->GH * d->GW;
    return d->nphase * ((M + t.bm - 1) / t.bm);
}

extern "C" int vg_gather_gemm_family(const vg_gg_desc* d, int dtype) {
    int rc = validate(d, dtype);
    if (rc) return rc;
    return dtype == VG_FP8 ? 3 : (narrowk_ok(d, dtype) ? 2 : 0);
}

extern "C" int vg_gather_gemm_tile_m(const vg_gg_desc* d, int dtype) {
    int rc = validate(d, dtype);
    if (rc) return rc;
    return pick_tile(d, dtype == VG_BF16, dtype == VG_FP8).bm;
}

extern "C" int64_t vg_gather_gemm_ws_bytes(const vg_gg_desc* d, int dtype) {
    int rc = validate(d, dtype);
    if (rc) return rc;
    return plan_splitk(d, dtype, pick_tile(d, dtype == VG_BF16, dtype == VG_FP8)).ws_bytes;
}

extern "C" int vg_gather_gemm(const vg_gg_desc* d, int dtype, void* stream) {
    int rc = validate(d, dtype);
    if (rc) return rc;
    PatchGeo pg;
    TileCfg t = pick_tile(d, dtype == VG_BF16, dtype == VG_FP8);
    const SplitK sk = plan_splitk(d, dtype, t);
    if (d->stats) {
        const int M = d->B * d->GH * d->GW;
        VG_CHECK_ARG(d->stats_capacity >= d->nphase * ((M + t.bm - 1) / t.bm), VG_EINVAL);
    }
    if (dtype == VG_F32) return dispatch<VG_F32>(d, t, vg_stream(stream), sk);
    if (dtype == VG_FP8) return dispatch<VG_FP8>(d, t, vg_stream(stream), sk);
    if (narrowk_ok(d, dtype)) return launch_narrowk(d, vg_stream(stream));
    {
        Q4Geo qg; int nr;
        if (t.bm == 256 && t.bn <= 32 && use_dma() && sk.ksplit <= 1 && ggq_geometry(d, &qg, &nr))
            return launch_phase4(d, qg, nr, vg_stream(stream));
        if (t.bm == 256 && t.bn == 32) return VG_EINVAL;       // (switch flipped between planning and launch)
    }
    if ((t.bm == 128 || t.bm == 256) && (t.bn == GP_BN || (t.bn == 64 && patch64()) || (t.bn == 32 && d->N == 32 && patch32())) && sk.ksplit <= 1 &&
        use_patch() && use_dma() && d->zeros != nullptr && patch_geometry(d, t.bm, &pg)) {
        const int m_tiles = (d->B * d->GH * d->GW) / t.bm, n_tiles = (d->N + t.bn - 1) / t.bn;
        dim3 grid(((m_tiles + 7) / 8) * 8 * n_tiles, 1, d->nphase);
        pg.n_major = n_major(d, n_tiles, 2) ? 1 : 0;
        if (t.bm == 256 && t.bn == 64) vg_launch_timed(0, (ggp_kernel<4, 64>), grid, dim3(512), 0, vg_stream(stream), *d, pg);
        else if (t.bm == 256) vg_launch_timed(0, ggp_kernel<4>, grid, dim3(512), 0, vg_stream(stream), *d, pg);
        else if (t.bn == 64 && pg.NPP <= 192 && patch_nr3()) vg_launch_timed(0, (ggp_kernel<2, 64, 3>), grid, dim3(256), 0, vg_stream(stream), *d, pg);
        else if (t.bn == 64) vg_launch_timed(0, (ggp_kernel<2, 64>), grid, dim3(256), 0, vg_stream(stream), *d, pg);
        else if (t.bn == 32) vg_launch_timed(0, (ggp_kernel<2, 32>), grid, dim3(256), 0, vg_stream(stream), *d, pg);
        else vg_launch_timed(0, ggp_kernel<2>, grid, dim3(256), 0, vg_stream(stream), *d, pg);
        return VG_LAUNCH_RC();
    }
    if (t.bm == 256 && t.bn >= 64) return VG_EINVAL;       // (env flipped between planning and launch)
    return dispatch<VG_BF16>(d, t, vg_stream(stream), sk);
}

// Weight-gradient implicit GEMM (autograd convolution_backward wgrad for nn.Conv2d /
// nn.ConvTranspose2d / nn.Linear of main_vae.py:23,47-48 and gan_code.py:21-84).
// Semantics: vg_wg_desc in include/vaegan_hip.h.
//
//   dW[np][cq][t] = sum_m P[m][np] * Q~[m][t][cq]        m = (b, gy, gx)
//
// The reduction dimension is the (huge) pixel index m, the output is the (small) weight
// tensor, so the grid is (kq tiles) x (np tiles) x (splits of the m range).  Every
// workgroup accumulates a 64x64 f32 tile over its m range on v_mfma_f32_16x16x4_f32 and
// stores it to a slab; a second kernel sums the slabs in fixed order (bitwise reproducible,
// no float atomics -- cdna_hip_programming.md Guideline 12) and scatters into the reference
// parameter layout.  Both operands are staged [m][channel] exactly as they sit in HBM (NHWC),
// which is the layout the 16x16x4 MFMA wants for a reduction over m: lane (i, k) reads
// element [4*step + k][16*tile + i], 16 consecutive floats per k row; rows are padded to 80
// floats so the two k rows of a 32-lane ds_read_b32 group hit disjoint bank halves.
#include "common.hpp"
#include <cstdlib>

namespace {

constexpr int WG_BNP = 64, WG_BKQ = 64, WG_BMK = 32, WG_LD = 80;

// XCD-aware workgroup order (cdna_hip_programming.md T1).  Workgroups are dealt round-robin over the 8 XCDs (private
// L2 each) in launch order x, y, z.  Every tile (x, y) of one split z reads the SAME rows of P and the same region of
// Q; with kq-tile counts that are multiples of 8 the plain order sends tile x to XCD x % 8, i.e. every XCD streams
// ALL rows of P and (through its taps) all of Q: 8x the operand traffic out of the Infinity Cache (G4: 537 MB, 77 us
// of loads for 37 us of MFMA).  Remapped, XCD k owns the k-th contiguous eighth of the (z, y, x) order: whole splits
// (or whole np-rows of one), dispatched back to back, so the rows are fetched into one L2 once.  Speed only.
struct WgTile { int x, y, z; };
__device__ __forceinline__ WgTile wg_tile(int xcd_order) {
    WgTile t = {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
    const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * (int)gridDim.z;
    if (xcd_order && (total & 7) == 0) {
        const int id = t.x + gx * (t.y + gy * t.z);
        const int l = (id & 7) * (total >> 3) + (id >> 3);
        t.x = l % gx;
        const int r = l / gx;
        t.y = r % gy;
        t.z = r / gy;
    }
    return t;
}


template <int DT>
__global__ __launch_bounds__(256) void wgrad_kernel(const vg_wg_desc d, int rows_per_split, int KQ, int NPpad, int xcd_order) {
    typedef ElemT<DT> E;
    constexpr int ESZ = E::size;
    constexpr int P16 = E::per16;
    constexpr int UPR = WG_BNP / P16;             // 16-byte units per tile row (16 f32 / 8 bf16)
    constexpr int RSTEP = 256 / UPR;              // rows covered per pass (16 / 32)
    constexpr int PASS = WG_BMK / RSTEP;          // 2 / 1
    __shared__ __attribute__((aligned(16))) float smem[2][2][WG_BMK * WG_LD];   // [buf][P|Q]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wnp = wave >> 1, wkq = wave & 1;
    const WgTile bt = wg_tile(xcd_order);
    const int kq0 = bt.x * WG_BKQ;
    const int np0 = bt.y * WG_BNP;
    const int M = d.B * d.GH * d.GW;
    const int GHW = d.GH * d.GW;
    const int m_begin = bt.z * rows_per_split;
    const int m_end = min(M, m_begin + rows_per_split);

    const int urow = tid / UPR;                   // row within a pass
    const int ucol = tid % UPR;                   // unit within the tile row
    // Q gather: this thread's unit maps to a fixed (tap, channel offset)
    const int kq_e = kq0 + ucol * P16;            // first element index along KQ
    const bool q_ok = kq_e < KQ;
    const int t = q_ok ? kq_e / d.QC : 0;
    const int cq = kq_e - t * d.QC;
    const int ta = t / d.TW, tb = t - ta * d.TW;
    const int qdy = d.y0 + d.DY * ta, qdx = d.x0 + d.DX * tb;
    const bool p_ok = np0 + ucol * P16 < d.PC;

    const unsigned char* Pb = reinterpret_cast<const unsigned char*>(d.P);
    const unsigned char* Qb = reinterpret_cast<const unsigned char*>(d.Q);

    u32x4 rp[PASS], rq[PASS];
    auto load_stage = [&](int ms) {
#pragma unroll
        for (int i = 0; i < PASS; ++i) {
            const int m = ms + urow + RSTEP * i;
            const bool ok = m < m_end;
            u32x4 vp = {0u, 0u, 0u, 0u}, vq = {0u, 0u, 0u, 0u};
            if (ok) {
                if (p_ok) vp = *reinterpret_cast<const u32x4*>(Pb + ((int64_t)m * d.PC + np0 + ucol * P16) * ESZ);
                if (q_ok) {
                    const int b = m / GHW;
                    const int r = m - b * GHW;
                    const int gy = r / d.GW;
                    const int gx = r - gy * d.GW;
                    const int iy = gy * d.SY + qdy, ix = gx * d.SX + qdx;
                    if ((unsigned)iy < (unsigned)d.QH && (unsigned)ix < (unsigned)d.QW)
                        vq = *reinterpret_cast<const u32x4*>(
                            Qb + (((int64_t)b * d.QH + iy) * d.QW + ix) * d.QC * ESZ + (int64_t)cq * ESZ);
                }
            }
            rp[i] = vp;
            rq[i] = vq;
        }
    };
    auto put = [&](float* dst, u32x4 v) {
        if constexpr (DT == VG_F32) {
            *reinterpret_cast<u32x4*>(dst) = v;
        } else {
            float4 lo, hi;
            lo.x = __uint_as_float(v[0] << 16); lo.y = __uint_as_float(v[0] & 0xffff0000u);
            lo.z = __uint_as_float(v[1] << 16); lo.w = __uint_as_float(v[1] & 0xffff0000u);
            hi.x = __uint_as_float(v[2] << 16); hi.y = __uint_as_float(v[2] & 0xffff0000u);
            hi.z = __uint_as_float(v[3] << 16); hi.w = __uint_as_float(v[3] & 0xffff0000u);
            *reinterpret_cast<float4*>(dst) = lo;
            *reinterpret_cast<float4*>(dst + 4) = hi;
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PASS; ++i) {
            const int r = urow + RSTEP * i;
            put(&smem[buf][0][r * WG_LD + ucol * P16], rp[i]);
            put(&smem[buf][1][r * WG_LD + ucol * P16], rq[i]);
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fi = lane & 15, fk = lane >> 4;
    const int nstage = (m_end - m_begin + WG_BMK - 1) / WG_BMK;
    if (nstage > 0) {
        load_stage(m_begin);
        store_stage(0);
    }
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        const int buf = s & 1;
        if (s + 1 < nstage) load_stage(m_begin + (s + 1) * WG_BMK);
        const float* sp = smem[buf][0];
        const float* sq = smem[buf][1];
#pragma unroll
        for (int ks = 0; ks < WG_BMK / 4; ++ks) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = sp[(ks * 4 + fk) * WG_LD + wnp * 32 + i * 16 + fi];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = sq[(ks * 4 + fk) * WG_LD + wkq * 32 + j * 16 + fi];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < nstage) store_stage(buf ^ 1);
        __syncthreads();
    }
    // slab[split][np][kq]
    float* slab = d.ws + (int64_t)bt.z * NPpad * (int64_t)(gridDim.x * WG_BKQ);
    const int ldk = gridDim.x * WG_BKQ;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int np = np0 + wnp * 32 + i * 16 + fk * 4 + r;
                const int kq = kq0 + wkq * 32 + j * 16 + fi;
                slab[(int64_t)np * ldk + kq] = acc[i][j][r];
            }
}

// ---------------------------------------------------------------------------------------------------
// bf16 variant on v_mfma_f32_16x16x32_bf16.  The reduction index of the MFMA is the pixel index m, which is
// the STRIDED dimension of both NHWC operands, so fragments are fetched with the gfx950 transposing LDS read
// (ds_read_b64_tr_b16: a 16-lane group reads a 4-row x 16-column block and receives it column-major).
//   * 128 x 128 output tile per workgroup (4 waves, 64 x 64 each = 4x4 MFMA tiles), 64 pixel rows per stage,
//     both operands staged [row][128 cols] exactly as they lie in HBM (8 x 16-byte loads in flight per thread);
//   * LDS rows are 256 B; the 32-byte column blocks are XOR-swizzled with (row & 7), and MFMA k index
//     (lane group g, half h, element q) is mapped to row 16h + 4g + q, so the 8 rows one 32-lane half touches
//     land in 8 different 32-byte bank groups: conflict-free transposed reads (cdna_hip_programming.md T10);
//     the k permutation is harmless because A and B use the same one;
//   * the (batch, y, x) decomposition of the 64 rows of a stage is done once by one wave into an LDS table.
constexpr int WB_T = 128, WB_SM = 64, WB_PITCH = 256;       // tile edge, rows per stage, LDS row bytes

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* tile, int krow0, int col0, int lane) {
    // operand fragment for 16 columns starting at col0, 32 k-rows starting at krow0
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int cb = col0 >> 4;
    const int r0 = krow0 + 4 * g + q, r1 = r0 + 16;
    const unsigned char* a0 = tile + r0 * WB_PITCH + ((cb ^ (r0 & 7)) << 5) + 8 * p;
    const unsigned char* a1 = tile + r1 * WB_PITCH + ((cb ^ (r1 & 7)) << 5) + 8 * p;
    bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)a0);
    bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)a1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(256) void wgrad_bf16_kernel(const vg_wg_desc d, int rows_per_split, int KQ, int NPpad, int xcd_order) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][WB_SM * WB_PITCH];   // [buf][P|Q]
    __shared__ int rowtab[2][WB_SM][3];                                                    // img base, iy0, ix0

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wnp = wave >> 1, wkq = wave & 1;
    const WgTile bt = wg_tile(xcd_order);
    const int kq0 = bt.x * WB_T;
    const int np0 = bt.y * WB_T;
    const int M = d.B * d.GH * d.GW;
    const int GHW = d.GH * d.GW;
    const int m_begin = bt.z * rows_per_split;
    const int m_end = min(M, m_begin + rows_per_split);

    const int unit = tid & 15;                     // 16-byte unit (8 bf16) within the 128-column tile row
    const int urow = tid >> 4;                     // 0..15, rows urow + 16*i
    const int kq_e = kq0 + unit * 8;
    const bool q_ok = kq_e < KQ;
    const int t = q_ok ? kq_e / d.QC : 0;
    const int cq = kq_e - t * d.QC;
    const int ta = t / d.TW, tb = t - ta * d.TW;
    const int qdy = d.DY * ta, qdx = d.DX * tb;
    const bool p_ok = np0 + unit * 8 < d.PC;
    const unsigned char* Pb = reinterpret_cast<const unsigned char*>(d.P);
    const unsigned char* Qb = reinterpret_cast<const unsigned char*>(d.Q);

    auto fill_table = [&](int buf, int ms) {
        if (tid < WB_SM) {
            const int m = ms + tid;
            int base = 0, iy0 = -(1 << 28), ix0 = 0;
            if (m < m_end) {
                const int b = m / GHW;
                const int r = m - b * GHW;
                const int gy = r / d.GW;
                const int gx = r - gy * d.GW;
                base = b * d.QH * d.QW;
                iy0 = gy * d.SY + d.y0;
                ix0 = gx * d.SX + d.x0;
            }
            rowtab[buf][tid][0] = base;
            rowtab[buf][tid][1] = iy0;
            rowtab[buf][tid][2] = ix0;
        }
    };

    u32x4 rp[4], rq[4];
    auto load_stage = [&](int buf, int ms) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = urow + 16 * i;
            const int m = ms + r;
            u32x4 vp = {0u, 0u, 0u, 0u}, vq = {0u, 0u, 0u, 0u};
            if (p_ok && m < m_end) vp = *reinterpret_cast<const u32x4*>(Pb + ((int64_t)m * d.PC + np0 + unit * 8) * 2);
            const int iy = rowtab[buf][r][1] + qdy, ix = rowtab[buf][r][2] + qdx;
            if (q_ok && (unsigned)iy < (unsigned)d.QH && (unsigned)ix < (unsigned)d.QW)
                vq = *reinterpret_cast<const u32x4*>(
                    Qb + ((int64_t)(rowtab[buf][r][0] + iy * d.QW + ix) * d.QC + cq) * 2);
            rp[i] = vp;
            rq[i] = vq;
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = urow + 16 * i;
            const int off = r * WB_PITCH + ((((unit >> 1) ^ (r & 7)) << 5) | ((unit & 1) << 4));
            *reinterpret_cast<u32x4*>(&smem[buf][0][off]) = rp[i];
            *reinterpret_cast<u32x4*>(&smem[buf][1][off]) = rq[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nstage = (m_end - m_begin + WB_SM - 1) / WB_SM;
    fill_table(0, m_begin);
    __syncthreads();
    if (nstage > 0) {
        load_stage(0, m_begin);
        store_stage(0);
    }
    if (nstage > 1) fill_table(1, m_begin + WB_SM);
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        const int buf = s & 1;
        if (s + 1 < nstage) load_stage(buf ^ 1, m_begin + (s + 1) * WB_SM);        // table[buf^1] filled last iteration
        const unsigned char* sp = smem[buf][0];
        const unsigned char* sq = smem[buf][1];
#pragma unroll
        for (int ks = 0; ks < WB_SM / 32; ++ks) {
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = tr_frag(sp, ks * 32, wnp * 64 + i * 16, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = tr_frag(sq, ks * 32, wkq * 64 + j * 16, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < nstage) store_stage(buf ^ 1);
        if (s + 2 < nstage) fill_table(buf, m_begin + (s + 2) * WB_SM);            // table[buf] no longer needed
        __syncthreads();
    }
    float* slab = d.ws + (int64_t)bt.z * NPpad * (int64_t)(gridDim.x * WB_T);
    const int ldk = gridDim.x * WB_T;
    const int fi = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int np = np0 + wnp * 64 + i * 16 + fk * 4 + r;
                const int kq = kq0 + wkq * 64 + j * 16 + fi;
                // the reduce kernel never reads the padding of a partly used tile (narrow layers: 3-channel
                // image operands, 32/64-channel outputs): do not spend slab bandwidth on it
                if (np < d.NP && kq < KQ) slab[(int64_t)np * ldk + kq] = acc[i][j][r];
            }
}

// ---------------------------------------------------------------------------------------------------
// LDS-DMA variant of the bf16 kernel (the default; VG_WG_DMA=0 selects the register-staged one above): same 128x128
// tile, MFMA and transposed fragment reads, but the operands go global -> LDS with global_load_lds_dwordx4 into a ring
// of WD_NBUF stages of WD_SM rows (2 x 32 KB = 64 KB, two workgroups per CU), no staging registers, no ds_write.  One wave instruction writes 4 rows x 256 B; a
// lane's row is 16j + 4*wave + (lane>>4), so (row & 7) -- the swizzle key -- does not depend on j and every lane
// fetches ONE fixed swizzled source unit (one fixed filter tap / channel offset) for the whole kernel.  Rows past
// the split range, channels past the tensor and out-of-image taps fetch from a zero page.
// Ring shape (measured, S=64 B=128 layer sweep, TFLOP/s incl. the slab reduce; register-staged kernel: G1 365,
// G2 438, G3 487): 32 rows x 3 slots 350 / 430 / 448, 32 x 4 355 / 410 / 451, 32 x 2 357 / 422 / 451,
// 64 rows x 2 slots 391 / 480 / 512 -- the barrier count per FLOP is what matters, not the depth of the ring.
#ifndef VG_WD_SM
#define VG_WD_SM 64
#endif
#ifndef VG_WD_NBUF
#define VG_WD_NBUF 2
#endif
constexpr int WD_SM = VG_WD_SM, WD_NBUF = VG_WD_NBUF, WD_STAGE = 2 * WD_SM * WB_PITCH;     // 16 KB per 32-row stage (P | Q)

#define WG_WAITCNT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

__global__ __launch_bounds__(256) void wgrad_bf16_dma_kernel(const vg_wg_desc d, int rows_per_split, int KQ, int NPpad, int xcd_order) {
    // one shared object: [ring of stages][row tables: 4 x 32 x {image base, iy0, ix0}]
    __shared__ __attribute__((aligned(16))) unsigned char smem[WD_NBUF * WD_STAGE + 4 * WD_SM * 3 * 4];
    int* const rowtab = reinterpret_cast<int*>(smem + WD_NBUF * WD_STAGE);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wnp = wave >> 1, wkq = wave & 1;
    const WgTile bt = wg_tile(xcd_order);
    const int kq0 = bt.x * WB_T;
    const int np0 = bt.y * WB_T;
    const int M = d.B * d.GH * d.GW;
    const int GHW = d.GH * d.GW;
    const int m_begin = bt.z * rows_per_split;
    const int m_end = min(M, m_begin + rows_per_split);

    const int rsub = lane >> 4;                              // row within the 4-row group of one DMA instruction
    const int upos = lane & 15;                              // 16-byte position within the 256-byte LDS row
    const int key = (4 * wave + rsub) & 7;                   // row & 7 for every row this lane ever loads
    const int unit = (((upos >> 1) ^ key) << 1) | (upos & 1);   // swizzled SOURCE unit (8 bf16 columns)
    const int kq_e = kq0 + unit * 8;
    const bool q_ok = kq_e < KQ;
    const int t = q_ok ? kq_e / d.QC : 0;
    const int cq = kq_e - t * d.QC;
    const int ta = t / d.TW, tb = t - ta * d.TW;
    const int qdy = d.DY * ta, qdx = d.DX * tb;
    const bool p_ok = np0 + unit * 8 < d.PC;
    const unsigned char* Pb = reinterpret_cast<const unsigned char*>(d.P);
    const unsigned char* Qb = reinterpret_cast<const unsigned char*>(d.Q);
    const int64_t zoffP = reinterpret_cast<const unsigned char*>(d.zeros) - Pb;
    const int64_t zoffQ = reinterpret_cast<const unsigned char*>(d.zeros) - Qb;

    auto fill_table = [&](int slot, int ms) {
        if (tid < WD_SM) {
            const int m = ms + tid;
            int base = 0, iy0 = -(1 << 28), ix0 = 0;
            if (m < m_end) {
                const int b = m / GHW;
                const int r = m - b * GHW;
                const int gy = r / d.GW;
                const int gx = r - gy * d.GW;
                base = b * d.QH * d.QW;
                iy0 = gy * d.SY + d.y0;
                ix0 = gx * d.SX + d.x0;
            }
            int* e = rowtab + (slot * WD_SM + tid) * 3;
            e[0] = base; e[1] = iy0; e[2] = ix0;
        }
    };
    auto issue_stage = [&](int buf, int slot, int ms) {
#ifdef VG_ABL_NO_LOAD
        return;
#endif
        unsigned char* sp = smem + buf * WD_STAGE;
        unsigned char* sq = sp + WD_SM * WB_PITCH;
        // all row-table entries of this lane's rows FIRST (one LDS round trip; read next to their use they were two
        // dependent LDS latencies in front of every Q instruction: ~1200 cycles per stage in which the wave issues
        // neither DMA nor MFMA), then every source offset, then the DMA instructions back to back
        int e0[WD_SM / 16], e1[WD_SM / 16], e2[WD_SM / 16];
#pragma unroll
        for (int j = 0; j < WD_SM / 16; ++j) {
            const int* e = rowtab + (slot * WD_SM + 16 * j + 4 * wave + rsub) * 3;
            e0[j] = e[0]; e1[j] = e[1]; e2[j] = e[2];
        }
        int64_t offp[WD_SM / 16], offq[WD_SM / 16];
#pragma unroll
        for (int j = 0; j < WD_SM / 16; ++j) {
            const int m = ms + 16 * j + 4 * wave + rsub;
            offp[j] = (p_ok && m < m_end) ? ((int64_t)m * d.PC + np0 + unit * 8) * 2 : zoffP;
            const int iy = e1[j] + qdy, ix = e2[j] + qdx;
            const bool ok = q_ok && (unsigned)iy < (unsigned)d.QH && (unsigned)ix < (unsigned)d.QW;
            offq[j] = ok ? ((int64_t)(e0[j] + iy * d.QW + ix) * d.QC + cq) * 2 : zoffQ;
        }
#pragma unroll
        for (int j = 0; j < WD_SM / 16; ++j) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Pb + offp[j]),
                                             (__attribute__((address_space(3))) void*)(sp + (16 * j + 4 * wave_u) * WB_PITCH),
                                             16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Qb + offq[j]),
                                             (__attribute__((address_space(3))) void*)(sq + (16 * j + 4 * wave_u) * WB_PITCH),
                                             16, 0, 0);
        }
    };
    constexpr int LDMA = 2 * (WD_SM / 16);                   // DMA instructions per wave per stage (4 | 8)

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nstage = (m_end - m_begin + WD_SM - 1) / WD_SM;
#pragma unroll
    for (int p = 0; p < WD_NBUF; ++p)
        if (p < nstage) fill_table(p, m_begin + p * WD_SM);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < WD_NBUF - 1; ++p)
        if (p < nstage) issue_stage(p, p, m_begin + p * WD_SM);
    int rb = 0, wb = WD_NBUF - 1;
    for (int s = 0; s < nstage; ++s) {
        // stages s+1 .. s+NBUF-2 may stay in flight
        const int ahead = min(WD_NBUF - 2, nstage - 1 - s);
        if (ahead * LDMA >= 8) WG_WAITCNT_VM(8);
        else if (ahead * LDMA >= 4) WG_WAITCNT_VM(4);
        else WG_WAITCNT_VM(0);
        __builtin_amdgcn_s_barrier();
        if (s + WD_NBUF - 1 < nstage) issue_stage(wb, (s + WD_NBUF - 1) & 3, m_begin + (s + WD_NBUF - 1) * WD_SM);
        if (s + WD_NBUF < nstage) fill_table((s + WD_NBUF) & 3, m_begin + (s + WD_NBUF) * WD_SM);
        const unsigned char* sp = smem + rb * WD_STAGE;
        const unsigned char* sq = sp + WD_SM * WB_PITCH;
#ifndef VG_ABLATE_COMPUTE
#pragma unroll
        for (int ks = 0; ks < WD_SM / 32; ++ks) {
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = tr_frag(sp, ks * 32, wnp * 64 + i * 16, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = tr_frag(sq, ks * 32, wkq * 64 + j * 16, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
#endif
        rb = rb == WD_NBUF - 1 ? 0 : rb + 1;
        wb = wb == WD_NBUF - 1 ? 0 : wb + 1;
    }
    static_assert(WD_NBUF >= 2 && WD_NBUF <= 4 && (LDMA == 4 || LDMA == 8), "vmcnt literals above");
    float* slab = d.ws + (int64_t)bt.z * NPpad * (int64_t)(gridDim.x * WB_T);
    const int ldk = gridDim.x * WB_T;
    const int fi = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int np = np0 + wnp * 64 + i * 16 + fk * 4 + r;
                const int kq = kq0 + wkq * 64 + j * 16 + fi;
                // the reduce kernel never reads the padding of a partly used tile (narrow layers: 3-channel
                // image operands, 32/64-channel outputs): do not spend slab bandwidth on it
#ifdef VG_ABL_NO_EPI
                if (np < d.NP && kq < KQ && acc[i][j][r] == 123.456f) slab[(int64_t)np * ldk + kq] = acc[i][j][r];
#else
                if (np < d.NP && kq < KQ) slab[(int64_t)np * ldk + kq] = acc[i][j][r];
#endif
            }
}

// ---------------------------------------------------------------------------------------------------
// Wave-specialised form of the LDS-DMA kernel (the default for bf16 where the kq tile count is even; VG_WG_SPEC):
// the same LDS image, fragment reads and slab epilogue, but DMA issue and arithmetic belong to DIFFERENT waves of one
// workgroup per CU -- 8 consumer waves (two per SIMD: transposed fragment reads + MFMA only) on a 128 x 256 tile (one
// P tile, two adjacent Q tiles: 25 % fewer operand bytes per FLOP) and 8 producer waves (row tables, source offsets,
// global_load_lds only), a ring of three 48 KB stages with two stages in flight across the single s_barrier per
// stage, counted vmcnt in the producers.  A producer wave owns row groups pw and pw + 8 of every operand tile:
// (row & 7), the swizzle key, is the same for both, so a lane still fetches one fixed source unit all kernel long.
// Why it wins where its parts lose (S=64 B=128, tools/ab_wgrad_spec.sh, all wgrad tests pass in every mode):
//   * one role per wave, 2 workgroups per CU (kernel above):            G1-G4 57 / 55 / 54 / 54 us, step 2.845 ms
//   * 128 x 256 tile, one role per wave, 1 workgroup per CU:            3-10 % slower (all waves in the same phase)
//   * 128 x 128 tile, 4 consumers + 8 producers (VG_WG_SPEC=1):         61 / 56 / 56 / 55 us (one consumer per SIMD
//     needs ~1000 cycles per stage for 512 cycles of MFMA: nothing fills its fragment-read latency)
//   * 128 x 256 tile, 8 consumers + 4 producers (VG_WG_SPEC=2):         56 / 53 / 51 / 50 us, step 2.797 ms
//   * 128 x 256 tile, 8 consumers + 8 producers (VG_WG_SPEC=3, default): 47 / 47 / 47 / 45 us = 726-770 TFLOP/s,
//     step 2.759 ms; wgrad family 458 -> 530 TFLOP/s.
// The kernels are bound by the per-CU operand ingest (~32 B/clk): the wide tile needs fewer
// bytes, two consumers per SIMD hide each other's LDS latency, and eight issuing waves keep the DMA stream dense.
constexpr int WS_NBUF = 3;

// NQT = 1: 128 x 128 tile, 4 consumer + 8 producer waves.  NQT = 2 (VG_WG_SPEC=2): 128 x 256 tile -- one P tile, two
// adjacent Q tiles, 8 consumer waves (two per SIMD) + 4 producer waves, 25 % fewer operand bytes per FLOP.
template <int NQT, int NPROD>
__global__ __launch_bounds__(64 * (4 * NQT + NPROD)) void wgrad_bf16_ws_kernel(const vg_wg_desc d, int rows_per_split, int KQ, int NPpad, int xcd_order) {
    constexpr int NCONS = 4 * NQT;
    constexpr int STAGE = (1 + NQT) * WD_SM * WB_PITCH;      // P | Q0 [| Q1]
    constexpr int NRG = 16 / NPROD;                           // row groups (of 4 rows) per producer wave and operand tile
    constexpr int LDMA = NRG * (1 + NQT);                     // DMA instructions per producer wave and stage
    static_assert(WD_SM == 64 && (NPROD == 8 || NPROD == 4), "producer wave pw owns row groups pw + NPROD * jj of a 64-row stage");
    __shared__ __attribute__((aligned(16))) unsigned char smem[WS_NBUF * STAGE + 4 * WD_SM * 3 * 4];
    int* const rowtab = reinterpret_cast<int*>(smem + WS_NBUF * STAGE);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool producer = wave_u >= NCONS;
    const int pw = wave_u - NCONS;                            // producer index (negative for consumers)
    const int wq = NQT == 2 ? (wave >> 2) & 1 : 0;            // consumers: which Q tile
    const int wnp = (wave & 3) >> 1, wkq = wave & 1;          // consumers: 64 x 64 quadrant of their 128 x 128 tile
    const WgTile bt = wg_tile(xcd_order);
    const int kq00 = bt.x * NQT * WB_T;
    const int np0 = bt.y * WB_T;
    const int M = d.B * d.GH * d.GW;
    const int GHW = d.GH * d.GW;
    const int m_begin = bt.z * rows_per_split;
    const int m_end = min(M, m_begin + rows_per_split);
    const int nstage = (m_end - m_begin + WD_SM - 1) / WD_SM;

    if (producer) {
        const int rsub = lane >> 4;                          // row within the 4-row group of one DMA instruction
        const int upos = lane & 15;                          // 16-byte position within the 256-byte LDS row
        const int key = (4 * pw + rsub) & 7;                 // row & 7 for every row group of this wave (NPROD * 4 % 8 == 0)
        const int unit = (((upos >> 1) ^ key) << 1) | (upos & 1);
        bool q_ok[NQT];
        int cq[NQT], qdy[NQT], qdx[NQT];
#pragma unroll
        for (int h = 0; h < NQT; ++h) {
            const int kq_e = kq00 + h * WB_T + unit * 8;
            q_ok[h] = kq_e < KQ;
            const int t = q_ok[h] ? kq_e / d.QC : 0;
            cq[h] = kq_e - t * d.QC;
            const int ta = t / d.TW, tb = t - ta * d.TW;
            qdy[h] = d.DY * ta;
            qdx[h] = d.DX * tb;
        }
        const bool p_ok = np0 + unit * 8 < d.PC;
        const unsigned char* Pb = reinterpret_cast<const unsigned char*>(d.P);
        const unsigned char* Qb = reinterpret_cast<const unsigned char*>(d.Q);
        const int64_t zoffP = reinterpret_cast<const unsigned char*>(d.zeros) - Pb;
        const int64_t zoffQ = reinterpret_cast<const unsigned char*>(d.zeros) - Qb;
        const int ptid = tid - 64 * NCONS;

        auto fill_table = [&](int slot, int ms) {            // first producer wave: one row per lane
            if (ptid < WD_SM) {
                const int m = ms + ptid;
                int base = 0, iy0 = -(1 << 28), ix0 = 0;
                if (m < m_end) {
                    const int b = m / GHW;
                    const int r = m - b * GHW;
                    const int gy = r / d.GW;
                    const int gx = r - gy * d.GW;
                    base = b * d.QH * d.QW;
                    iy0 = gy * d.SY + d.y0;
                    ix0 = gx * d.SX + d.x0;
                }
                int* e = rowtab + (slot * WD_SM + ptid) * 3;
                e[0] = base; e[1] = iy0; e[2] = ix0;
            }
        };
        auto issue_stage = [&](int buf, int slot, int ms) {
            unsigned char* sp = smem + buf * STAGE;
            int e0[NRG], e1[NRG], e2[NRG];
#pragma unroll
            for (int jj = 0; jj < NRG; ++jj) {
                const int* e = rowtab + (slot * WD_SM + 4 * (pw + NPROD * jj) + rsub) * 3;
                e0[jj] = e[0]; e1[jj] = e[1]; e2[jj] = e[2];
            }
#pragma unroll
            for (int jj = 0; jj < NRG; ++jj) {
                const int rg = pw + NPROD * jj;              // row group (4 rows) of these instructions
                const int m = ms + 4 * rg + rsub;
                const int64_t offp = (p_ok && m < m_end) ? ((int64_t)m * d.PC + np0 + unit * 8) * 2 : zoffP;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Pb + offp),
                                                 (__attribute__((address_space(3))) void*)(sp + 4 * rg * WB_PITCH),
                                                 16, 0, 0);
#pragma unroll
                for (int h = 0; h < NQT; ++h) {
                    const int iy = e1[jj] + qdy[h], ix = e2[jj] + qdx[h];
                    const bool ok = q_ok[h] && (unsigned)iy < (unsigned)d.QH && (unsigned)ix < (unsigned)d.QW;
                    const int64_t offq = ok ? ((int64_t)(e0[jj] + iy * d.QW + ix) * d.QC + cq[h]) * 2 : zoffQ;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Qb + offq),
                                                     (__attribute__((address_space(3))) void*)(sp + ((1 + h) * WD_SM + 4 * rg) * WB_PITCH),
                                                     16, 0, 0);
                }
            }
        };
        // row tables of the first three stages, visible to all producer waves before the first issue
#pragma unroll
        for (int p = 0; p < WS_NBUF; ++p)
            if (p < nstage) fill_table(p, m_begin + p * WD_SM);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                   // barrier P (all waves)
        if (nstage > 0) issue_stage(0, 0, m_begin);
        if (nstage > 1) issue_stage(1, 1, m_begin + WD_SM);
        for (int s = 0; s < nstage; ++s) {
            // stage s has landed once at most the LDMA instructions of stage s+1 are outstanding (in-order completion)
            if (s + 1 < nstage) {
                if constexpr (LDMA == 4) WG_WAITCNT_VM(4);
                else if constexpr (LDMA == 6) WG_WAITCNT_VM(6);
                else WG_WAITCNT_VM(12);
            } else {
                WG_WAITCNT_VM(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");               // barrier s: consumers may read s,
            if (s + 2 < nstage) issue_stage((s + 2) % WS_NBUF, (s + 2) & 3, m_begin + (s + 2) * WD_SM);   // slot of s-1 is free
            if (s + 3 < nstage) fill_table((s + 3) & 3, m_begin + (s + 3) * WD_SM);
        }
        static_assert(LDMA == 4 || LDMA == 6 || LDMA == 12, "vmcnt literals above");
        return;
    }

    // ---------------- consumers ----------------
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_barrier();                                                          // barrier P
    for (int s = 0; s < nstage; ++s) {
        __builtin_amdgcn_s_barrier();                                                      // barrier s
        const unsigned char* sp = smem + (s % WS_NBUF) * STAGE;
        const unsigned char* sq = sp + (1 + wq) * WD_SM * WB_PITCH;
#pragma unroll
        for (int ks = 0; ks < WD_SM / 32; ++ks) {
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = tr_frag(sp, ks * 32, wnp * 64 + i * 16, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = tr_frag(sq, ks * 32, wkq * 64 + j * 16, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    const int ldk = gridDim.x * NQT * WB_T;
    float* slab = d.ws + (int64_t)bt.z * NPpad * (int64_t)ldk;
    const int kq0 = kq00 + wq * WB_T;
    const int fi = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int np = np0 + wnp * 64 + i * 16 + fk * 4 + r;
                const int kq = kq0 + wkq * 64 + j * 16 + fi;
                if (np < d.NP && kq < KQ) slab[(int64_t)np * ldk + kq] = acc[i][j][r];
            }
}

// Sum the split slabs in fixed order and write the reference parameter layout.  One thread owns VC consecutive
// gathered channels (cq) of one np row and walks its T filter taps: per split lane the slab reads are 16-byte
// vectors that form whole 128-byte lines across the EL threads of a block, and the T taps of a weight row are
// contiguous in OIHW (s_t == 1), so every thread writes contiguous runs instead of 4-byte scatters.
// A block is EL (np, cq-group) pairs x SPL split lanes (EL*SPL = 256): with few outputs and ~1000 slabs a plain
// per-element loop is a serial chain of HBM latencies, so the slabs are walked by SPL lanes, combined via LDS.
template <int TT, int VC>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const vg_wg_desc d, int nsplit, int NPpad, int ldk, int SPL) {
    extern __shared__ __attribute__((aligned(16))) float red[];      // [SPL][EL*VC][TT]
    const int EL = 256 / SPL;
    const int e = threadIdx.x % EL, sp = threadIdx.x / EL;
    const int T = d.TH * d.TW;
    const int ngrp = (d.NQ + VC - 1) / VC;
    const int64_t total = (int64_t)d.NP * ngrp;
    const int64_t idx = (int64_t)blockIdx.x * EL + e;
    const bool ok = idx < total;
    const int cq0 = ok ? (int)(idx % ngrp) * VC : 0;
    const int np = ok ? (int)(idx / ngrp) : 0;
    const int64_t slab_stride = (int64_t)NPpad * ldk;
    const float* src = d.ws + (int64_t)np * ldk + cq0;
    {
        const int t0 = blockIdx.y * TT;                 // tap groups run in parallel (more blocks, shorter chains)
        float s[TT][VC];
#pragma unroll
        for (int j = 0; j < TT; ++j)
#pragma unroll
            for (int v = 0; v < VC; ++v) s[j][v] = 0.f;
        if (ok) {
            // U slab rows (U * TT loads) in flight per thread: one row at a time made the walk over up to 32 rows a
            // chain of dependent L2 latencies (11 us per launch, 23 launches per step)
            constexpr int U = (TT <= 4) ? 4 : 1;
            auto row = [&](int k, float (&acc)[TT][VC]) {
#pragma unroll
                for (int j = 0; j < TT; ++j) {
#pragma unroll
                    for (int v = 0; v < VC; ++v) acc[j][v] = 0.f;
                    if (t0 + j < T) {
                        const float* q = src + k * slab_stride + (int64_t)(t0 + j) * d.QC;
                        if (VC == 4) {
                            const float4 x = *reinterpret_cast<const float4*>(q);
                            acc[j][0] = x.x; acc[j][1 % VC] = x.y; acc[j][2 % VC] = x.z; acc[j][3 % VC] = x.w;
                        } else {
                            acc[j][0] = q[0];
                        }
                    }
                }
            };
            int k = sp;
            for (; k + (U - 1) * SPL < nsplit; k += U * SPL) {
                float t[U][TT][VC];
#pragma unroll
                for (int u = 0; u < U; ++u) row(k + u * SPL, t[u]);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int j = 0; j < TT; ++j)
#pragma unroll
                        for (int v = 0; v < VC; ++v) s[j][v] += t[u][j][v];
            }
            for (; k < nsplit; k += SPL) {
                float t[TT][VC];
                row(k, t);
#pragma unroll
                for (int j = 0; j < TT; ++j)
#pragma unroll
                    for (int v = 0; v < VC; ++v) s[j][v] += t[j][v];
            }
        }
        // all partials -> LDS once; then every thread adds the SPL partials of a few outputs in fixed order,
        // consecutive threads taking consecutive taps of one weight row (contiguous runs in OIHW)
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TT; ++j)
#pragma unroll
            for (int v = 0; v < VC; ++v) red[((sp * EL + e) * VC + v) * TT + j] = s[j][v];
        __syncthreads();
        const int nout = EL * VC * TT;
        for (int q = threadIdx.x; q < nout; q += 256) {
            const int j = q % TT, ev = q / TT;
            const int ee = ev / VC, v = ev - ee * VC;
            const int64_t oidx = (int64_t)blockIdx.x * EL + ee;
            if (oidx >= total || t0 + j >= T) continue;
            const int ocq = (int)(oidx % ngrp) * VC + v;
            if (ocq >= d.NQ) continue;
            const int onp = (int)(oidx / ngrp);
            float acc = 0.f;
            for (int k = 0; k < SPL; ++k) acc += red[(k * EL * VC + ev) * TT + j];
            float* dst = d.dW + (int64_t)onp * d.s_np + (int64_t)ocq * d.s_cq + (int64_t)(t0 + j) * d.s_t;
            *dst = d.accumulate ? (*dst + acc) : acc;
        }
    }
}

// Streaming form of the slab reduce for the weight-heavy layers (few splits, large dW: the Generator's 1024->512 and
// 512->256 layers, the Discriminator's 256->512): a block owns one np row x 64 gathered channels x all T <= 16 taps.
// Thread (tap, 4 channels) sums its float4 over the splits with whole 256-byte slab rows per tap (coalesced, 4 splits in
// flight), the [tap][cq] -> [cq][tap] turn happens in LDS, and the block writes ONE contiguous 64*T-float run of the
// reference weight layout ([np][cq][kh][kw]: s_cq == T, s_t == 1).  With one split this is the pure layout change the
// generic kernel spent 32 us on (67 MB at 2 TB/s); same fixed summation order (k = 0, 1, ...), bitwise reproducible.
__global__ __launch_bounds__(256) void wgrad_reduce_t_kernel(const vg_wg_desc d, int nsplit, int NPpad, int ldk) {
    __shared__ __attribute__((aligned(16))) float tile[64 * 16];
    const int T = d.TH * d.TW;
    const int cqt = d.QC >> 6;                                   // 64-channel tiles per np row
    const int np = blockIdx.x / cqt, cq0 = (blockIdx.x - np * cqt) << 6;
    const int t = threadIdx.x >> 4, c4 = threadIdx.x & 15;
    float4 s = {0.f, 0.f, 0.f, 0.f};
    if (t < T) {
        const float* src = d.ws + (int64_t)np * ldk + (int64_t)t * d.QC + cq0 + c4 * 4;
        const int64_t stride = (int64_t)NPpad * ldk;
        int k = 0;
        for (; k + 4 <= nsplit; k += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(src + (k + u) * stride);
#pragma unroll
            for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; k < nsplit; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(src + k * stride);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        // [cq][tap] image, 16 lanes (c4) write the same tap column of 16 different rows: with plain rows of T = 16
        // floats that is ONE bank pair (SQ_LDS_BANK_CONFLICT was 91 % of this kernel's LDS cycles); the column is
        // XOR-ed with the row's c4, which spreads a wave's write over 16 banks
        const int sw = T == 16 ? c4 : 0;
        float* tl = tile + (c4 * 4) * T + (t ^ sw);
        tl[0] = s.x; tl[T] = s.y; tl[2 * T] = s.z; tl[3 * T] = s.w;
    }
    __syncthreads();
    float* dst = d.dW + (int64_t)np * d.s_np + (int64_t)cq0 * T;
    for (int i = threadIdx.x * 4; i < 64 * T; i += 1024) {       // 64*T % 4 == 0; run start is 16-byte aligned (T*cq0 % 4 == 0)
        float4 v;
        if (T == 16) {
            const float* row = tile + (i & ~15);
            const int key = (i >> 6) & 15, c = i & 15;           // row i/16 = cq, its c4 = cq >> 2
            v.x = row[c ^ key]; v.y = row[(c + 1) ^ key]; v.z = row[(c + 2) ^ key]; v.w = row[(c + 3) ^ key];
        } else {
            v = *reinterpret_cast<const float4*>(tile + i);
        }
        if (d.accumulate) {
            const float4 o = *reinterpret_cast<const float4*>(dst + i);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *reinterpret_cast<float4*>(dst + i) = v;
    }
}

// the streaming reduce takes: <= 16 taps, contiguous taps, channel stride == tap count, whole 64-channel tiles,
// 16-byte aligned runs
inline bool Ttaps_ok(const vg_wg_desc* d) {
    const int on = vg_sw().wg_reduce_t;
    const int T = d->TH * d->TW;
    return on && T <= 16 && d->s_t == 1 && d->s_cq == T && d->QC % 64 == 0 && d->NQ == d->QC && (d->s_np % 4) == 0 &&
           ((64 * T) % 4) == 0 && vg_aligned16(d->dW);
}

struct Plan { int tiles_kq, tiles_np, nsplit, rows_per_split, KQ, NPpad, tile; int64_t ws_bytes; };

inline int wg_target() { return vg_sw().wg_target; }

// 0: one role per wave (wgrad_bf16_dma_kernel); 1: 128 x 128 tile, 4 consumer + 8 producer waves; 2: 128 x 256 tile,
// 8 consumers + 4 producers; 3 (default): 128 x 256 tile, 8 consumers + 8 producers.
inline int wg_spec() { return vg_sw().wg_spec; }

inline bool wg_use_dma(const vg_wg_desc* d) {
    // default on with 64-row stages (+5..10 % on the generator layers); VG_WG_DMA=0 -> register-staged kernel
    return vg_sw().wg_dma != 0 && d->zeros != nullptr;
}

inline int make_plan(const vg_wg_desc* d, int dtype, Plan* p) {
    VG_CHECK_ARG(d != nullptr, VG_EINVAL);
    VG_CHECK_ARG(dtype == VG_F32 || dtype == VG_BF16, VG_ENOSUP);
    const int esz = dtype == VG_F32 ? 4 : 2;
    VG_CHECK_ARG(d->B > 0 && d->GH > 0 && d->GW > 0 && d->PC > 0 && d->QC > 0, VG_EINVAL);
    VG_CHECK_ARG(d->NP > 0 && d->NP <= d->PC && d->NQ > 0 && d->NQ <= d->QC, VG_EINVAL);
    VG_CHECK_ARG((d->PC * esz) % 16 == 0 && (d->QC * esz) % 16 == 0, VG_EALIGN);
    VG_CHECK_ARG(d->TH > 0 && d->TW > 0, VG_EINVAL);
    const int64_t M = (int64_t)d->B * d->GH * d->GW;
    VG_CHECK_ARG(M < (1ll << 31), VG_EINVAL);
    p->KQ = d->TH * d->TW * d->QC;
    const int tile = dtype == VG_F32 ? WG_BNP : WB_T;          // output tile edge
    const int srows = dtype == VG_F32 ? WG_BMK : (wg_use_dma(d) ? WD_SM : WB_SM);     // pixel rows per stage
    p->tile = tile;
    p->tiles_kq = (p->KQ + tile - 1) / tile;
    p->tiles_np = (d->PC + tile - 1) / tile;
    p->NPpad = p->tiles_np * tile;
    const int tiles = p->tiles_kq * p->tiles_np;
    int64_t stages = (M + srows - 1) / srows;
    int nsplit = (int)((wg_target() + tiles - 1) / tiles);    // ~2 workgroups per CU
    // at least 8 (f32) / 4 (bf16) stages of work per workgroup, at most 1024 splits
    const int min_stages = dtype == VG_F32 ? 8 : 4;
    if (nsplit > stages / min_stages) nsplit = (int)(stages / min_stages);
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 1024) nsplit = 1024;
    int64_t rps = ((stages + nsplit - 1) / nsplit) * srows;
    nsplit = (int)((M + rps - 1) / rps);
    p->nsplit = nsplit;
    p->rows_per_split = (int)rps;
    p->ws_bytes = (int64_t)nsplit * p->NPpad * (int64_t)(p->tiles_kq * tile) * 4;
    return 0;
}

}  // namespace

extern "C" int64_t vg_wgrad_ws_bytes(const vg_wg_desc* d, int dtype) {
    Plan p;
    int rc = make_plan(d, dtype, &p);
    return rc ? (int64_t)rc : p.ws_bytes;
}

extern "C" int vg_wgrad(const vg_wg_desc* d, int dtype, void* stream) {
    Plan p;
    int rc = make_plan(d, dtype, &p);
    if (rc) return rc;
    VG_CHECK_ARG(d->P && d->Q && d->dW && d->ws, VG_EINVAL);
    VG_CHECK_ARG(vg_aligned16(d->P) && vg_aligned16(d->Q) && vg_aligned16(d->ws), VG_EALIGN);
    VG_CHECK_ARG(d->ws_bytes >= p.ws_bytes, VG_EINVAL);
    hipStream_t s = vg_stream(stream);
    dim3 grid(p.tiles_kq, p.tiles_np, p.nsplit);
    // XCD-aware order (wg_tile) where it pays: few tiles per split, many splits, long operands (measured S=64 B=128:
    // G4 81 -> 55 us, D1 2B 44 -> 31, D1 26 -> 21, G3 55 -> 54; layers with >= 128 tiles per split lose 2-3 us)
    const int xcd_env = vg_sw().wg_xcd;
    const int64_t Mrows = (int64_t)d->B * d->GH * d->GW;
    const int xcd_order = xcd_env == 2 || (xcd_env == 1 && p.tiles_kq * p.tiles_np <= 32 && p.nsplit >= 16 && Mrows >= 32768);
    if (dtype == VG_F32)
        vg_launch_timed(1, wgrad_kernel<VG_F32>, grid, dim3(256), 0, s, *d, p.rows_per_split, p.KQ, p.NPpad, xcd_order);
    else if (wg_use_dma(d) && wg_spec() == 3 && p.tiles_kq % 2 == 0)
        vg_launch_timed(1, (wgrad_bf16_ws_kernel<2, 8>), dim3(p.tiles_kq / 2, p.tiles_np, p.nsplit), dim3(1024), 0, s, *d,
                        p.rows_per_split, p.KQ, p.NPpad, xcd_order);
    else if (wg_use_dma(d) && wg_spec() == 2 && p.tiles_kq % 2 == 0)
        vg_launch_timed(1, (wgrad_bf16_ws_kernel<2, 4>), dim3(p.tiles_kq / 2, p.tiles_np, p.nsplit), dim3(768), 0, s, *d,
                        p.rows_per_split, p.KQ, p.NPpad, xcd_order);
    else if (wg_use_dma(d) && wg_spec() != 0)
        vg_launch_timed(1, (wgrad_bf16_ws_kernel<1, 8>), grid, dim3(768), 0, s, *d, p.rows_per_split, p.KQ, p.NPpad, xcd_order);
    else if (wg_use_dma(d))
        vg_launch_timed(1, wgrad_bf16_dma_kernel, grid, dim3(256), 0, s, *d, p.rows_per_split, p.KQ, p.NPpad, xcd_order);
    else
        vg_launch_timed(1, wgrad_bf16_kernel, grid, dim3(256), 0, s, *d, p.rows_per_split, p.KQ, p.NPpad, xcd_order);
    rc = VG_LAUNCH_RC();
    if (rc) return rc;
    // weight-heavy layers with few splits: streaming transpose-reduce (one contiguous run of dW per block)
    if (dtype == VG_BF16 && Ttaps_ok(d) && p.nsplit <= 16 && (int64_t)d->NP * (d->QC >> 6) >= 256) {
        hipLaunchKernelGGL(wgrad_reduce_t_kernel, dim3((unsigned)(d->NP * (d->QC >> 6))), dim3(256), 0, s, *d, p.nsplit,
                           p.NPpad, p.tiles_kq * p.tile);
        return VG_LAUNCH_RC();
    }
    const bool vec = (d->NQ % 4 == 0) && (d->QC % 4 == 0);
    const int VC = vec ? 4 : 1;
    const int64_t total = (int64_t)d->NP * ((d->NQ + VC - 1) / VC);
    const int SPL = p.nsplit >= 64 ? 32 : (p.nsplit >= 8 ? 8 : 1);
    const int EL = 256 / SPL;
    const int ldk = p.tiles_kq * p.tile;
    const int Ttaps = d->TH * d->TW;
    // 16 taps per thread only when there is plenty of parallelism anyway; otherwise 4 (tap groups -> blockIdx.y)
    const bool big = Ttaps >= 16 && (total + EL - 1) / EL >= 2048;
    const int TTv = big ? 16 : 4;
    const dim3 rgrid((unsigned)((total + EL - 1) / EL), (unsigned)((Ttaps + TTv - 1) / TTv));
    const size_t shm = (size_t)256 * VC * TTv * sizeof(float);                    // <= 64 KB
    if (vec && big) hipLaunchKernelGGL((wgrad_reduce_kernel<16, 4>), rgrid, dim3(256), shm, s, *d, p.nsplit, p.NPpad, ldk, SPL);
    else if (vec) hipLaunchKernelGGL((wgrad_reduce_kernel<4, 4>), rgrid, dim3(256), shm, s, *d, p.nsplit, p.NPpad, ldk, SPL);
    else if (big) hipLaunchKernelGGL((wgrad_reduce_kernel<16, 1>), rgrid, dim3(256), shm, s, *d, p.nsplit, p.NPpad, ldk, SPL);
    else hipLaunchKernelGGL((wgrad_reduce_kernel<4, 1>), rgrid, dim3(256), shm, s, *d, p.nsplit, p.NPpad, ldk, SPL);
    return VG_LAUNCH_RC();
}

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

namespace {

constexpr int WG_BNP = 64, WG_BKQ = 64, WG_BMK = 32, WG_LD = 80;

template <int DT>
__global__ __launch_bounds__(256) void wgrad_kernel(const vg_wg_desc d, int rows_per_split, int KQ, int NPpad) {
    typedef ElemT<DT> E;
    constexpr int ESZ = E::size;
    constexpr int P16 = E::per16;
    constexpr int UPR = WG_BNP / P16;             // 16-byte units per tile row (16 f32 / 8 bf16)
    constexpr int RSTEP = 256 / UPR;              // rows covered per pass (16 / 32)
    constexpr int PASS = WG_BMK / RSTEP;          // 2 / 1
    __shared__ __attribute__((aligned(16))) float smem[2][2][WG_BMK * WG_LD];   // [buf][P|Q]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wnp = wave >> 1, wkq = wave & 1;
    const int kq0 = blockIdx.x * WG_BKQ;
    const int np0 = blockIdx.y * WG_BNP;
    const int M = d.B * d.GH * d.GW;
    const int GHW = d.GH * d.GW;
    const int m_begin = blockIdx.z * rows_per_split;
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
    float* slab = d.ws + (int64_t)blockIdx.z * NPpad * (int64_t)(gridDim.x * WG_BKQ);
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

__global__ void wgrad_reduce_kernel(const vg_wg_desc d, int nsplit, int NPpad, int ldk) {
    // one thread per (np, t, cq) real weight element
    const int T = d.TH * d.TW;
    const int64_t total = (int64_t)d.NP * T * d.NQ;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cqi = (int)(idx % d.NQ);
    const int64_t r = idx / d.NQ;
    const int t = (int)(r % T);
    const int np = (int)(r / T);
    const int kq = t * d.QC + cqi;
    const int64_t slab_stride = (int64_t)NPpad * ldk;
    const float* src = d.ws + (int64_t)np * ldk + kq;
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += src[k * slab_stride];
    float* dst = d.dW + (int64_t)np * d.s_np + (int64_t)cqi * d.s_cq + (int64_t)t * d.s_t;
    *dst = d.accumulate ? (*dst + s) : s;
}

struct Plan { int tiles_kq, tiles_np, nsplit, rows_per_split, KQ, NPpad; int64_t ws_bytes; };

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
    p->tiles_kq = (p->KQ + WG_BKQ - 1) / WG_BKQ;
    p->tiles_np = (d->PC + WG_BNP - 1) / WG_BNP;
    p->NPpad = p->tiles_np * WG_BNP;
    const int tiles = p->tiles_kq * p->tiles_np;
    int64_t stages = (M + WG_BMK - 1) / WG_BMK;
    int nsplit = (int)((1024 + tiles - 1) / tiles);
    // at least 8 stages of work per workgroup, at most 1024 splits
    if (nsplit > stages / 8) nsplit = (int)(stages / 8);
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 1024) nsplit = 1024;
    int64_t rps = ((stages + nsplit - 1) / nsplit) * WG_BMK;
    nsplit = (int)((M + rps - 1) / rps);
    p->nsplit = nsplit;
    p->rows_per_split = (int)rps;
    p->ws_bytes = (int64_t)nsplit * p->NPpad * (int64_t)(p->tiles_kq * WG_BKQ) * 4;
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
    if (dtype == VG_F32)
        hipLaunchKernelGGL(wgrad_kernel<VG_F32>, grid, dim3(256), 0, s, *d, p.rows_per_split, p.KQ, p.NPpad);
    else
        hipLaunchKernelGGL(wgrad_kernel<VG_BF16>, grid, dim3(256), 0, s, *d, p.rows_per_split, p.KQ, p.NPpad);
    rc = VG_LAUNCH_RC();
    if (rc) return rc;
    const int64_t total = (int64_t)d->NP * d->TH * d->TW * d->NQ;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, *d, p.nsplit,
                       p.NPpad, p.tiles_kq * WG_BKQ);
    return VG_LAUNCH_RC();
}

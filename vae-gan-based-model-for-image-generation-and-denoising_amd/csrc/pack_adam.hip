// Operand packing (f32 parameters in the reference layouts -> K-major GEMM operands) and the
// flat Adam step (torch.optim.Adam as built at vaegan_code.py:42-44).
#include "common.hpp"

namespace {

template <int DT>
__device__ __forceinline__ void pack_body(const vg_pack_desc& d) {
    const int64_t total = (int64_t)d.nphase * d.N * d.Kp;
    const int T = d.TH * d.TW;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx % d.Kp);
        const int64_t r = idx / d.Kp;
        const int n = (int)(r % d.N);
        const int p = (int)(r / d.N);
        const int t = k / d.IC;
        const int ci = k - t * d.IC;
        float v = 0.f;
        if (t < T && ci < d.C) {
            if (d.tap_in_n) {
                // n = tap*CO + co ; src[ci][co][tap]  (ConvTranspose2d on a 1x1 input, gan_code.py:21)
                const int CO = d.N / d.KHW;
                const int tap = n / CO, co = n - tap * CO;
                v = d.src[(int64_t)ci * d.s_c + (int64_t)co * d.s_n + tap];
            } else {
                const int a = t / d.TW, c = t - a * d.TW;
                const int kh = d.kh0[p] + d.kh_step * a;
                const int kw = d.kw0[p] + d.kw_step * c;
                v = d.src[(int64_t)n * d.s_n + (int64_t)ci * d.s_c + kh * d.KW + kw];
            }
        }
        store1<DT>(d.dst, idx, v);
    }
}

template <int DT>
__global__ void pack_kernel(const vg_pack_desc d) { pack_body<DT>(d); }

// All operand copies of one network in ONE launch: descriptor table in device memory, blockIdx.y = descriptor.
// Tiled through LDS: a block owns PK_NT rows (n) x PK_CT channels (c) x up to PK_TT filter taps of the SOURCE
// tensor, reads them along the source's fastest dimensions (whole 64..1024-byte runs instead of 4-byte gathers
// with a 64-byte stride), and writes the K-major operand rows of every sub-pixel phase with the channel index
// fastest (128-byte runs).  Each source element is read exactly once per operand.
constexpr int PK_NT = 4, PK_CT = 64, PK_TT = 16;

__host__ __device__ inline int pack_tiles(const vg_pack_desc& d) {
    const int rowsN = d.tap_in_n ? d.N / d.KHW : d.N;
    return ((d.IC + PK_CT - 1) / PK_CT) * ((rowsN + PK_NT - 1) / PK_NT) * ((d.KHW + PK_TT - 1) / PK_TT);
}

// One workgroup per tile over ALL descriptors of the table (flat grid: a 2-D (tile, descriptor) grid sized by the
// largest layer launched ~50k workgroups of which most had nothing to do): descriptor i owns the flat tiles
// [tile_start_i, tile_start_{i+1}); the owner of a tile is found with one load round + ballot over <= 64 entries.
__device__ __forceinline__ int pk_sel4(const int32_t* a, int p) {      // a[p] for p < 4 without a memory round trip
    return p == 0 ? a[0] : p == 1 ? a[1] : p == 2 ? a[2] : a[3];
}

template <int DT>
__global__ __launch_bounds__(256) void pack_multi_kernel(const vg_pack_desc* __restrict__ descs, int ndesc) {
    __shared__ float tile[PK_NT][PK_CT][PK_TT + 1];
    // owner of this tile: every wave finds it by itself (one load round + ballot) and keeps it in an SGPR, so the
    // descriptor fields below are scalar loads -- a workgroup is one short dependent chain (lookup -> descriptor ->
    // source tile -> stores), and the launch is bound by the length of that chain, not by bytes
    const int ts = (int)(threadIdx.x & 63) < ndesc ? descs[threadIdx.x & 63].tile_start : 0x7fffffff;
    const int owner = __builtin_amdgcn_readfirstlane(__popcll(__ballot(ts <= (int)blockIdx.x)) - 1);
    const vg_pack_desc d = descs[owner];
    const int rowsN = d.tap_in_n ? d.N / d.KHW : d.N;           // rows of the source "n" index
    const int ctiles = (d.IC + PK_CT - 1) / PK_CT;
    const int ntiles = (rowsN + PK_NT - 1) / PK_NT;
    const int ttiles = (d.KHW + PK_TT - 1) / PK_TT;
    const int nblocks = ctiles * ntiles * ttiles;
    const int T = d.TH * d.TW;
    for (int blk = (int)blockIdx.x - d.tile_start; blk < nblocks; blk += nblocks) {
        const int ct = blk % ctiles;
        const int nt = (blk / ctiles) % ntiles;
        const int tt = blk / (ctiles * ntiles);
        const int c0 = ct * PK_CT, n0 = nt * PK_NT, tap0 = tt * PK_TT;
        const int ntap = min(PK_TT, d.KHW - tap0);
        __syncthreads();
        // ---- load (division-free for power-of-two tap counts): consecutive lanes walk the source contiguously ----
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const bool pow2 = (ntap & (ntap - 1)) == 0;
        const int sh = 31 - __builtin_clz(ntap);
        const bool vec4 = (ntap == PK_TT) && ((d.s_n | d.s_c | tap0) % 4 == 0) &&
                          ((reinterpret_cast<uintptr_t>(d.src) & 15u) == 0);
        if (vec4) {
            // 16-tap tiles: 1024 float4 per block, 4 per thread, all in flight at once
            float4 v[4];
            int nn_[4], cc_[4], t4_[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = k * 256 + threadIdx.x;            // 0..1023
                int nn, cc, t4;
                if (d.s_c <= d.s_n) { t4 = (idx & 3) * 4; cc = (idx >> 2) & (PK_CT - 1); nn = idx >> 8; }   // [n][c][tap]
                else { t4 = (idx & 3) * 4; nn = (idx >> 2) & (PK_NT - 1); cc = idx >> 4; }                  // [c][n][tap]
                nn_[k] = nn; cc_[k] = cc; t4_[k] = t4;
                const int n = n0 + nn, c = c0 + cc;
                v[k] = float4{0.f, 0.f, 0.f, 0.f};
                if (n < rowsN && c < d.C)
                    v[k] = *reinterpret_cast<const float4*>(d.src + (int64_t)n * d.s_n + (int64_t)c * d.s_c + tap0 + t4);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float* t = &tile[nn_[k]][cc_[k]][t4_[k]];
                t[0] = v[k].x; t[1] = v[k].y; t[2] = v[k].z; t[3] = v[k].w;
            }
        } else if (d.s_c <= d.s_n) {
            // [n][c][tap] source: for a fixed n the (c, tap) block is one contiguous run -> wave = n row
            const int n = n0 + wave;
            for (int e = lane; e < PK_CT * ntap; e += 64) {
                const int cc = pow2 ? (e >> sh) : (e / ntap);
                const int tp = pow2 ? (e & (ntap - 1)) : (e - cc * ntap);
                const int c = c0 + cc;
                float v = 0.f;
                if (n < rowsN && c < d.C) v = d.src[(int64_t)n * d.s_n + (int64_t)c * d.s_c + tap0 + tp];
                tile[wave][cc][tp] = v;
            }
        } else {
            // [c][n][tap] source: for a fixed c the (n, tap) block of the 4 rows is contiguous -> lanes = (n, tap)
            const int per_c = PK_NT * ntap;
            for (int cc = wave; cc < PK_CT; cc += 4) {
                const int c = c0 + cc;
                for (int e = lane; e < per_c; e += 64) {
                    const int nn = pow2 ? (e >> sh) : (e / ntap);
                    const int tp = pow2 ? (e & (ntap - 1)) : (e - nn * ntap);
                    const int n = n0 + nn;
                    float v = 0.f;
                    if (n < rowsN && c < d.C) v = d.src[(int64_t)n * d.s_n + (int64_t)c * d.s_c + tap0 + tp];
                    tile[nn][cc][tp] = v;
                }
            }
        }
        __syncthreads();
        // ---- store: wave = n row of the tile, lane = channel (128-byte runs), loops over phases and taps ----
        {
            const int nn = wave, cc = lane;
            const int n = n0 + nn, c = c0 + cc;
            if (d.tap_in_n) {
                // dst[(tap*CO + co)][ci]: one K-major row per (tap, co)
                if (n < rowsN) {
                    for (int tp = 0; tp < ntap; ++tp) {
                        const int64_t row = ((int64_t)(tap0 + tp) * rowsN + n) * d.Kp;
                        if (c < d.Kp) store1<DT>(d.dst, row + c, c < d.IC ? tile[nn][cc][tp] : 0.f);
                        if (ct == ctiles - 1)
                            for (int k = ctiles * PK_CT + lane; k < d.Kp; k += 64) store1<DT>(d.dst, row + k, 0.f);
                    }
                }
                continue;
            }
            // 16-byte stores: a lane owns VE consecutive channels, LPR lanes cover the 64-channel run of one
            // (phase, tap) row and the wave walks 64/LPR rows per instruction (IC is a multiple of VE by contract)
            constexpr int VE = 16 / ElemT<DT>::size;             // 4 (f32) / 8 (bf16)
            constexpr int LPR = PK_CT / VE;                      // 16 / 8 lanes per row run
            constexpr int RPI = 64 / LPR;                        // 4 / 8 rows per wave instruction
            const int sub = lane / LPR, cv = (lane % LPR) * VE;
            const int cbase = c0 + cv;
            if (n < d.N && cbase < d.IC) {
                const int rows = d.nphase * T;
                for (int r0 = 0; r0 < rows; r0 += RPI) {
                    const int r = r0 + sub;
                    if (r >= rows) break;
                    const int p = r / T, t = r - p * T;
                    const int a = t / d.TW, b = t - a * d.TW;
                    const int ft = (pk_sel4(d.kh0, p) + d.kh_step * a) * d.KW + pk_sel4(d.kw0, p) + d.kw_step * b - tap0;
                    if (ft < 0 || ft >= ntap) continue;
                    typename ElemT<DT>::type* q = reinterpret_cast<typename ElemT<DT>::type*>(d.dst) +
                                                  ((int64_t)p * d.N + n) * d.Kp + (int64_t)t * d.IC + cbase;
                    if constexpr (DT == VG_F32) {
                        float4 v = {tile[nn][cv][ft], tile[nn][cv + 1][ft], tile[nn][cv + 2][ft], tile[nn][cv + 3][ft]};
                        *reinterpret_cast<float4*>(q) = v;
                    } else {
                        u32x4 v;
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            v[k] = (uint32_t)ElemT<VG_BF16>::from_f32(tile[nn][cv + 2 * k][ft]) |
                                   ((uint32_t)ElemT<VG_BF16>::from_f32(tile[nn][cv + 2 * k + 1][ft]) << 16);
                        *reinterpret_cast<u32x4*>(q) = v;
                    }
                }
            }
            (void)c; (void)cc;
        }
        // K padding [T*IC, Kp) of the rows of this n tile (written once, by the first channel/tap tile)
        if (ct == 0 && tt == 0) {
            const int w = d.Kp - T * d.IC;
            for (int e = threadIdx.x; e < d.nphase * PK_NT * w; e += 256) {
                const int k = T * d.IC + e % w;
                const int r = e / w;
                const int n = n0 + r % PK_NT, p = r / PK_NT;
                if (n < d.N) store1<DT>(d.dst, ((int64_t)p * d.N + n) * d.Kp + k, 0.f);
            }
        }
    }
}

// state[0] = t (float), state[1] = lr/(1-b1^t), state[2] = sqrt(1-b2^t)   (double math)
__global__ void adam_prep_kernel(float* state, double lr, double beta1, double beta2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double t = (double)state[0] + 1.0;
        state[0] = (float)t;
        const double bc1 = 1.0 - pow(beta1, t);
        const double bc2 = 1.0 - pow(beta2, t);
        state[1] = (float)(lr / bc1);
        state[2] = (float)sqrt(bc2);
    }
}

// The iteration's prologue in ONE single-thread launch: the noise generator's iteration counter (vg_rng_advance) and the
// step counters / bias corrections of up to VG_PROLOGUE_MAX optimizers (what adam_prep_kernel does per optimizer).
struct PrologueArgs {
    unsigned long long* rng;
    float* zero;                          // loss slots of the iteration, zeroed here instead of by a memset node
    int nzero;
    float* state[VG_PROLOGUE_MAX];
    double lr[VG_PROLOGUE_MAX], beta1[VG_PROLOGUE_MAX], beta2[VG_PROLOGUE_MAX];
    int n;
};
__global__ void step_prologue_kernel(const PrologueArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (a.rng != nullptr) a.rng[1] += 1ull;
    for (int i = 0; i < a.nzero; ++i) a.zero[i] = 0.f;
    for (int i = 0; i < a.n; ++i) {
        float* state = a.state[i];
        const double t = (double)state[0] + 1.0;
        state[0] = (float)t;
        state[1] = (float)(a.lr[i] / (1.0 - pow(a.beta1[i], t)));
        state[2] = (float)sqrt(1.0 - pow(a.beta2[i], t));
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n4,
                                                   int64_t n, float w1, float beta2, float w2, float eps,
                                                   float gscale, const float* __restrict__ state) {
    const float step_size = state[1];
    const float bc2_sqrt = state[2];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float* P = &pp.x; float* G = &gg.x; float* Mv = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gr = G[k] * gscale;
            Mv[k] = Mv[k] + w1 * (gr - Mv[k]);                 // exp_avg.lerp_(grad, 1-beta1)
            V[k] = V[k] * beta2 + w2 * (gr * gr);              // mul_(beta2).addcmul_(g, g, 1-beta2)
            const float denom = sqrtf(V[k]) / bc2_sqrt + eps;
            P[k] = P[k] - step_size * (Mv[k] / denom);         // addcdiv_(exp_avg, denom, -step_size)
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    // tail (n not a multiple of 4)
    if (blockIdx.x == 0) {
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
            const float gr = g[i] * gscale;
            float mi = m[i], vi = v[i];
            mi = mi + w1 * (gr - mi);
            vi = vi * beta2 + w2 * (gr * gr);
            const float denom = sqrtf(vi) / bc2_sqrt + eps;
            p[i] = p[i] - step_size * (mi / denom);
            m[i] = mi;
            v[i] = vi;
        }
    }
}

// Two optimizers in ONE launch (the Encoder's and the Generator's step at the end of the iteration, vaegan_code.py:134-135):
// the first nb0 workgroups walk buffer 0, the rest buffer 1 -- the arithmetic per element is adam_kernel's.
struct Adam2Args {
    float* p[2]; const float* g[2]; float* m[2]; float* v[2]; const float* state[2];
    int64_t n[2];
    float w1[2], beta2[2], w2[2], eps[2], gscale[2];
    int nb0;
};
__global__ __launch_bounds__(256) void adam2_kernel(const Adam2Args a) {
    const int which = blockIdx.x < a.nb0 ? 0 : 1;
    const int bid = which == 0 ? blockIdx.x : blockIdx.x - a.nb0;
    const int nb = which == 0 ? a.nb0 : (int)gridDim.x - a.nb0;
    float* __restrict__ p = a.p[which]; const float* __restrict__ g = a.g[which];
    float* __restrict__ m = a.m[which]; float* __restrict__ v = a.v[which];
    const float w1 = a.w1[which], beta2 = a.beta2[which], w2 = a.w2[which], eps = a.eps[which], gscale = a.gscale[which];
    const float step_size = a.state[which][1];
    const float bc2_sqrt = a.state[which][2];
    const int64_t n = a.n[which], n4 = n / 4;
    for (int64_t i = (int64_t)bid * blockDim.x + threadIdx.x; i < n4; i += (int64_t)nb * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float* P = &pp.x; float* G = &gg.x; float* Mv = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gr = G[k] * gscale;
            Mv[k] = Mv[k] + w1 * (gr - Mv[k]);
            V[k] = V[k] * beta2 + w2 * (gr * gr);
            const float denom = sqrtf(V[k]) / bc2_sqrt + eps;
            P[k] = P[k] - step_size * (Mv[k] / denom);
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (bid == 0) {
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
            const float gr = g[i] * gscale;
            float mi = m[i], vi = v[i];
            mi = mi + w1 * (gr - mi);
            vi = vi * beta2 + w2 * (gr * gr);
            const float denom = sqrtf(vi) / bc2_sqrt + eps;
            p[i] = p[i] - step_size * (mi / denom);
            m[i] = mi;
            v[i] = vi;
        }
    }
}

}  // namespace

extern "C" int vg_pack_weights(const vg_pack_desc* d, int dtype, void* stream) {
    VG_CHECK_ARG(d && d->src && d->dst, VG_EINVAL);
    VG_CHECK_ARG(dtype == VG_F32 || dtype == VG_BF16, VG_ENOSUP);
    VG_CHECK_ARG(d->nphase >= 1 && d->nphase <= VG_MAX_PHASE && d->N > 0 && d->C > 0 && d->IC >= d->C, VG_EINVAL);
    VG_CHECK_ARG(d->Kp >= d->TH * d->TW * d->IC, VG_EINVAL);
    if (d->tap_in_n) VG_CHECK_ARG(d->KHW > 0 && d->N % d->KHW == 0 && d->TH == 1 && d->TW == 1, VG_EINVAL);
    const int64_t total = (int64_t)d->nphase * d->N * d->Kp;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (dtype == VG_F32) hipLaunchKernelGGL(pack_kernel<VG_F32>, dim3(blocks), dim3(256), 0, vg_stream(stream), *d);
    else hipLaunchKernelGGL(pack_kernel<VG_BF16>, dim3(blocks), dim3(256), 0, vg_stream(stream), *d);
    return VG_LAUNCH_RC();
}

extern "C" int vg_pack_tile_count(const vg_pack_desc* d) {
    VG_CHECK_ARG(d && d->N > 0 && d->IC > 0 && d->KHW > 0, VG_EINVAL);
    if (d->tap_in_n) VG_CHECK_ARG(d->N % d->KHW == 0, VG_EINVAL);
    return pack_tiles(*d);
}

extern "C" int vg_pack_weights_multi(const vg_pack_desc* descs_dev, int n, int64_t total_tiles, int dtype,
                                     void* stream) {
    VG_CHECK_ARG(descs_dev && n > 0 && n <= 64 && total_tiles > 0 && total_tiles < (1ll << 31), VG_EINVAL);
    VG_CHECK_ARG(dtype == VG_F32 || dtype == VG_BF16, VG_ENOSUP);
    const dim3 grid((unsigned)total_tiles);
    if (dtype == VG_F32) hipLaunchKernelGGL(pack_multi_kernel<VG_F32>, grid, dim3(256), 0, vg_stream(stream), descs_dev, n);
    else hipLaunchKernelGGL(pack_multi_kernel<VG_BF16>, grid, dim3(256), 0, vg_stream(stream), descs_dev, n);
    return VG_LAUNCH_RC();
}

static int adam_launch(float* p, const float* g, float* m, float* v, int64_t n, double beta1, double beta2, double eps,
                       float grad_scale, float* state, hipStream_t s) {
    const int64_t n4 = n / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    // the weights are computed in double exactly as Python does (1 - 0.9 = 0.09999999999999998), then
    // narrowed once to float, which is what the ATen kernels do with the scalar arguments.
    const float w1 = (float)(1.0 - beta1);
    const float w2 = (float)(1.0 - beta2);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, p, g, m, v, n4, n, w1, (float)beta2, w2, (float)eps, grad_scale,
                       state);
    return VG_LAUNCH_RC();
}

extern "C" int vg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                            double beta2, double eps, float grad_scale, float* state, void* stream) {
    VG_CHECK_ARG(p && g && m && v && state && n > 0, VG_EINVAL);
    VG_CHECK_ARG(lr >= 0.0, VG_EINVAL);                      // torch.optim.Adam: "Invalid learning rate"
    VG_CHECK_ARG(vg_aligned16(p) && vg_aligned16(g) && vg_aligned16(m) && vg_aligned16(v), VG_EALIGN);
    hipStream_t s = vg_stream(stream);
    hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(64), 0, s, state, lr, beta1, beta2);
    int rc = VG_LAUNCH_RC();
    if (rc) return rc;
    return adam_launch(p, g, m, v, n, beta1, beta2, eps, grad_scale, state, s);
}

extern "C" int vg_adam_apply(float* p, const float* g, float* m, float* v, int64_t n, double beta1, double beta2, double eps,
                             float grad_scale, const float* state, void* stream) {
    VG_CHECK_ARG(p && g && m && v && state && n > 0, VG_EINVAL);
    VG_CHECK_ARG(vg_aligned16(p) && vg_aligned16(g) && vg_aligned16(m) && vg_aligned16(v), VG_EALIGN);
    return adam_launch(p, g, m, v, n, beta1, beta2, eps, grad_scale, const_cast<float*>(state), vg_stream(stream));
}

extern "C" int vg_adam_apply2(float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n,
                              const double* beta1, const double* beta2, const double* eps, const float* grad_scale,
                              const float* const* state, void* stream) {
    VG_CHECK_ARG(p && g && m && v && n && beta1 && beta2 && eps && grad_scale && state, VG_EINVAL);
    Adam2Args a{};
    int nb[2];
    for (int i = 0; i < 2; ++i) {
        VG_CHECK_ARG(p[i] && g[i] && m[i] && v[i] && state[i] && n[i] > 0, VG_EINVAL);
        VG_CHECK_ARG(vg_aligned16(p[i]) && vg_aligned16(g[i]) && vg_aligned16(m[i]) && vg_aligned16(v[i]), VG_EALIGN);
        a.p[i] = p[i]; a.g[i] = g[i]; a.m[i] = m[i]; a.v[i] = v[i]; a.state[i] = state[i]; a.n[i] = n[i];
        a.w1[i] = (float)(1.0 - beta1[i]); a.beta2[i] = (float)beta2[i]; a.w2[i] = (float)(1.0 - beta2[i]);
        a.eps[i] = (float)eps[i]; a.gscale[i] = grad_scale[i];
        int64_t b = (n[i] / 4 + 255) / 256;
        nb[i] = (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));       // adam_launch's grid: same elements per thread, same results
    }
    a.nb0 = nb[0];
    hipLaunchKernelGGL(adam2_kernel, dim3(nb[0] + nb[1]), dim3(256), 0, vg_stream(stream), a);
    return VG_LAUNCH_RC();
}

extern "C" int vg_step_prologue(uint64_t* rng, float* const* states, const double* lr, const double* beta1,
                                const double* beta2, int n, float* zero, int nzero, void* stream) {
    VG_CHECK_ARG(n >= 0 && n <= VG_PROLOGUE_MAX && (n == 0 || (states && lr && beta1 && beta2)) && (rng || n > 0), VG_EINVAL);
    VG_CHECK_ARG(nzero >= 0 && nzero <= 64 && (nzero == 0 || zero != nullptr), VG_EINVAL);
    PrologueArgs a{};
    a.rng = (unsigned long long*)rng;
    a.zero = zero;
    a.nzero = nzero;
    a.n = n;
    for (int i = 0; i < n; ++i) {
        VG_CHECK_ARG(states[i] != nullptr && lr[i] >= 0.0, VG_EINVAL);
        a.state[i] = states[i]; a.lr[i] = lr[i]; a.beta1[i] = beta1[i]; a.beta2[i] = beta2[i];
    }
    hipLaunchKernelGGL(step_prologue_kernel, dim3(1), dim3(1), 0, vg_stream(stream), a);
    return VG_LAUNCH_RC();
}

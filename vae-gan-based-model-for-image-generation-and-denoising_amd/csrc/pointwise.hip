// Layout changes at the NCHW boundary, instance noise, reparameterisation + KL, the
// Discriminator's 1-channel head, and the BCE / MSE losses (vaegan_code.py:74-78, 88-117).
// All of these are HBM- or latency-bound pointwise / small-reduction kernels.
#include "common.hpp"
#include "noise.hpp"

namespace {

// bf16 -> e4m3fn with a power-of-two pre-scale; 8 elements (16 B in, 8 B out) per thread and pass
__global__ __launch_bounds__(256) void cast_fp8_kernel(const u32x4* __restrict__ x, uint2* __restrict__ y, int64_t nvec,
                                                       float scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        const u32x4 v = x[i];
        float f[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f[2 * k] = __uint_as_float(v[k] << 16) * scale;
            f[2 * k + 1] = __uint_as_float(v[k] & 0xffff0000u) * scale;
        }
        uint2 o;
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w, true);
        o.x = (uint32_t)w;
        w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w, true);
        o.y = (uint32_t)w;
        y[i] = o;
    }
}

__global__ void rng_advance_kernel(unsigned long long* state) { state[1] += 1ull; }

__global__ __launch_bounds__(256) void randn_fill_kernel(float* __restrict__ out, int64_t n, NoiseSrc src) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = noise_at(src, i);
}

// ---- NCHW f32  <->  NHWC (dtype, channels padded to CP) ----------------------------------------
// Four consecutive pixels of a row per thread (HW % 4 == 0, CP == 8, C <= 4, 16-byte aligned planes): float4 loads of
// every channel plane, ONE Philox block per channel for the four pixels' noise, four 16-byte pixel stores.
template <int DT>
__global__ __launch_bounds__(256) void nchw_to_nhwc_x4_kernel(const float* __restrict__ x, const NoiseSrc eps, float sigma,
                                                              void* __restrict__ y, int64_t npix4, int C, int HW,
                                                              float lo, float hi, float* __restrict__ y_nchw,
                                                              void* __restrict__ y_plain = nullptr) {
    static_assert(DT == VG_BF16, "8 bf16 channels = one 16-byte pixel");
    const bool noisy = eps.eps != nullptr || eps.rng != nullptr;
    for (int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i4 < npix4; i4 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = i4 * 4;
        const int64_t b = i / HW;
        const int64_t hw = i - b * HW;
        float v[4][4];                                          // [pixel][channel]
        float q[4][4];                                          // the same pixels without the noise (y_plain)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < C) {
                const int64_t src = (b * C + c) * HW + hw;
                const float4 t4 = *reinterpret_cast<const float4*>(x + src);
                float t[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
                for (int p = 0; p < 4; ++p) q[p][c] = t[p];
                if (noisy) {
                    const Normal4 nz = noise4_at(eps, src);
#pragma unroll
                    for (int p = 0; p < 4; ++p) t[p] = t[p] + sigma * nz.v[p];
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) { t[p] = fminf(fmaxf(t[p], lo), hi); v[p][c] = t[p]; }
                if (y_nchw) *reinterpret_cast<float4*>(y_nchw + src) = float4{t[0], t[1], t[2], t[3]};
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p) { v[p][c] = 0.f; q[p][c] = 0.f; }
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            u32x4 o;
            o[0] = (uint32_t)ElemT<VG_BF16>::from_f32(v[p][0]) | ((uint32_t)ElemT<VG_BF16>::from_f32(v[p][1]) << 16);
            o[1] = (uint32_t)ElemT<VG_BF16>::from_f32(v[p][2]) | ((uint32_t)ElemT<VG_BF16>::from_f32(v[p][3]) << 16);
            o[2] = 0u; o[3] = 0u;
            *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(y) + (i + p) * 16) = o;
            if (y_plain) {
                o[0] = (uint32_t)ElemT<VG_BF16>::from_f32(q[p][0]) | ((uint32_t)ElemT<VG_BF16>::from_f32(q[p][1]) << 16);
                o[1] = (uint32_t)ElemT<VG_BF16>::from_f32(q[p][2]) | ((uint32_t)ElemT<VG_BF16>::from_f32(q[p][3]) << 16);
                *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(y_plain) + (i + p) * 16) = o;
            }
        }
    }
}

template <int DT>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, const NoiseSrc eps,
                                                           float sigma, void* __restrict__ y, int64_t npix, int C,
                                                           int HW, int CP, float lo, float hi, float* __restrict__ y_nchw) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW;
        const int64_t hw = i - b * HW;
        for (int c0 = 0; c0 < CP; c0 += 4) {          // CP % 4 == 0: one 8- / 16-byte store per 4 channels
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + k;
                v[k] = 0.f;
                if (c < C) {
                    const int64_t src = (b * C + c) * HW + hw;
                    float t = x[src];
                    if (eps.eps || eps.rng) t = t + sigma * noise_at(eps, src);
                    t = fminf(fmaxf(t, lo), hi);
                    if (y_nchw) y_nchw[src] = t;
                    v[k] = t;
                }
            }
            store4<DT>(y, i * CP + c0, float4{v[0], v[1], v[2], v[3]});
        }
    }
}

// ---- data path: batch assembly from the HBM-resident u8 dataset (dataset_code.py:137-178) ---------------------
// images: [N][H][W][C] u8 (as the JPEG decoder leaves them), idx: B sample indices (the sampler's batch).
// out[b][c][h][w] = (u/255 - 0.5)/0.5 : transforms.ToTensor() (float32 u / 255) followed by
// transforms.Normalize((0.5,), (0.5,)) (x.sub(0.5).div(0.5)), dataset_code.py:147-150 -- IEEE division and
// subtraction in that order, so the batch equals the reference's bit for bit.
__global__ __launch_bounds__(256) void gather_u8_kernel(const uint8_t* __restrict__ images,
                                                        const int64_t* __restrict__ idx, int64_t npix, int C, int HW,
                                                        int64_t N, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW;
        const int64_t hw = i - b * HW;
        int64_t n = idx[b];
        if (n < 0 || n >= N) n = 0;                    // never read outside the dataset (host validates too)
        const uint8_t* src = images + (n * HW + hw) * C;
        for (int c = 0; c < C; ++c) {
            const float t = __fdiv_rn((float)src[c], 255.0f);
            out[(b * C + c) * HW + hw] = __fdiv_rn(__fsub_rn(t, 0.5f), 0.5f);
        }
    }
}

template <int DT>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const void* __restrict__ x, float* __restrict__ y,
                                                           int64_t npix, int C, int HW, int CP, int apply_tanh) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW;
        const int64_t hw = i - b * HW;
        for (int c0 = 0; c0 < C; c0 += 4) {           // CP % 4 == 0: one 8- / 16-byte load per 4 channels
            const float4 q = load4<DT>(x, i * CP + c0);
            const float v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + k;
                if (c < C) y[(b * C + c) * HW + hw] = apply_tanh ? tanhf(v[k]) : v[k];
            }
        }
    }
}

// Generator output in one pass (vaegan_code.py:83,92): recon = tanh(pre) as NCHW f32 (for the MSE and tanh') AND
// recon_noisy = recon + sigma * eps in the Discriminator's NHWC input layout.
template <int DT>
__global__ __launch_bounds__(256) void nhwc_tanh_noisy_kernel(const void* __restrict__ x, float* __restrict__ y,
                                                              const NoiseSrc eps, float sigma,
                                                              void* __restrict__ yn, int64_t npix, int C, int HW,
                                                              int CP) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW;
        const int64_t hw = i - b * HW;
        for (int c0 = 0; c0 < CP; c0 += 4) {
            const float4 q = load4<DT>(x, i * CP + c0);
            const float v[4] = {q.x, q.y, q.z, q.w};
            float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + k;
                if (c < C) {
                    const int64_t dst = (b * C + c) * HW + hw;
                    const float t = tanhf(v[k]);
                    y[dst] = t;
                    o[k] = t + sigma * noise_at(eps, dst);
                }
            }
            store4<DT>(yn, i * CP + c0, float4{o[0], o[1], o[2], o[3]});
        }
    }
}

template <int DT>
__global__ __launch_bounds__(256) void nchw_grad_to_nhwc_kernel(const float* __restrict__ dy,
                                                                const float* __restrict__ t, void* __restrict__ dx,
                                                                int64_t npix, int C, int HW, int CP,
                                                                const void* __restrict__ add) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW;
        const int64_t hw = i - b * HW;
        for (int c0 = 0; c0 < CP; c0 += 4) {
            float v[4];
            float a[4] = {0.f, 0.f, 0.f, 0.f};
            if (add) { const float4 q = load4<DT>(add, i * CP + c0); a[0] = q.x; a[1] = q.y; a[2] = q.z; a[3] = q.w; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + k;
                v[k] = 0.f;
                if (c < C) {
                    const int64_t src = (b * C + c) * HW + hw;
                    v[k] = dy[src];
                    if (add) v[k] = v[k] + a[k];          // second gradient branch, already in the engine layout
                    if (t) { const float tv = t[src]; v[k] = v[k] * (1.f - tv * tv); }
                }
            }
            store4<DT>(dx, i * CP + c0, float4{v[0], v[1], v[2], v[3]});
        }
    }
}

// ---- reparameterisation / KL ----------------------------------------------------------------------
template <int DT>
__global__ void reparam_fwd_kernel(const void* __restrict__ mulv, const NoiseSrc eps, void* __restrict__ z,
                                   float* __restrict__ lvc, int B, int L, int MP, int ZP) {
    const int total = B * ZP;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / ZP, j = i - b * ZP;
        float zv = 0.f;
        if (j < L) {
            const float mu = load1<DT>(mulv, (int64_t)b * MP + j);
            float lv = load1<DT>(mulv, (int64_t)b * MP + L + j);
            lv = fminf(fmaxf(lv, -10.f), 10.f);
            lvc[b * L + j] = lv;
            zv = mu + expf(0.5f * lv) * noise_at(eps, b * L + j);
        }
        store1<DT>(z, i, zv);
    }
}

template <int DT>
__global__ __launch_bounds__(1024) void kl_kernel(const void* __restrict__ mulv, const float* __restrict__ lvc, int B,
                                                  int L, int MP, float divisor, float* __restrict__ out,
                                                  const float* __restrict__ mse_ws, int mse_nparts, double mse_n,
                                                  float* __restrict__ mse_loss) {
    // one workgroup (a single scalar result, fixed summation order); 1024 threads keep the B*L-element loop short
    __shared__ double red[16];
    if (mse_ws != nullptr && threadIdx.x < 64) {
        // the second stage of vg_mse_forward_backward rides along (its own launch was 4.7 us for 1024 partials): wave 0,
        // exactly mse_final_kernel's arithmetic and order
        double s = 0.0;
        for (int i = threadIdx.x; i < mse_nparts; i += 64) s += (double)mse_ws[i];
        s = wave_sum_d(s);
        if (threadIdx.x == 0) mse_loss[0] = (float)(s / mse_n);
    }
    double s = 0.0;
    const int total = B * L;
    int i = threadIdx.x;
    for (; i + 3 * (int)blockDim.x < total; i += 4 * blockDim.x) {        // four element pairs in flight per thread
        float mu[4], lv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = i + u * blockDim.x;
            const int b = e / L, j = e - b * L;
            mu[u] = load1<DT>(mulv, (int64_t)b * MP + j);
            lv[u] = lvc[e];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) s += (double)(1.f + lv[u] - mu[u] * mu[u] - expf(lv[u]));
    }
    for (; i < total; i += blockDim.x) {
        const int b = i / L, j = i - b * L;
        const float mu = load1<DT>(mulv, (int64_t)b * MP + j);
        const float lv = lvc[i];
        s += (double)(1.f + lv - mu * mu - expf(lv));
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
        out[0] = (float)(-0.5 * t) / divisor;
    }
}

template <int DT>
__global__ void reparam_kl_bwd_kernel(const void* __restrict__ mulv, const float* __restrict__ lvc,
                                      const NoiseSrc eps, const void* __restrict__ dz, float kl_scale,
                                      void* __restrict__ dmulv, int B, int L, int MP, int ZP) {
    const int total = B * MP;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / MP, j = i - b * MP;
        float g = 0.f;
        if (j < L) {                                   // d mu = dz + kl_scale * mu
            const float mu = load1<DT>(mulv, (int64_t)b * MP + j);
            g = load1<DT>(dz, (int64_t)b * ZP + j) + kl_scale * mu;
        } else if (j < 2 * L) {                        // d logvar (through the clamp, inclusive bounds)
            const int jj = j - L;
            const float raw = load1<DT>(mulv, (int64_t)b * MP + j);
            if (raw >= -10.f && raw <= 10.f) {
                const float lv = lvc[b * L + jj];
                const float d = load1<DT>(dz, (int64_t)b * ZP + jj);
                g = d * (0.5f * expf(0.5f * lv) * noise_at(eps, b * L + jj)) + kl_scale * 0.5f * (expf(lv) - 1.f);
            }
        }
        store1<DT>(dmulv, i, g);
    }
}

// ---- Discriminator head: Conv2d(C,1,k=HxW) + Sigmoid on the final feature map -------------------
template <int DT>
__global__ __launch_bounds__(256) void dot_sigmoid_fwd_kernel(const void* __restrict__ x, const void* __restrict__ w,
                                                              float* __restrict__ p, int B, int K) {
    // one workgroup per image (a single wave per image was a 32-deep chain of dependent loads: 13 us for 0.1 MB);
    // four loads in flight per lane, fixed-order combine (wave shuffle, then 4 partials through LDS)
    __shared__ float red[4];
    const int img = blockIdx.x;
    const int tid = threadIdx.x;
    float s = 0.f;
    int k = tid * 4;
    for (; k + 3 * 1024 < K; k += 4 * 1024) {
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = load4<DT>(x, (int64_t)img * K + k + u * 1024);
            b[u] = load4<DT>(w, k + u * 1024);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) s += a[u].x * b[u].x + a[u].y * b[u].y + a[u].z * b[u].z + a[u].w * b[u].w;
    }
    for (; k < K; k += 1024) {
        const float4 a = load4<DT>(x, (int64_t)img * K + k);
        const float4 b = load4<DT>(w, k);
        s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        const float t = (red[0] + red[1]) + (red[2] + red[3]);
        p[img] = 1.f / (1.f + expf(-t));
    }
}

template <int DT>
__global__ __launch_bounds__(256) void dot_sigmoid_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                                                              const void* __restrict__ w, void* __restrict__ dx,
                                                              float* __restrict__ dlogit, int B, int K) {
    const int64_t nvec = (int64_t)B * K / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)((i * 4) / K);
        const int k = (int)((i * 4) - (int64_t)b * K);
        const float pv = p[b];
        const float dl = dp[b] * (pv * (1.f - pv));          // sigmoid_backward: grad * (1 - y) * y
        if (k == 0) dlogit[b] = dl;
        if (dx) {
            float4 wv = load4<DT>(w, k);
            wv.x *= dl; wv.y *= dl; wv.z *= dl; wv.w *= dl;
            store4<DT>(dx, i * 4, wv);
        }
    }
}

template <int DT>
__global__ __launch_bounds__(1024) void dot_wgrad_kernel(const void* __restrict__ x, const float* __restrict__ dlogit,
                                                         float* __restrict__ dw, int B, int K, int C, int HW,
                                                         int accumulate) {
    // 64 columns (NHWC-flattened k) per block x 16 batch lanes (4 lanes made each thread walk B/4 images one
    // dependent load at a time: 14 us); four loads in flight, fixed-order combine through LDS
    __shared__ float red[16][64];
    const int kl = threadIdx.x & 63, bl = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + kl;
    float s = 0.f;
    if (k < K) {
        int b = bl;
        for (; b + 48 < B; b += 64) {
            float v[4], g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { v[u] = load1<DT>(x, (int64_t)(b + 16 * u) * K + k); g[u] = dlogit[b + 16 * u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) s += g[u] * v[u];
        }
        for (; b < B; b += 16) s += dlogit[b] * load1<DT>(x, (int64_t)b * K + k);
    }
    red[bl][kl] = s;
    __syncthreads();
    if (bl != 0 || k >= K) return;
    s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += red[r][kl];
    const int hw = k / C, c = k - hw * C;
    float* dst = dw + (int64_t)c * HW + hw;                   // reference layout [1][C][kh][kw]
    *dst = accumulate ? (*dst + s) : s;
}

// One sample's BCE term, -(t * max(log p, -100) + (1 - t) * max(log(1 - p), -100)) (torch.nn.BCELoss),
// with floating-point contraction OFF: the three kernels that evaluate it (bce_kernel, bce_pair_kernel, head_bwd_kernel)
// must round identically, and hipcc is free to contract a*b + c*d into an fma around EITHER product.
__device__ __forceinline__ float bce_term(float pv, float target) {
#pragma clang fp contract(off)
    const float l1 = fmaxf(logf(pv), -100.f);
    const float l2 = fmaxf(logf(1.f - pv), -100.f);
    const float a = target * l1;
    const float b = (1.f - target) * l2;
    return -(a + b);
}

// ---- Discriminator head, backward of BCE(sigmoid(dot)) in ONE launch (round 3) --------------------------------------
// What vaegan_code.py:99-104 / :115,133 run behind the head: bce (loss + dL/dp), sigmoid backward, the head's data
// gradient dx[b,k] = dlogit[b] * w[k] and its weight gradient dw[k] = sum_b dlogit[b] * x[b,k] were three launches
// (bce[_pair]_kernel, dot_sigmoid_bwd_kernel, dot_wgrad_kernel).  Here a workgroup owns 64 columns k for ALL rows: it
// re-derives dlogit[0..groups*B) into LDS (B*groups floats; the arithmetic of the three kernels, operation for operation),
// forms its dw columns with dot_wgrad_kernel's lane mapping and summation order and writes its dx columns; workgroup 0
// also writes the loss with bce_pair_kernel's mapping -- every output is bit-identical to the three-launch path.
constexpr int HB_MAXROWS = 4096;
template <int DT>
__global__ __launch_bounds__(1024) void head_bwd_kernel(const float* __restrict__ p, const void* __restrict__ x,
                                                        const void* __restrict__ w, void* __restrict__ dx,
                                                        float* __restrict__ dw, float* __restrict__ dlogit_out, int B,
                                                        int groups, float t0, float t1, float gscale,
                                                        float* __restrict__ loss, int accumulate_loss, int accumulate_dw,
                                                        int K, int C, int HW) {
    __shared__ float dl[HB_MAXROWS];
    __shared__ float red[16][64];
    __shared__ double lred[2][4];
    const int R = B * groups;
    for (int r = threadIdx.x; r < R; r += 1024) {
        const float target = r < B ? t0 : t1;
        const float pv = p[r];
        const float dpv = gscale * ((pv - target) / fmaxf((1.f - pv) * pv, 1e-12f)) / (float)B;     // bce_kernel
        const float d = dpv * (pv * (1.f - pv));                                                     // dot_sigmoid_bwd_kernel
        dl[r] = d;
        if (blockIdx.x == 0 && dlogit_out) dlogit_out[r] = d;
    }
    if (blockIdx.x == 0 && threadIdx.x < 256) {             // the loss: bce_pair_kernel's 256-thread mapping per half
        for (int h = 0; h < groups; ++h) {
            const float target = h == 0 ? t0 : t1;
            double sacc = 0.0;
            for (int b = threadIdx.x; b < B; b += 256) {
                sacc += (double)bce_term(p[h * B + b], target);
            }
            sacc = wave_sum_d(sacc);
            if ((threadIdx.x & 63) == 0) lred[h][threadIdx.x >> 6] = sacc;
        }
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const float v0 = (float)((lred[0][0] + lred[0][1] + lred[0][2] + lred[0][3]) / (double)B);
        float out = accumulate_loss ? loss[0] + v0 : v0;
        if (groups == 2) out = out + (float)((lred[1][0] + lred[1][1] + lred[1][2] + lred[1][3]) / (double)B);
        loss[0] = out;
    }
    // ---- dw: 64 columns x 16 batch lanes, four loads in flight, fixed-order combine (dot_wgrad_kernel) ----
    const int kl = threadIdx.x & 63, bl = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + kl;
    if (dw != nullptr) {
        float s = 0.f;
        if (k < K) {
            int b = bl;
            for (; b + 48 < R; b += 64) {
                float v[4], g[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { v[u] = load1<DT>(x, (int64_t)(b + 16 * u) * K + k); g[u] = dl[b + 16 * u]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) s += g[u] * v[u];
            }
            for (; b < R; b += 16) s += dl[b] * load1<DT>(x, (int64_t)b * K + k);
        }
        red[bl][kl] = s;
        __syncthreads();
        if (bl == 0 && k < K) {
            s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += red[r][kl];
            const int hw = k / C, c = k - hw * C;
            float* dst = dw + (int64_t)c * HW + hw;               // reference layout [1][C][kh][kw]
            *dst = accumulate_dw ? (*dst + s) : s;
        }
    }
    // ---- dx: this workgroup's 64 columns of every row, 4 columns (one 8 / 16-byte store) per thread ----
    if (dx != nullptr) {
        const int kq = threadIdx.x & 15, bq = threadIdx.x >> 4;
        const int k4 = blockIdx.x * 64 + kq * 4;
        if (k4 < K) {
            const float4 wv = load4<DT>(w, k4);
            for (int b = bq; b < R; b += 64) {
                const float d = dl[b];
                float4 o = wv;
                o.x *= d; o.y *= d; o.z *= d; o.w *= d;
                store4<DT>(dx, (int64_t)b * K + k4, o);
            }
        }
    }
}

// ---- losses -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ p, float target, int B, float gscale,
                                                  float* __restrict__ loss, int accumulate, float* __restrict__ dp) {
    __shared__ double red[4];
    double s = 0.0;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float pv = p[b];
        s += (double)bce_term(pv, target);
        if (dp) dp[b] = gscale * ((pv - target) / fmaxf((1.f - pv) * pv, 1e-12f)) / (float)B;
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (float)((red[0] + red[1] + red[2] + red[3]) / (double)B);
        loss[0] = accumulate ? loss[0] + v : v;
    }
}

// The Discriminator loss of a grouped pass (vaegan_code.py:98-103): p = [real half | fake half], loss = BCE(real half, t0) +
// BCE(fake half, t1) -- two bce_kernel launches' worth in one (same thread mapping per half, same order of the f64 sums,
// loss = v0 + v1 in f32 as the accumulating second launch computed it: bit-identical).
__global__ __launch_bounds__(256) void bce_pair_kernel(const float* __restrict__ p, float t0, float t1, int B, float gscale,
                                                       float* __restrict__ loss, int accumulate, float* __restrict__ dp) {
    __shared__ double red[2][4];
    float out = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float target = h == 0 ? t0 : t1;
        double s = 0.0;
        for (int b = threadIdx.x; b < B; b += blockDim.x) {
            const float pv = p[h * B + b];
            s += (double)bce_term(pv, target);
            if (dp) dp[h * B + b] = gscale * ((pv - target) / fmaxf((1.f - pv) * pv, 1e-12f)) / (float)B;
        }
        s = wave_sum_d(s);
        if ((threadIdx.x & 63) == 0) red[h][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v0 = (float)((red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (double)B);
        const float v1 = (float)((red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (double)B);
        out = accumulate ? loss[0] + v0 : v0;
        loss[0] = out + v1;
    }
}

// WGAN critic / generator losses (gan_code.py:306-315, :328): loss (+)= sign * mean(p), dp = sign * gscale / B
__global__ __launch_bounds__(256) void mean_loss_kernel(const float* __restrict__ p, float sign, int B, float gscale,
                                                        float* __restrict__ loss, int accumulate,
                                                        float* __restrict__ dp) {
    __shared__ double red[4];
    double s = 0.0;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        s += (double)p[b];
        if (dp) dp[b] = sign * gscale / (float)B;
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = sign * (float)((red[0] + red[1] + red[2] + red[3]) / (double)B);
        loss[0] = accumulate ? loss[0] + v : v;
    }
}

// WGAN weight clipping (gan_code.py:320-321) over an optimizer's flat parameter buffer
__global__ __launch_bounds__(256) void clamp_kernel(float* __restrict__ p, int64_t n, float lo, float hi) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = fminf(fmaxf(p[i], lo), hi);
}

__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          int64_t n, float gcoef, float* __restrict__ d_a,
                                                          float* __restrict__ ws) {
    __shared__ double red[4];
    double s = 0.0;
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 x = reinterpret_cast<const float4*>(a)[i];
        const float4 y = reinterpret_cast<const float4*>(b)[i];
        float4 d;
        d.x = x.x - y.x; d.y = x.y - y.y; d.z = x.z - y.z; d.w = x.w - y.w;
        s += (double)(d.x * d.x) + (double)(d.y * d.y) + (double)(d.z * d.z) + (double)(d.w * d.w);
        if (d_a) {
            d.x *= gcoef; d.y *= gcoef; d.z *= gcoef; d.w *= gcoef;
            reinterpret_cast<float4*>(d_a)[i] = d;
        }
    }
    if (blockIdx.x == 0) {
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
            const float d = a[i] - b[i];
            s += (double)(d * d);
            if (d_a) d_a[i] = d * gcoef;
        }
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) ws[blockIdx.x] = (float)(red[0] + red[1] + red[2] + red[3]);
}

__global__ void mse_final_kernel(const float* __restrict__ ws, int nparts, double n, float* __restrict__ loss) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) s += (double)ws[i];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) loss[0] = (float)(s / n);
}

__global__ __launch_bounds__(256) void axpy_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                   float alpha, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = a[i] + alpha * b[i];
}

// ---- SSIM (denoise evaluation, vaegan_code.py:143,174): gaussian 11x11 (sigma 1.5), reflect padding, per-channel
// maps, border of 5 cropped before averaging (the torchmetrics recipe restated in oracle/vaegan_ref.py:ssim;
// parity unpinned -- torchmetrics is not installed).  Inputs NCHW f32 in [-1,1], mapped to [0,1] on the fly.
__global__ __launch_bounds__(256) void ssim_kernel(const float* __restrict__ a, const float* __restrict__ b, int BC, int H,
                                                   int W, float* __restrict__ partial) {
    __shared__ float g[11];
    __shared__ double red[4];
    if (threadIdx.x < 11) {
        float s = 0.f;
        for (int k = 0; k < 11; ++k) s += expf(-((k - 5) * (k - 5)) / (2.f * 1.5f * 1.5f));
        g[threadIdx.x] = expf(-((int(threadIdx.x) - 5) * (int(threadIdx.x) - 5)) / (2.f * 1.5f * 1.5f)) / s;
    }
    __syncthreads();
    const int IH = H - 10, IW = W - 10;                     // interior kept after the crop
    const int64_t total = (int64_t)BC * IH * IW;
    const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % IW) + 5;
        const int y = (int)((i / IW) % IH) + 5;
        const int64_t pl = i / ((int64_t)IW * IH);
        const float* pa = a + pl * H * W;
        const float* pb = b + pl * H * W;
        float ma = 0.f, mb = 0.f, saa = 0.f, sbb = 0.f, sab = 0.f;
        for (int dy = -5; dy <= 5; ++dy) {
            const int yy = y + dy;                              // interior pixels never need the reflect
            const float gy = g[dy + 5];
            for (int dx = -5; dx <= 5; ++dx) {
                const float w = gy * g[dx + 5];
                const float va = (pa[yy * W + x + dx] + 1.f) * 0.5f;
                const float vb = (pb[yy * W + x + dx] + 1.f) * 0.5f;
                ma += w * va; mb += w * vb; saa += w * va * va; sbb += w * vb * vb; sab += w * va * vb;
            }
        }
        const float vaa = saa - ma * ma, vbb = sbb - mb * mb, vab = sab - ma * mb;
        acc += (double)(((2.f * ma * mb + c1) * (2.f * vab + c2)) / ((ma * ma + mb * mb + c1) * (vaa + vbb + c2)));
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (float)(red[0] + red[1] + red[2] + red[3]);
}

inline int blocks_for(int64_t n, int cap = 4096) {
    int64_t b = (n + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

#define DISPATCH_DT(KERNEL, GRID, BLOCK, STREAM, ...)                                              \
    do {                                                                                           \
        if (dtype == VG_F32) hipLaunchKernelGGL(KERNEL<VG_F32>, GRID, BLOCK, 0, STREAM, __VA_ARGS__);   \
        else hipLaunchKernelGGL(KERNEL<VG_BF16>, GRID, BLOCK, 0, STREAM, __VA_ARGS__);                  \
    } while (0)

#define CHECK_DT() VG_CHECK_ARG(dtype == VG_F32 || dtype == VG_BF16, VG_ENOSUP)

extern "C" int vg_nchw_to_nhwc(const float* x, const float* eps, float sigma, void* y, int B, int C, int H, int W,
                               int CP, int dtype, void* stream) {
    CHECK_DT();
    VG_CHECK_ARG(x && y && B > 0 && C > 0 && H > 0 && W > 0 && CP >= C && CP % 4 == 0, VG_EINVAL);
    const int64_t npix = (int64_t)B * H * W;
    DISPATCH_DT(nchw_to_nhwc_kernel, dim3(blocks_for(npix)), dim3(256), vg_stream(stream), x, (NoiseSrc{eps, nullptr, 0u}), sigma, y, npix, C,
                H * W, CP, -3.0e38f, 3.0e38f, (float*)nullptr);
    return VG_LAUNCH_RC();
}

extern "C" int vg_nchw_to_nhwc_rng(const float* x, const uint64_t* rng, int draw, float sigma, void* y, int B, int C,
                                   int H, int W, int CP, int dtype, void* stream) {
    CHECK_DT();
    VG_CHECK_ARG(x && rng && y && draw >= 0 && draw < 256 && B > 0 && C > 0 && H > 0 && W > 0 && CP >= C && CP % 4 == 0, VG_EINVAL);
    const int64_t npix = (int64_t)B * H * W;
    if (dtype == VG_BF16 && CP == 8 && C <= 4 && (H * W) % 4 == 0 && vg_aligned16(x) && vg_aligned16(y)) {
        hipLaunchKernelGGL(nchw_to_nhwc_x4_kernel<VG_BF16>, dim3(blocks_for(npix / 4)), dim3(256), 0, vg_stream(stream), x,
                           (NoiseSrc{nullptr, (const unsigned long long*)rng, (uint32_t)draw}), sigma, y, npix / 4, C, H * W,
                           -3.0e38f, 3.0e38f, (float*)nullptr);
        return VG_LAUNCH_RC();
    }
    DISPATCH_DT(nchw_to_nhwc_kernel, dim3(blocks_for(npix)), dim3(256), vg_stream(stream), x,
                (NoiseSrc{nullptr, (const unsigned long long*)rng, (uint32_t)draw}), sigma, y, npix, C, H * W, CP, -3.0e38f, 3.0e38f,
                (float*)nullptr);
    return VG_LAUNCH_RC();
}

// x (NCHW f32) -> y_noisy = x + sigma * eps AND y_plain = x, both NHWC bf16 with 8 channels, in one pass over x: the
// Encoder's input and the Discriminator's noisy real batch (vaegan_code.py:74 and :91) are the same image batch.
// eps: injected noise tensor or NULL; rng / draw: in-kernel generator (exactly one of eps / rng).  VG_ENOSUP where the
// four-pixel kernel does not apply (the caller then converts twice).
extern "C" int vg_nchw_to_nhwc_pair(const float* x, const float* eps, const uint64_t* rng, int draw, float sigma,
                                    void* y_noisy, void* y_plain, int B, int C, int H, int W, int CP, int dtype,
                                    void* stream) {
    VG_CHECK_ARG(x && y_noisy && y_plain && ((eps != nullptr) != (rng != nullptr)) && draw >= 0 && draw < 256 && B > 0 &&
                 C > 0 && H > 0 && W > 0, VG_EINVAL);
    if (!(dtype == VG_BF16 && CP == 8 && C <= 4 && (H * W) % 4 == 0 && vg_aligned16(x) && vg_aligned16(y_noisy) &&
          vg_aligned16(y_plain) && (eps == nullptr || vg_aligned16(eps))))
        return VG_ENOSUP;
    const int64_t npix = (int64_t)B * H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_x4_kernel<VG_BF16>, dim3(blocks_for(npix / 4)), dim3(256), 0, vg_stream(stream), x,
                       (NoiseSrc{eps, (const unsigned long long*)rng, (uint32_t)draw}), sigma, y_noisy, npix / 4, C, H * W,
                       -3.0e38f, 3.0e38f, (float*)nullptr, y_plain);
    return VG_LAUNCH_RC();
}

extern "C" int vg_rng_advance(uint64_t* rng, void* stream) {
    VG_CHECK_ARG(rng != nullptr, VG_EINVAL);
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, vg_stream(stream), (unsigned long long*)rng);
    return VG_LAUNCH_RC();
}

extern "C" int vg_randn(float* out, int64_t n, const uint64_t* rng, int draw, void* stream) {
    VG_CHECK_ARG(out && rng && n > 0 && draw >= 0 && draw < 256, VG_EINVAL);
    hipLaunchKernelGGL(randn_fill_kernel, dim3(blocks_for(n)), dim3(256), 0, vg_stream(stream), out, n,
                       NoiseSrc{nullptr, (const unsigned long long*)rng, (uint32_t)draw});
    return VG_LAUNCH_RC();
}

extern "C" int vg_cast_fp8(const void* x_bf16, void* y_fp8, int64_t n, int shift, void* stream) {
    VG_CHECK_ARG(x_bf16 && y_fp8 && n > 0 && n % 8 == 0 && shift >= -16 && shift <= 16, VG_EINVAL);
    VG_CHECK_ARG(vg_aligned16(x_bf16) && (reinterpret_cast<uintptr_t>(y_fp8) & 7u) == 0, VG_EALIGN);
    hipLaunchKernelGGL(cast_fp8_kernel, dim3(blocks_for(n / 8)), dim3(256), 0, vg_stream(stream),
                       reinterpret_cast<const u32x4*>(x_bf16), reinterpret_cast<uint2*>(y_fp8), n / 8, ldexpf(1.f, shift));
    return VG_LAUNCH_RC();
}

extern "C" int vg_memset_zero(void* p, int64_t nbytes, void* stream) {
    VG_CHECK_ARG(p && nbytes > 0, VG_EINVAL);
    return (int)hipMemsetAsync(p, 0, (size_t)nbytes, vg_stream(stream));
}

extern "C" int vg_gather_normalize_u8(const uint8_t* images, int64_t N, const int64_t* idx, int B, int C, int H, int W,
                                      float* out, void* stream) {
    VG_CHECK_ARG(images && idx && out && N > 0 && B > 0 && C > 0 && H > 0 && W > 0, VG_EINVAL);
    const int64_t npix = (int64_t)B * H * W;
    hipLaunchKernelGGL(gather_u8_kernel, dim3(blocks_for(npix)), dim3(256), 0, vg_stream(stream), images, idx, npix, C,
                       H * W, N, out);
    return VG_LAUNCH_RC();
}

extern "C" int vg_noisy_clamp_to_nhwc(const float* x, const float* eps, float sigma, float lo, float hi, void* y,
                                      float* y_nchw, int B, int C, int H, int W, int CP, int dtype, void* stream) {
    CHECK_DT();
    VG_CHECK_ARG(x && eps && y && B > 0 && C > 0 && H > 0 && W > 0 && CP >= C && CP % 4 == 0 && lo <= hi, VG_EINVAL);
    const int64_t npix = (int64_t)B * H * W;
    DISPATCH_DT(nchw_to_nhwc_kernel, dim3(blocks_for(npix)), dim3(256), vg_stream(stream), x, (NoiseSrc{eps, nullptr, 0u}), sigma, y, npix, C,
                H * W, CP, lo, hi, y_nchw);
    return VG_LAUNCH_RC();
}

extern "C" int vg_nhwc_to_nchw(const void* x, float* y, int B, int C, int H, int W, int CP, int apply_tanh, int dtype,
                               void* stream) {
    CHECK_DT();
    VG_CHECK_ARG(x && y && B > 0 && C > 0 && H > 0 && W > 0 && CP >= C && CP % 4 == 0, VG_EINVAL);
    const int64_t npix = (int64_t)B * H * W;
    DISPATCH_DT(nhwc_to_nchw_kernel, dim3(blocks_for(npix)), dim3(256), vg_stream(stream), x, y, npix, C, H * W, CP,
                apply_tanh);
    return VG_LAUNCH_RC();
}

extern "C" int vg_nhwc_tanh_to_nchw_noisy(const void* x, float* y_nchw, const float* eps, float sigma, void* y_noisy_nhwc,
                                          int B, int C, int H, int W, int CP, int dtype, void* stream) {
    CHECK_DT();
    VG_CHECK_ARG(x && y_nchw && eps && y_noisy_nhwc && B > 0 && C > 0 && H > 0 && W > 0 && CP >= C && CP % 4 == 0,
                 VG_EINVAL);
    const int64_t npix = (int64_t)B * H * W;
    DISPATCH_DT(nhwc_tanh_noisy_kernel, dim3(blocks_for(npix)), dim3(256), vg_stream(stream), x, y_nchw, (NoiseSrc{eps, nullptr, 0u}), sigma,
                y_noisy_nhwc, npix, C, H * W, CP);
    return VG_LAUNCH_RC();
}

extern "C" int vg_nhwc_tanh_to_nchw_noisy_rng(const void* x, float* y_nchw, const uint64_t* rng, int draw, float sigma,
                                              void* y_noisy_nhwc, int B, int C, int H, int W, int CP, int dtype,
                                              void* stream) {
    CHECK_DT();
    VG_CHECK_ARG(x && y_nchw && rng && y_noisy_nhwc && draw >= 0 && draw < 256 && B > 0 && C > 0 && H > 0 && W > 0 &&
                 CP >= C && CP % 4 == 0, VG_EINVAL);
    const int64_t npix = (int64_t)B * H * W;
    DISPATCH_DT(nhwc_tanh_noisy_kernel, dim3(blocks_for(npix)), dim3(256), vg_stream(stream), x, y_nchw,
                (NoiseSrc{nullptr, (const unsigned long long*)rng, (uint32_t)draw}), sigma, y_noisy_nhwc, npix, C, H * W, CP);
    return VG_LAUNCH_RC();
}

extern "C" int vg_nchw_grad_add_to_nhwc(const float* dy, const void* add_nhwc, const float* tanh_out, void* dx, int B,
                                        int C, int H, int W, int CP, int dtype, void* stream) {
    CHECK_DT();
    VG_CHECK_ARG(dy && add_nhwc && dx && B > 0 && C > 0 && H > 0 && W > 0 && CP >= C && CP % 4 == 0, VG_EINVAL);
    const int64_t npix = (int64_t)B * H * W;
    DISPATCH_DT(nchw_grad_to_nhwc_kernel, dim3(blocks_for(npix)), dim3(256), vg_stream(stream), dy, tanh_out, dx, npix,
                C, H * W, CP, add_nhwc);
    return VG_LAUNCH_RC();
}

extern "C" int vg_nchw_grad_to_nhwc(const float* dy, const float* tanh_out, void* dx, int B, int C, int H, int W,
                                    int CP, int dtype, void* stream) {
    CHECK_DT();
    VG_CHECK_ARG(dy && dx && B > 0 && C > 0 && H > 0 && W > 0 && CP >= C && CP % 4 == 0, VG_EINVAL);
    const int64_t npix = (int64_t)B * H * W;
    DISPATCH_DT(nchw_grad_to_nhwc_kernel, dim3(blocks_for(npix)), dim3(256), vg_stream(stream), dy, tanh_out, dx, npix,
                C, H * W, CP, (const void*)nullptr);
    return VG_LAUNCH_RC();
}

extern "C" int vg_reparam_forward(const void* mulv, const float* eps, void* z, float* lv_clamped, int B, int L, int MP,
                                  int ZP, int dtype, void* stream) {
    CHECK_DT();
    VG_CHECK_ARG(mulv && eps && z && lv_clamped && B > 0 && L > 0 && MP >= 2 * L && ZP >= L, VG_EINVAL);
    DISPATCH_DT(reparam_fwd_kernel, dim3(blocks_for((int64_t)B * ZP)), dim3(256), vg_stream(stream), mulv, (NoiseSrc{eps, nullptr, 0u}), z,
                lv_clamped, B, L, MP, ZP);
    return VG_LAUNCH_RC();
}

extern "C" int vg_reparam_forward_rng(const void* mulv, const uint64_t* rng, int draw, void* z, float* lv_clamped, int B,
                                      int L, int MP, int ZP, int dtype, void* stream) {
    CHECK_DT();
    VG_CHECK_ARG(mulv && rng && z && lv_clamped && draw >= 0 && draw < 256 && B > 0 && L > 0 && MP >= 2 * L && ZP >= L, VG_EINVAL);
    DISPATCH_DT(reparam_fwd_kernel, dim3(blocks_for((int64_t)B * ZP)), dim3(256), vg_stream(stream), mulv,
                (NoiseSrc{nullptr, (const unsigned long long*)rng, (uint32_t)draw}), z, lv_clamped, B, L, MP, ZP);
    return VG_LAUNCH_RC();
}

extern "C" int vg_kl_forward(const void* mulv, const float* lv_clamped, int B, int L, int MP, float divisor, float* out,
                             int dtype, void* stream) {
    CHECK_DT();
    VG_CHECK_ARG(mulv && lv_clamped && out && B > 0 && L > 0 && MP >= 2 * L && divisor != 0.f, VG_EINVAL);
    DISPATCH_DT(kl_kernel, dim3(1), dim3(1024), vg_stream(stream), mulv, lv_clamped, B, L, MP, divisor, out,
                (const float*)nullptr, 0, 1.0, (float*)nullptr);
    return VG_LAUNCH_RC();
}

extern "C" int vg_kl_forward_mse_final(const void* mulv, const float* lv_clamped, int B, int L, int MP, float divisor,
                                       float* out, const float* mse_ws, int mse_nparts, int64_t mse_n, float* mse_loss,
                                       int dtype, void* stream) {
    CHECK_DT();
    VG_CHECK_ARG(mulv && lv_clamped && out && B > 0 && L > 0 && MP >= 2 * L && divisor != 0.f, VG_EINVAL);
    VG_CHECK_ARG(mse_ws && mse_loss && mse_nparts > 0 && mse_n > 0, VG_EINVAL);
    DISPATCH_DT(kl_kernel, dim3(1), dim3(1024), vg_stream(stream), mulv, lv_clamped, B, L, MP, divisor, out, mse_ws,
                mse_nparts, (double)mse_n, mse_loss);
    return VG_LAUNCH_RC();
}

extern "C" int vg_mse_partial(const float* a, const float* b, int64_t n, float gscale, float* d_a, float* ws,
                              int ws_capacity, int* nparts_out, void* stream) {
    VG_CHECK_ARG(a && b && ws && nparts_out && n > 0 && ws_capacity >= 1, VG_EINVAL);
    int blocks = blocks_for(n / 4 + 1, 1024);
    if (blocks > ws_capacity) blocks = ws_capacity;
    const float gcoef = (float)((double)gscale * 2.0 / (double)n);
    hipLaunchKernelGGL(mse_partial_kernel, dim3(blocks), dim3(256), 0, vg_stream(stream), a, b, n, gcoef, d_a, ws);
    *nparts_out = blocks;
    return VG_LAUNCH_RC();
}

extern "C" int vg_reparam_kl_backward(const void* mulv, const float* lv_clamped, const float* eps, const void* dz,
                                      float kl_scale, void* dmulv, int B, int L, int MP, int ZP, int dtype,
                                      void* stream) {
    CHECK_DT();
    VG_CHECK_ARG(mulv && lv_clamped && eps && dz && dmulv && B > 0 && L > 0 && MP >= 2 * L && ZP >= L, VG_EINVAL);
    DISPATCH_DT(reparam_kl_bwd_kernel, dim3(blocks_for((int64_t)B * MP)), dim3(256), vg_stream(stream), mulv,
                lv_clamped, (NoiseSrc{eps, nullptr, 0u}), dz, kl_scale, dmulv, B, L, MP, ZP);
    return VG_LAUNCH_RC();
}

extern "C" int vg_reparam_kl_backward_rng(const void* mulv, const float* lv_clamped, const uint64_t* rng, int draw,
                                          const void* dz, float kl_scale, void* dmulv, int B, int L, int MP, int ZP,
                                          int dtype, void* stream) {
    CHECK_DT();
    VG_CHECK_ARG(mulv && lv_clamped && rng && dz && dmulv && draw >= 0 && draw < 256 && B > 0 && L > 0 && MP >= 2 * L &&
                 ZP >= L, VG_EINVAL);
    DISPATCH_DT(reparam_kl_bwd_kernel, dim3(blocks_for((int64_t)B * MP)), dim3(256), vg_stream(stream), mulv,
                lv_clamped, (NoiseSrc{nullptr, (const unsigned long long*)rng, (uint32_t)draw}), dz, kl_scale, dmulv, B, L, MP, ZP);
    return VG_LAUNCH_RC();
}

extern "C" int vg_dot_sigmoid_forward(const void* x, const void* w, float* p, int B, int K, int dtype, void* stream) {
    CHECK_DT();
    VG_CHECK_ARG(x && w && p && B > 0 && K > 0 && K % 4 == 0, VG_EINVAL);
    DISPATCH_DT(dot_sigmoid_fwd_kernel, dim3(B), dim3(256), vg_stream(stream), x, w, p, B, K);
    return VG_LAUNCH_RC();
}

extern "C" int vg_dot_sigmoid_backward(const float* p, const float* dp, const void* w, void* dx, float* dlogit, int B,
                                       int K, int dtype, void* stream) {
    CHECK_DT();
    VG_CHECK_ARG(p && dp && w && dlogit && B > 0 && K > 0 && K % 4 == 0, VG_EINVAL);
    DISPATCH_DT(dot_sigmoid_bwd_kernel, dim3(blocks_for((int64_t)B * K / 4)), dim3(256), vg_stream(stream), p, dp, w, dx,
                dlogit, B, K);
    return VG_LAUNCH_RC();
}

extern "C" int vg_dot_wgrad(const void* x, const float* dlogit, float* dw, int B, int K, int C, int HW, int accumulate,
                            int dtype, void* stream) {
    CHECK_DT();
    VG_CHECK_ARG(x && dlogit && dw && B > 0 && K > 0 && C > 0 && HW > 0 && C * HW == K, VG_EINVAL);
    DISPATCH_DT(dot_wgrad_kernel, dim3((K + 63) / 64), dim3(1024), vg_stream(stream), x, dlogit, dw, B, K, C, HW,
                accumulate);
    return VG_LAUNCH_RC();
}

extern "C" int vg_bce_forward_backward(const float* p, float target, int B, float gscale, float* loss, int accumulate,
                                       float* dp, void* stream) {
    VG_CHECK_ARG(p && loss && B > 0, VG_EINVAL);
    hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(256), 0, vg_stream(stream), p, target, B, gscale, loss, accumulate, dp);
    return VG_LAUNCH_RC();
}

extern "C" int vg_bce_pair_forward_backward(const float* p, float target0, float target1, int B, float gscale, float* loss,
                                            int accumulate, float* dp, void* stream) {
    VG_CHECK_ARG(p && loss && B > 0, VG_EINVAL);
    hipLaunchKernelGGL(bce_pair_kernel, dim3(1), dim3(256), 0, vg_stream(stream), p, target0, target1, B, gscale, loss,
                       accumulate, dp);
    return VG_LAUNCH_RC();
}

extern "C" int vg_head_backward(const float* p, const void* x, const void* w, void* dx, float* dw, float* dlogit, int B,
                                int groups, float target0, float target1, float gscale, float* loss, int accumulate_loss,
                                int accumulate_dw, int K, int C, int HW, int dtype, void* stream) {
    CHECK_DT();
    VG_CHECK_ARG(p && x && w && loss && B > 0 && (groups == 1 || groups == 2) && K > 0 && C > 0 && HW > 0 && C * HW == K, VG_EINVAL);
    VG_CHECK_ARG(K % 4 == 0, VG_EALIGN);
    VG_CHECK_ARG(B * groups <= HB_MAXROWS, VG_ENOSUP);
    DISPATCH_DT(head_bwd_kernel, dim3((K + 63) / 64), dim3(1024), vg_stream(stream), p, x, w, dx, dw, dlogit, B, groups,
                target0, target1, gscale, loss, accumulate_loss, accumulate_dw, K, C, HW);
    return VG_LAUNCH_RC();
}

extern "C" int vg_mean_forward_backward(const float* p, float sign, int B, float gscale, float* loss, int accumulate,
                                        float* dp, void* stream) {
    VG_CHECK_ARG(p && loss && B > 0, VG_EINVAL);
    hipLaunchKernelGGL(mean_loss_kernel, dim3(1), dim3(256), 0, vg_stream(stream), p, sign, B, gscale, loss, accumulate, dp);
    return VG_LAUNCH_RC();
}

extern "C" int vg_clamp(float* p, int64_t n, float lo, float hi, void* stream) {
    VG_CHECK_ARG(p && n > 0 && lo <= hi, VG_EINVAL);
    hipLaunchKernelGGL(clamp_kernel, dim3(blocks_for(n)), dim3(256), 0, vg_stream(stream), p, n, lo, hi);
    return VG_LAUNCH_RC();
}

extern "C" int vg_mse_forward_backward(const float* a, const float* b, int64_t n, float gscale, float* loss, float* d_a,
                                       float* ws, int ws_capacity, void* stream) {
    VG_CHECK_ARG(a && b && loss && ws && n > 0 && ws_capacity >= 1, VG_EINVAL);
    VG_CHECK_ARG(vg_aligned16(a) && vg_aligned16(b) && (d_a == nullptr || vg_aligned16(d_a)), VG_EALIGN);
    int blocks = blocks_for(n / 4, 1024);
    if (blocks > ws_capacity) blocks = ws_capacity;
    const float gcoef = (float)((double)gscale * 2.0 / (double)n);
    hipLaunchKernelGGL(mse_partial_kernel, dim3(blocks), dim3(256), 0, vg_stream(stream), a, b, n, gcoef, d_a, ws);
    int rc = VG_LAUNCH_RC();
    if (rc) return rc;
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(64), 0, vg_stream(stream), ws, blocks, (double)n, loss);
    return VG_LAUNCH_RC();
}

extern "C" int vg_ssim(const float* a, const float* b, int B, int C, int H, int W, float* out, float* ws,
                       int ws_capacity, void* stream) {
    VG_CHECK_ARG(a && b && out && ws && B > 0 && C > 0 && H > 10 && W > 10 && ws_capacity >= 1, VG_EINVAL);
    const int64_t total = (int64_t)B * C * (H - 10) * (W - 10);
    int blocks = blocks_for(total, 1024);
    if (blocks > ws_capacity) blocks = ws_capacity;
    hipLaunchKernelGGL(ssim_kernel, dim3(blocks), dim3(256), 0, vg_stream(stream), a, b, B * C, H, W, ws);
    int rc = VG_LAUNCH_RC();
    if (rc) return rc;
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(64), 0, vg_stream(stream), ws, blocks, (double)total, out);
    return VG_LAUNCH_RC();
}

extern "C" int vg_axpy(const float* a, const float* b, float alpha, float* out, int64_t n, void* stream) {
    VG_CHECK_ARG(a && b && out && n > 0, VG_EINVAL);
    hipLaunchKernelGGL(axpy_kernel, dim3(blocks_for(n)), dim3(256), 0, vg_stream(stream), a, b, alpha, out, n);
    return VG_LAUNCH_RC();
}

// ---- runtime switches (common.hpp): read once at load, re-read by vg_reload_switches() ----
#include <cstdlib>
namespace {
VgSwitches g_sw;
int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
void read_switches() {
    g_sw.gg_dma = env_int("VG_GG_DMA", 1);
    g_sw.tile_min_wgs = env_int("VG_TILE_MIN_WGS", 512);
    g_sw.gg_patch = env_int("VG_GG_PATCH", 1);
    g_sw.gg_patch64 = env_int("VG_GG_PATCH64", 1);
    g_sw.gg_patch32 = env_int("VG_GG_PATCH32", 1);
    g_sw.gg_phase4 = env_int("VG_GG_PHASE4", 1);
    g_sw.gg_phase4_min = env_int("VG_GG_PHASE4_MIN", 512);
    g_sw.gg_patch_nr3 = env_int("VG_GG_PATCH_NR3", 1);
    g_sw.patch256_min = env_int("VG_PATCH256_MIN", 256);
    g_sw.patch256x64_min = env_int("VG_PATCH256X64_MIN", 512);
    g_sw.splitk_max_tiles = env_int("VG_SPLITK_MAX_TILES", 128);
    g_sw.splitk_wgs = env_int("VG_SPLITK_WGS", 1024);
    g_sw.splitk_bigk = env_int("VG_SPLITK_BIGK", 1);
    g_sw.splitk_general = env_int("VG_SPLITK_GENERAL", 0);
    g_sw.gg_nmajor = env_int("VG_GG_NMAJOR", 1);
    g_sw.edge = env_int("VG_EDGE", 1);
    g_sw.wg_reduce_t = env_int("VG_WG_REDUCE_T", 1);
    g_sw.wg_target = env_int("VG_WG_TARGET", 512);
    g_sw.wg_spec = env_int("VG_WG_SPEC", 3);
    g_sw.wg_dma = env_int("VG_WG_DMA", 1);
    g_sw.wg_xcd = env_int("VG_WG_XCD", 1);
    g_sw.bn_fused_fwd = env_int("VG_BN_FUSED_FWD", 1);
    g_sw.bn_onepass = env_int("VG_BN_ONEPASS", 0);
}
struct SwitchInit { SwitchInit() { read_switches(); } } g_switch_init;
}  // namespace
const VgSwitches& vg_sw() { return g_sw; }
extern "C" int vg_reload_switches(void) {
    read_switches();
    return 0;
}

VgTiming& vg_timing() {
    static VgTiming t;
    return t;
}

std::atomic<uint64_t>& vg_launch_counter() {
    static std::atomic<uint64_t> n{0};
    return n;
}

extern "C" uint64_t vg_launch_count(void) { return vg_launch_counter().load(std::memory_order_relaxed); }

extern "C" int vg_timing_enable(int on) {
    VgTiming& t = vg_timing();
    std::lock_guard<std::mutex> g(t.mu);
    t.on = on != 0;
    return 0;
}

// Synchronises the recorded events of `family`, returns the summed kernel time and the launch count, recycles them.
extern "C" int vg_timing_collect(int family, double* total_ms, int* launches) {
    VG_CHECK_ARG(family >= 0 && family < 5 && total_ms && launches, VG_EINVAL);
    VgTiming& t = vg_timing();
    std::lock_guard<std::mutex> g(t.mu);
    double sum = 0.0;
    for (auto& pr : t.rec[family]) {
        (void)hipEventSynchronize(pr.second);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) sum += ms;
        t.pool.push_back(pr.first);
        t.pool.push_back(pr.second);
    }
    *total_ms = sum;
    *launches = (int)t.rec[family].size();
    t.rec[family].clear();
    return 0;
}

extern "C" int vg_abi_version(void) { return VG_ABI_VERSION; }
extern "C" const char* vg_build_info(void) { return "vaegan_hip gfx950: gather-GEMM f32(16x16x4)/bf16(16x16x32) MFMA"; }

// Operand packing (f32 parameters in the reference layouts -> K-major GEMM operands) and the
// flat Adam step (torch.optim.Adam as built at vaegan_code.py:42-44).
#include "common.hpp"

namespace {

template <int DT>
__global__ void pack_kernel(const vg_pack_desc d) {
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

extern "C" int vg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                            double beta2, double eps, float grad_scale, float* state, void* stream) {
    VG_CHECK_ARG(p && g && m && v && state && n > 0, VG_EINVAL);
    VG_CHECK_ARG(vg_aligned16(p) && vg_aligned16(g) && vg_aligned16(m) && vg_aligned16(v), VG_EALIGN);
    hipStream_t s = vg_stream(stream);
    hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(64), 0, s, state, lr, beta1, beta2);
    int rc = VG_LAUNCH_RC();
    if (rc) return rc;
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

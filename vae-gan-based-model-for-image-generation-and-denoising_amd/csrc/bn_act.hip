// BatchNorm2d (train/eval) + ReLU/LeakyReLU forward and backward on NHWC tensors.
// Replaces the ATen batch_norm / leaky_relu / relu kernels reached through nn.BatchNorm2d,
// nn.LeakyReLU and nn.ReLU at main_vae.py:24-25 and gan_code.py:22-82 (+ their backward).
//
// All tensors are [rows = B*H*W][C] with C contiguous, so a per-channel reduction is a column
// sum: a workgroup owns a contiguous block of rows, each thread a fixed 4-channel column
// (16-byte loads, fully coalesced), partial sums go wave -> LDS -> one slab row per
// workgroup, and a one-thread-per-channel finalize kernel adds the slab rows in fixed order
// in double precision.  HBM-bound: ideal traffic is one read (+ one write) of the tensor.
#include "common.hpp"

namespace {

struct RedPlan { int nparts, rows_per_part, ncolblk; };

inline RedPlan plan_reduce(int64_t rows, int C) {
    const int cols = C / 4;
    const int ncol = cols < 256 ? cols : 256;
    const int rpp = 256 / ncol;
    const int64_t passes = (rows + rpp - 1) / rpp;
    int64_t np = passes / 8;
    if (np < 1) np = 1;
    if (np > 1024) np = 1024;
    int64_t rows_per_part = ((passes + np - 1) / np) * rpp;
    RedPlan p;
    p.nparts = (int)((rows + rows_per_part - 1) / rows_per_part);
    p.rows_per_part = (int)rows_per_part;
    p.ncolblk = (cols + 255) / 256;
    return p;
}

// MODE 0: (sum x, sum x*x).  MODE 1: (sum dz, sum dz*xhat) with dz = dy*act'(scale*x+shift).
template <int DT, int MODE>
__global__ __launch_bounds__(256) void col_reduce_kernel(const void* __restrict__ x, const void* __restrict__ dy,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, int64_t rows, int C,
                                                         int act, float slope, float* __restrict__ partial,
                                                         int rows_per_part, int64_t rows_per_group,
                                                         int parts_per_group, int64_t gstride) {
    __shared__ float4 red[2][256];
    const int cols = C >> 2;
    const int cb = blockIdx.y * 256;
    const int ncol = min(256, cols - cb);
    const int rpp = 256 / ncol;
    const int tid = threadIdx.x;
    const int tr = tid / ncol, tc = tid - tr * ncol;
    const bool active = tr < rpp;
    const int c4 = (cb + tc) * 4;
    // groups: independent row ranges with their own coefficient sets (two Discriminator passes in one launch)
    const int grp = blockIdx.x / parts_per_group;
    const int64_t gbase = (int64_t)grp * rows_per_group;
    const int64_t r0 = gbase + (int64_t)(blockIdx.x - grp * parts_per_group) * rows_per_part;
    const int64_t r1 = min(min(rows, gbase + rows_per_group), r0 + (int64_t)rows_per_part);
    const int64_t go = (int64_t)grp * gstride;
    float4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        float4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f}, mu = sh, is = sc;
        if (MODE == 1) {
            if (scale) { sc = *reinterpret_cast<const float4*>(scale + go + c4); sh = *reinterpret_cast<const float4*>(shift + go + c4); }
            mu = *reinterpret_cast<const float4*>(mean + go + c4);
            is = *reinterpret_cast<const float4*>(invstd + go + c4);
        }
        auto accum = [&](const float4& v, const float4& g) {
            if (MODE == 0) {
                s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
                s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
            } else {
                const float dz0 = act_bwd(sc.x * v.x + sh.x, g.x, act, slope);
                const float dz1 = act_bwd(sc.y * v.y + sh.y, g.y, act, slope);
                const float dz2 = act_bwd(sc.z * v.z + sh.z, g.z, act, slope);
                const float dz3 = act_bwd(sc.w * v.w + sh.w, g.w, act, slope);
                s1.x += dz0; s1.y += dz1; s1.z += dz2; s1.w += dz3;
                s2.x += dz0 * ((v.x - mu.x) * is.x); s2.y += dz1 * ((v.y - mu.y) * is.y);
                s2.z += dz2 * ((v.z - mu.z) * is.z); s2.w += dz3 * ((v.w - mu.w) * is.w);
            }
        };
        // four rows (eight loads) in flight per thread; summation order unchanged
        int64_t r = r0 + tr;
        for (; r + 3 * rpp < r1; r += 4 * rpp) {
            float4 v[4], g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = load4<DT>(x, (r + (int64_t)u * rpp) * C + c4);
                g[u] = (MODE == 1) ? load4<DT>(dy, (r + (int64_t)u * rpp) * C + c4) : float4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) accum(v[u], g[u]);
        }
        for (; r < r1; r += rpp) {
            const float4 v = load4<DT>(x, r * C + c4);
            const float4 g = (MODE == 1) ? load4<DT>(dy, r * C + c4) : float4{0.f, 0.f, 0.f, 0.f};
            accum(v, g);
        }
    }
    red[0][tid] = s1;
    red[1][tid] = s2;
    __syncthreads();
    if (tid < ncol) {
        float4 a = red[0][tid], b = red[1][tid];
        for (int k = 1; k < rpp; ++k) {
            const float4 u = red[0][k * ncol + tid], w = red[1][k * ncol + tid];
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            b.x += w.x; b.y += w.y; b.z += w.z; b.w += w.w;
        }
        const int64_t part = blockIdx.x;
        *reinterpret_cast<float4*>(partial + (part * 2 + 0) * C + c4) = a;
        *reinterpret_cast<float4*>(partial + (part * 2 + 1) * C + c4) = b;
    }
}

// Sum the [nparts][2][C] partial slabs for 8 channels per workgroup: 128 "planes" of threads stride over the
// slab rows (a one-thread-per-channel loop over up to 4096 rows is pure latency: it cost 23 % of the step; with
// 32 planes these ~48 launches per iteration were still 5 us each, 6 % of it), double accumulation, then a
// fixed-order combine (xor-shuffles over the 8 planes of a wave, LDS over the 16 waves) -> bitwise reproducible.
// Returns the sums to threads tid < 8.
constexpr int FIN_CH = 8, FIN_PL = 128;
__device__ __forceinline__ bool slab_sums(const float* __restrict__ slabs, int nparts, int C, double& s1, double& s2,
                                          int& c_out) {
    constexpr int NW = FIN_CH * FIN_PL / 64;
    __shared__ double red[NW][FIN_CH][2];
    const int cl = threadIdx.x & (FIN_CH - 1);
    const int pl = threadIdx.x / FIN_CH;
    const int c = blockIdx.x * FIN_CH + cl;
    double a = 0.0, b = 0.0;
    if (c < C) {
        // 4 slab rows (8 loads) in flight per thread: the loop is a chain of L2/HBM latencies otherwise
        int p = pl;
        for (; p + 3 * FIN_PL < nparts; p += 4 * FIN_PL) {
            float x[4], y[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                x[k] = slabs[((int64_t)(p + k * FIN_PL) * 2 + 0) * C + c];
                y[k] = slabs[((int64_t)(p + k * FIN_PL) * 2 + 1) * C + c];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { a += (double)x[k]; b += (double)y[k]; }
        }
        for (; p < nparts; p += FIN_PL) {
            a += (double)slabs[((int64_t)p * 2 + 0) * C + c];
            b += (double)slabs[((int64_t)p * 2 + 1) * C + c];
        }
    }
    // a wave holds 8 planes x 8 channels (lane = 8*plane + channel): butterfly over the plane bits
#pragma unroll
    for (int o = FIN_CH; o < 64; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < FIN_CH) { red[wave][lane][0] = a; red[wave][lane][1] = b; }
    __syncthreads();
    if (pl != 0 || c >= C) return false;
    a = red[0][cl][0]; b = red[0][cl][1];
#pragma unroll
    for (int k = 1; k < NW; ++k) { a += red[k][cl][0]; b += red[k][cl][1]; }
    s1 = a; s2 = b; c_out = c;
    return true;
}

__global__ __launch_bounds__(FIN_CH * FIN_PL) void bn_finalize_kernel(
    const float* __restrict__ stats, int nparts, int C, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
    float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift) {
    double s1, s2;
    int c;
    if (!slab_sums(stats, nparts, C, s1, s2, c)) return;
    const double mu = s1 / count;
    double var = s2 / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float muf = (float)mu;
    mean[c] = muf;
    invstd[c] = is;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * is;
    scale[c] = sc;
    shift[c] = b - muf * sc;
    if (rmean) {
        const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * muf;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    }
}

// ---- grouped finalize: `groups` independent row blocks (a Discriminator iteration's real and fake batches) in
// ONE launch; a channel's groups are processed in order by the same thread, so the running statistics are updated
// real-then-fake exactly as two separate forward calls would (and dgamma/dbeta accumulate in that order).
__global__ __launch_bounds__(FIN_CH * FIN_PL) void bn_finalize_grouped_kernel(
    const float* __restrict__ stats, int nparts, int groups, int C, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
    float* __restrict__ coeffs) {
    for (int g = 0; g < groups; ++g) {
        double s1, s2;
        int c;
        const bool lead = slab_sums(stats + (int64_t)g * nparts * 2 * C, nparts, C, s1, s2, c);
        if (lead) {
            float* co = coeffs + (int64_t)g * 4 * C;
            const double mu = s1 / count;
            double var = s2 / count - mu * mu;
            if (var < 0.0) var = 0.0;
            const float is = (float)(1.0 / sqrt(var + (double)eps));
            const float muf = (float)mu;
            co[c] = muf;
            co[C + c] = is;
            const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
            const float sc = gm * is;
            co[2 * C + c] = sc;
            co[3 * C + c] = bt - muf * sc;
            if (rmean) {
                const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
                rmean[c] = (1.f - momentum) * rmean[c] + momentum * muf;
                rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
            }
        }
        __syncthreads();                                    // slab_sums' LDS scratch is reused by the next group
    }
}

__global__ __launch_bounds__(FIN_CH * FIN_PL) void bn_bwd_finalize_grouped_kernel(
    const float* __restrict__ partial, int nparts, int groups, int C, double count, const float* __restrict__ gamma,
    const float* __restrict__ coeffs, float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
    float* __restrict__ coef) {
    for (int g = 0; g < groups; ++g) {
        double s1, s2;
        int c;
        const bool lead = slab_sums(partial + (int64_t)g * nparts * 2 * C, nparts, C, s1, s2, c);
        if (lead) {
            const float fs1 = (float)s1, fs2 = (float)s2;
            const bool acc = accumulate || g > 0;
            if (dgamma) dgamma[c] = acc ? dgamma[c] + fs2 : fs2;
            if (dbeta) dbeta[c] = acc ? dbeta[c] + fs1 : fs1;
            const float a = (gamma ? gamma[c] : 1.f) * coeffs[(int64_t)g * 4 * C + C + c];      // gamma * invstd
            float* cf = coef + (int64_t)g * 3 * C;
            cf[c] = a;
            cf[C + c] = (float)((double)a * s2 / count);
            cf[2 * C + c] = (float)((double)a * s1 / count);
        }
        __syncthreads();
    }
}

// ---- synchronised BatchNorm (one process per GPU, statistics over the GLOBAL batch) --------------------------
// The host all-reduces the f64 [2][C] sums between these kernels (vaegan_amd ddp.py); arithmetic after the sums
// is the same as bn_finalize_kernel / bn_bwd_finalize_kernel.
__global__ __launch_bounds__(FIN_CH * FIN_PL) void slab_sums_kernel(const float* __restrict__ slabs, int nparts, int C,
                                                                    double* __restrict__ sums) {
    double s1, s2;
    int c;
    if (!slab_sums(slabs, nparts, C, s1, s2, c)) return;
    sums[c] = s1;
    sums[C + c] = s2;
}

__global__ void bn_finalize_sums_kernel(const double* __restrict__ sums, int C, double count,
                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                        float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
                                        float* __restrict__ mean, float* __restrict__ invstd,
                                        float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double mu = sums[c] / count;
    double var = sums[C + c] / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float muf = (float)mu;
    mean[c] = muf;
    invstd[c] = is;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * is;
    scale[c] = sc;
    shift[c] = b - muf * sc;
    if (rmean) {
        const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * muf;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    }
}

// dgamma/dbeta are this rank's LOCAL sums (the gradient all-reduce averages them like every other parameter
// gradient); the dx coefficients use the GLOBAL sums and the global element count.
__global__ void bn_bwd_finalize_sums_kernel(const double* __restrict__ gsums, const double* __restrict__ lsums, int C,
                                            double count, const float* __restrict__ gamma,
                                            const float* __restrict__ invstd, float* __restrict__ dgamma,
                                            float* __restrict__ dbeta, int accumulate, float* __restrict__ coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float l1 = (float)lsums[c], l2 = (float)lsums[C + c];
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + l2 : l2;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + l1 : l1;
    const float a = (gamma ? gamma[c] : 1.f) * invstd[c];
    coef[c] = a;
    coef[C + c] = (float)((double)a * gsums[C + c] / count);
    coef[2 * C + c] = (float)((double)a * gsums[c] / count);
}

__global__ void bn_eval_kernel(const float* gamma, const float* beta, const float* rmean, const float* rvar,
                               float eps, int C, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float is = 1.f / sqrtf(rvar[c] + eps);
    const float sc = (gamma ? gamma[c] : 1.f) * is;
    scale[c] = sc;
    shift[c] = (beta ? beta[c] : 0.f) - rmean[c] * sc;
}

template <int DT>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int64_t nvec, int cols,
                                                         int act, float slope, int64_t nvec_per_group,
                                                         int64_t gstride, uint32_t* __restrict__ y8) {
    auto one = [&](int64_t i, float4 v) {
        const int c4 = (int)(i % cols) * 4;
        if (scale) {
            const int64_t go = (i / nvec_per_group) * gstride;
            const float4 sc = *reinterpret_cast<const float4*>(scale + go + c4);
            const float4 sh = *reinterpret_cast<const float4*>(shift + go + c4);
            v.x = sc.x * v.x + sh.x; v.y = sc.y * v.y + sh.y; v.z = sc.z * v.z + sh.z; v.w = sc.w * v.w + sh.w;
        }
        v.x = act_fwd(v.x, act, slope); v.y = act_fwd(v.y, act, slope);
        v.z = act_fwd(v.z, act, slope); v.w = act_fwd(v.w, act, slope);
        store4<DT>(y, i * 4, v);
        if (y8) {                   // e4m3 twin of the activated tensor for an fp8 forward GEMM (same NHWC layout)
            int w = __builtin_amdgcn_cvt_pk_fp8_f32(v.x, v.y, 0, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v.z, v.w, w, true);
            y8[i] = (uint32_t)w;
        }
    };
    // four vectors in flight per thread (the grid is capped: big tensors give each thread several iterations)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = load4<DT>(x, (i + u * stride) * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) one(i + u * stride, v[u]);
    }
    for (; i < nvec; i += stride) one(i, load4<DT>(x, i * 4));
}

__global__ __launch_bounds__(FIN_CH * FIN_PL) void bn_bwd_finalize_kernel(
    const float* __restrict__ partial, int nparts, int C, double count, const float* __restrict__ gamma,
    const float* __restrict__ invstd, float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
    float* __restrict__ coef) {
    double s1, s2;
    int c;
    if (!slab_sums(partial, nparts, C, s1, s2, c)) return;
    const float fs1 = (float)s1, fs2 = (float)s2;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + fs2 : fs2;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + fs1 : fs1;
    const float a = (gamma ? gamma[c] : 1.f) * invstd[c];
    coef[c] = a;
    coef[C + c] = (float)((double)a * s2 / count);
    coef[2 * C + c] = (float)((double)a * s1 / count);
}

template <int DT>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const void* __restrict__ x, const void* __restrict__ dy,
                                                               void* __restrict__ dx,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd,
                                                               const float* __restrict__ coef, int64_t nvec, int cols,
                                                               int C, int act, float slope, int64_t nvec_per_group,
                                                               int64_t gstride, int64_t cstride) {
    auto one = [&](int64_t i, const float4& v, const float4& g) {
        const int c4 = (int)(i % cols) * 4;
        const int64_t grp = i / nvec_per_group;
        const int64_t go = grp * gstride;
        const float* cf = coef + grp * cstride;
        const float4 sc = *reinterpret_cast<const float4*>(scale + go + c4);
        const float4 sh = *reinterpret_cast<const float4*>(shift + go + c4);
        const float4 mu = *reinterpret_cast<const float4*>(mean + go + c4);
        const float4 is = *reinterpret_cast<const float4*>(invstd + go + c4);
        const float4 ca = *reinterpret_cast<const float4*>(cf + c4);
        const float4 cbv = *reinterpret_cast<const float4*>(cf + C + c4);
        const float4 cc = *reinterpret_cast<const float4*>(cf + 2 * C + c4);
        float4 o;
        o.x = ca.x * act_bwd(sc.x * v.x + sh.x, g.x, act, slope) - cbv.x * ((v.x - mu.x) * is.x) - cc.x;
        o.y = ca.y * act_bwd(sc.y * v.y + sh.y, g.y, act, slope) - cbv.y * ((v.y - mu.y) * is.y) - cc.y;
        o.z = ca.z * act_bwd(sc.z * v.z + sh.z, g.z, act, slope) - cbv.z * ((v.z - mu.z) * is.z) - cc.z;
        o.w = ca.w * act_bwd(sc.w * v.w + sh.w, g.w, act, slope) - cbv.w * ((v.w - mu.w) * is.w) - cc.w;
        store4<DT>(dx, i * 4, o);
    };
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < nvec; i += 4 * stride) {          // eight loads in flight per thread
        float4 v[4], g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { v[u] = load4<DT>(x, (i + u * stride) * 4); g[u] = load4<DT>(dy, (i + u * stride) * 4); }
#pragma unroll
        for (int u = 0; u < 4; ++u) one(i + u * stride, v[u], g[u]);
    }
    for (; i < nvec; i += stride) one(i, load4<DT>(x, i * 4), load4<DT>(dy, i * 4));
}

// ---- train-mode BatchNorm finalize + normalise + activation in ONE launch (bf16, small statistics slabs) ----------
// bn_finalize_grouped_kernel is a 5 us launch whose only output is 4 C floats; for layers whose slab is small
// (<= 200 rows per group, tensor <= 9 MB) every workgroup of the elementwise pass can afford to redo it: a workgroup of 1024 threads
// owns 64 channels of a block of rows, sums the slab rows of its group for those channels (16 part lanes x 64
// channels, double, fixed order -- every workgroup of a channel slice computes bit-identical coefficients), keeps
// scale / shift in LDS and applies them.  The workgroups of row block 0 also publish the coefficients
// ([groups][4][C], what the backward pass reads) and update the running statistics, group after group.
constexpr int FF_CH = 64, FF_PL = 16, FF_TH = FF_CH * FF_PL, FF_MAXPARTS = 200;
constexpr int64_t FF_MAXBYTES = 9ll << 20;       // tensor size up to which re-deriving the coefficients per workgroup pays

__device__ __forceinline__ void ff_slice_stats(const float* __restrict__ slabs, int nparts, int C, int c0, double count,
                                               double (*red)[FF_CH][2], double& mu_out, double& var_out) {
    const int ch = threadIdx.x & (FF_CH - 1), pl = threadIdx.x / FF_CH;
    double a = 0.0, b = 0.0;
    for (int p = pl; p < nparts; p += FF_PL) {
        a += (double)slabs[((int64_t)p * 2 + 0) * C + c0 + ch];
        b += (double)slabs[((int64_t)p * 2 + 1) * C + c0 + ch];
    }
    red[pl][ch][0] = a;
    red[pl][ch][1] = b;
    __syncthreads();
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < FF_PL; ++k) { s1 += red[k][ch][0]; s2 += red[k][ch][1]; }
    const double mu = s1 / count;
    double var = s2 / count - mu * mu;
    if (var < 0.0) var = 0.0;
    mu_out = mu;
    var_out = var;
    __syncthreads();                                           // red[] is reused by the next group (publisher blocks)
}

__global__ __launch_bounds__(FF_TH) void bn_fin_act_fwd_kernel(
    const uint16_t* __restrict__ x, uint16_t* __restrict__ y, const float* __restrict__ stats, int nparts, int groups,
    int C, double count, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ rmean,
    float* __restrict__ rvar, float momentum, float eps, float* __restrict__ coeffs, int64_t rows_per_group,
    int rows_per_block, int act, float slope) {
    __shared__ double red[FF_PL][FF_CH][2];
    __shared__ float s_sc[FF_CH], s_sh[FF_CH];
    const int tid = threadIdx.x, ch = tid & (FF_CH - 1);
    const int c0 = blockIdx.y * FF_CH, grp = blockIdx.z;
    const bool lead = tid < FF_CH && c0 + ch < C;              // one thread per channel of the slice
    const float gm = (lead && gamma) ? gamma[c0 + ch] : 1.f, bt = (lead && beta) ? beta[c0 + ch] : 0.f;
    if (blockIdx.x == 0 && grp == 0) {
        // publisher of this channel slice: every group's coefficients, running statistics in group order
        float rm = (lead && rmean) ? rmean[c0 + ch] : 0.f, rv = (lead && rmean) ? rvar[c0 + ch] : 0.f;
        for (int g = 0; g < groups; ++g) {
            double mu, var;
            ff_slice_stats(stats + (int64_t)g * nparts * 2 * C, nparts, C, c0, count, red, mu, var);
            if (lead) {
                const float is = (float)(1.0 / sqrt(var + (double)eps));
                const float muf = (float)mu, sc = gm * is, sh = bt - muf * sc;
                float* co = coeffs + (int64_t)g * 4 * C + c0 + ch;
                co[0] = muf; co[C] = is; co[2 * C] = sc; co[3 * C] = sh;
                if (g == 0) { s_sc[ch] = sc; s_sh[ch] = sh; }
                if (rmean) {
                    const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
                    rm = (1.f - momentum) * rm + momentum * muf;
                    rv = (1.f - momentum) * rv + momentum * (float)unbiased;
                }
            }
        }
        if (lead && rmean) { rmean[c0 + ch] = rm; rvar[c0 + ch] = rv; }
    } else {
        double mu, var;
        ff_slice_stats(stats + (int64_t)grp * nparts * 2 * C, nparts, C, c0, count, red, mu, var);
        if (lead) {
            const float is = (float)(1.0 / sqrt(var + (double)eps));
            const float muf = (float)mu, sc = gm * is;
            s_sc[ch] = sc;
            s_sh[ch] = bt - muf * sc;
        }
    }
    __syncthreads();
    // ---- apply: a thread owns 8 channels (16 bytes) of a row; 8 threads per row, 128 rows per pass ----
    const int u8 = (tid & 7) * 8, rl = tid >> 3;
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc[k] = s_sc[u8 + k]; sh[k] = s_sh[u8 + k]; }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows_per_group ? r0 + rows_per_block : rows_per_group;
    const int64_t base = (int64_t)grp * rows_per_group;
    if (c0 + u8 < C) {
        for (int64_t r = r0 + rl; r < r1; r += FF_TH / 8) {
            const int64_t off = (base + r) * C + c0 + u8;
            const u32x4 v = *reinterpret_cast<const u32x4*>(x + off);
            u32x4 o;
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                const float lo = act_fwd(sc[2 * k2] * __uint_as_float(v[k2] << 16) + sh[2 * k2], act, slope);
                const float hi = act_fwd(sc[2 * k2 + 1] * __uint_as_float(v[k2] & 0xffff0000u) + sh[2 * k2 + 1], act, slope);
                o[k2] = (uint32_t)ElemT<VG_BF16>::from_f32(lo) | ((uint32_t)ElemT<VG_BF16>::from_f32(hi) << 16);
            }
            *reinterpret_cast<u32x4*>(y + off) = o;
        }
    }
}

inline bool bn_fin_fwd_ok(int nparts_per_group, int groups, int C, int64_t rows, int dtype) {
    const int mode = vg_sw().bn_fused_fwd;
    // measured per shape (tools/bn_fwd_bench.py, S=64 B=128): wins 1-3 us per layer up to 8.4 MB / 196 slab rows
    // (G0 7.2 -> 4.5, D3 6.2 -> 4.3, E3 5.5 -> 3.8 us), loses beyond (G2 16.8 MB 13.4 -> 15.0, D1 256 slab rows 9.1 -> 11.4)
    return mode != 0 && dtype == VG_BF16 && C % FF_CH == 0 && groups >= 1 && rows % groups == 0 &&
           nparts_per_group > 0 && nparts_per_group <= FF_MAXPARTS && rows * C * 2 <= FF_MAXBYTES;
}

// ---- backward twin: bn_bwd_finalize_grouped + bn_act_bwd_apply in one launch (same eligibility, same structure) ----
// Every workgroup sums the column-reduce partials (sum dz | sum dz * xhat) of its group for its 64 channels, forms the
// three coefficients of bn_bwd_finalize_grouped_kernel and applies dx = a * dz - b * xhat - c; the workgroups of row
// block 0 accumulate dgamma / dbeta, group after group.
__global__ __launch_bounds__(FF_TH) void bn_bwd_fin_apply_kernel(
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, uint16_t* __restrict__ dx,
    const float* __restrict__ partial, int nparts, int groups, int C, double count, const float* __restrict__ gamma,
    const float* __restrict__ coeffs, float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
    int64_t rows_per_group, int rows_per_block, int act, float slope) {
    __shared__ double red[FF_PL][FF_CH][2];
    __shared__ float s_co[7][FF_CH];                            // mean | invstd | scale | shift | a | b | c
    const int tid = threadIdx.x, ch = tid & (FF_CH - 1), pl = tid / FF_CH;
    const int c0 = blockIdx.y * FF_CH, grp = blockIdx.z;
    const bool lead = tid < FF_CH && c0 + ch < C;
    const float gm = (lead && gamma) ? gamma[c0 + ch] : 1.f;
    auto slice_sums = [&](int g, double& s1, double& s2) {
        const float* pp = partial + (int64_t)g * nparts * 2 * C;
        double a = 0.0, b = 0.0;
        for (int p = pl; p < nparts; p += FF_PL) {
            a += (double)pp[((int64_t)p * 2 + 0) * C + c0 + ch];
            b += (double)pp[((int64_t)p * 2 + 1) * C + c0 + ch];
        }
        red[pl][ch][0] = a;
        red[pl][ch][1] = b;
        __syncthreads();
        s1 = 0.0; s2 = 0.0;
#pragma unroll
        for (int k = 0; k < FF_PL; ++k) { s1 += red[k][ch][0]; s2 += red[k][ch][1]; }
        __syncthreads();
    };
    auto publish_coef = [&](int g, double s1, double s2) {      // lead threads: this group's coefficients into LDS
        const float* co = coeffs + (int64_t)g * 4 * C + c0 + ch;
        const float is = co[C];
        const float a = gm * is;
        s_co[0][ch] = co[0]; s_co[1][ch] = is; s_co[2][ch] = co[2 * C]; s_co[3][ch] = co[3 * C];
        s_co[4][ch] = a;
        s_co[5][ch] = (float)((double)a * s2 / count);
        s_co[6][ch] = (float)((double)a * s1 / count);
    };
    if (blockIdx.x == 0 && grp == 0) {
        float dg = (lead && dgamma && accumulate) ? dgamma[c0 + ch] : 0.f;
        float db = (lead && dbeta && accumulate) ? dbeta[c0 + ch] : 0.f;
        for (int g = 0; g < groups; ++g) {                      // dgamma / dbeta accumulate in group order
            double s1, s2;
            slice_sums(g, s1, s2);
            if (lead) {
                if (g == 0) publish_coef(0, s1, s2);            // this workgroup applies group 0's rows
                const bool acc = accumulate || g > 0;
                dg = acc ? dg + (float)s2 : (float)s2;
                db = acc ? db + (float)s1 : (float)s1;
            }
        }
        if (lead) {
            if (dgamma) dgamma[c0 + ch] = dg;
            if (dbeta) dbeta[c0 + ch] = db;
        }
    } else {
        double s1, s2;
        slice_sums(grp, s1, s2);
        if (lead) publish_coef(grp, s1, s2);
    }
    __syncthreads();
    const int u8 = (tid & 7) * 8, rl = tid >> 3;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows_per_group ? r0 + rows_per_block : rows_per_group;
    const int64_t base = (int64_t)grp * rows_per_group;
    if (c0 + u8 < C) {
        for (int64_t r = r0 + rl; r < r1; r += FF_TH / 8) {
            const int64_t off = (base + r) * C + c0 + u8;
            const u32x4 v = *reinterpret_cast<const u32x4*>(x + off);
            const u32x4 gq = *reinterpret_cast<const u32x4*>(dy + off);
            u32x4 o;
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                float ov[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int k = u8 + 2 * k2 + h;
                    const float yv = __uint_as_float(h ? (v[k2] & 0xffff0000u) : (v[k2] << 16));
                    const float gr = __uint_as_float(h ? (gq[k2] & 0xffff0000u) : (gq[k2] << 16));
                    ov[h] = s_co[4][k] * act_bwd(s_co[2][k] * yv + s_co[3][k], gr, act, slope) -
                            s_co[5][k] * ((yv - s_co[0][k]) * s_co[1][k]) - s_co[6][k];
                }
                o[k2] = (uint32_t)ElemT<VG_BF16>::from_f32(ov[0]) | ((uint32_t)ElemT<VG_BF16>::from_f32(ov[1]) << 16);
            }
            *reinterpret_cast<u32x4*>(dx + off) = o;
        }
    }
}

template <int DT>
__global__ __launch_bounds__(256) void act_bwd_kernel(const void* __restrict__ x, const void* __restrict__ dy,
                                                      void* __restrict__ dx, int64_t nvec, int act, float slope) {
    auto one = [&](int64_t i, const float4& v, const float4& g) {
        float4 o;
        o.x = act_bwd(v.x, g.x, act, slope); o.y = act_bwd(v.y, g.y, act, slope);
        o.z = act_bwd(v.z, g.z, act, slope); o.w = act_bwd(v.w, g.w, act, slope);
        store4<DT>(dx, i * 4, o);
    };
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        float4 v[4], g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { v[u] = load4<DT>(x, (i + u * stride) * 4); g[u] = load4<DT>(dy, (i + u * stride) * 4); }
#pragma unroll
        for (int u = 0; u < 4; ++u) one(i + u * stride, v[u], g[u]);
    }
    for (; i < nvec; i += stride) one(i, load4<DT>(x, i * 4), load4<DT>(dy, i * 4));
}

__global__ __launch_bounds__(FIN_CH * FIN_PL) void bias_finalize_kernel(const float* __restrict__ partial, int nparts,
                                                                        int C, int NC, float* __restrict__ dbias,
                                                                        int accumulate) {
    double s1, s2;
    int c;
    if (!slab_sums(partial, nparts, C, s1, s2, c)) return;
    if (c >= NC) return;
    dbias[c] = accumulate ? dbias[c] + (float)s1 : (float)s1;
}

inline int ew_blocks(int64_t nvec) {
    int64_t b = (nvec + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

inline int check_rows_c(const void* x, int64_t rows, int C, int dtype) {
    VG_CHECK_ARG(x != nullptr && rows > 0 && C > 0, VG_EINVAL);
    VG_CHECK_ARG(dtype == VG_F32 || dtype == VG_BF16, VG_ENOSUP);
    VG_CHECK_ARG(C % 4 == 0, VG_EALIGN);
    VG_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 7u) == 0, VG_EALIGN);
    return 0;
}

// groups > 1: `rows` = groups * rows_per_group; the slabs of group g are parts [g*nparts_out, (g+1)*nparts_out)
template <int MODE>
int launch_reduce(const void* x, const void* dy, const float* scale, const float* shift, const float* mean,
                  const float* invstd, int64_t rows, int C, int act, float slope, float* partial, int capacity,
                  int* nparts_out, int dtype, hipStream_t s, int groups = 1, int64_t gstride = 0) {
    VG_CHECK_ARG(groups >= 1 && rows % groups == 0, VG_EINVAL);
    const int64_t rpg = rows / groups;
    RedPlan p = plan_reduce(rpg, C);
    if (nparts_out) *nparts_out = p.nparts;
    VG_CHECK_ARG(partial != nullptr && capacity >= p.nparts * groups, VG_EINVAL);
    dim3 grid(p.nparts * groups, p.ncolblk);
    if (dtype == VG_F32)
        vg_launch_timed(4, (col_reduce_kernel<VG_F32, MODE>), grid, dim3(256), 0, s, x, dy, scale, shift, mean, invstd,
                           rows, C, act, slope, partial, p.rows_per_part, rpg, p.nparts, gstride);
    else
        vg_launch_timed(4, (col_reduce_kernel<VG_BF16, MODE>), grid, dim3(256), 0, s, x, dy, scale, shift, mean, invstd,
                           rows, C, act, slope, partial, p.rows_per_part, rpg, p.nparts, gstride);
    return VG_LAUNCH_RC();
}

}  // namespace

extern "C" int vg_bn_finalize(const float* stats, int nparts, int C, int64_t count, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                              float* mean, float* invstd, float* scale, float* shift, void* stream) {
    VG_CHECK_ARG(stats && nparts > 0 && C > 0 && count > 0 && mean && invstd && scale && shift, VG_EINVAL);
    VG_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), VG_EINVAL);
    vg_launch_timed(4, bn_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_PL), 0, vg_stream(stream), stats, nparts, C,
                       (double)count, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale,
                       shift);
    return VG_LAUNCH_RC();
}

extern "C" int vg_bn_finalize_grouped(const float* stats, int nparts_per_group, int groups, int C,
                                      int64_t count_per_group, const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, float momentum, float eps,
                                      float* coeffs, void* stream) {
    VG_CHECK_ARG(stats && coeffs && nparts_per_group > 0 && groups > 0 && C > 0 && count_per_group > 0, VG_EINVAL);
    VG_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), VG_EINVAL);
    vg_launch_timed(4, bn_finalize_grouped_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_PL), 0,
                       vg_stream(stream), stats, nparts_per_group, groups, C, (double)count_per_group, gamma, beta,
                       running_mean, running_var, momentum, eps, coeffs);
    return VG_LAUNCH_RC();
}

extern "C" int vg_bn_backward_finalize_grouped(const float* partial, int nparts_per_group, int groups, int C,
                                               int64_t count_per_group, const float* gamma, const float* coeffs,
                                               float* dgamma, float* dbeta, int accumulate, float* coef,
                                               void* stream) {
    VG_CHECK_ARG(partial && coeffs && coef && nparts_per_group > 0 && groups > 0 && C > 0 && count_per_group > 0,
                 VG_EINVAL);
    vg_launch_timed(4, bn_bwd_finalize_grouped_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_PL), 0,
                       vg_stream(stream), partial, nparts_per_group, groups, C, (double)count_per_group, gamma, coeffs,
                       dgamma, dbeta, accumulate, coef);
    return VG_LAUNCH_RC();
}

extern "C" int vg_slab_sums(const float* slabs, int nparts, int C, double* sums, void* stream) {
    VG_CHECK_ARG(slabs && sums && nparts > 0 && C > 0, VG_EINVAL);
    vg_launch_timed(4, slab_sums_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_PL), 0, vg_stream(stream),
                       slabs, nparts, C, sums);
    return VG_LAUNCH_RC();
}

extern "C" int vg_bn_finalize_sums(const double* sums, int C, int64_t count, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, float momentum, float eps, float* mean,
                                   float* invstd, float* scale, float* shift, void* stream) {
    VG_CHECK_ARG(sums && C > 0 && count > 0 && mean && invstd && scale && shift, VG_EINVAL);
    VG_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), VG_EINVAL);
    vg_launch_timed(4, bn_finalize_sums_kernel, dim3((C + 255) / 256), dim3(256), 0, vg_stream(stream), sums, C,
                       (double)count, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale,
                       shift);
    return VG_LAUNCH_RC();
}

extern "C" int vg_bn_backward_finalize_sums(const double* global_sums, const double* local_sums, int C, int64_t count,
                                            const float* gamma, const float* invstd, float* dgamma, float* dbeta,
                                            int accumulate, float* coef, void* stream) {
    VG_CHECK_ARG(global_sums && local_sums && C > 0 && count > 0 && invstd && coef, VG_EINVAL);
    vg_launch_timed(4, bn_bwd_finalize_sums_kernel, dim3((C + 255) / 256), dim3(256), 0, vg_stream(stream),
                       global_sums, local_sums, C, (double)count, gamma, invstd, dgamma, dbeta, accumulate, coef);
    return VG_LAUNCH_RC();
}

extern "C" int vg_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                 const float* running_var, float eps, int C, float* scale, float* shift,
                                 void* stream) {
    VG_CHECK_ARG(running_mean && running_var && scale && shift && C > 0, VG_EINVAL);
    vg_launch_timed(4, bn_eval_kernel, dim3((C + 63) / 64), dim3(64), 0, vg_stream(stream), gamma, beta, running_mean,
                       running_var, eps, C, scale, shift);
    return VG_LAUNCH_RC();
}

extern "C" int vg_bn_finalize_act_forward_supported(int nparts_per_group, int groups, int C, int64_t rows, int dtype) {
    return bn_fin_fwd_ok(nparts_per_group, groups, C, rows, dtype) ? 1 : 0;
}

extern "C" int vg_bn_finalize_act_forward(const void* x, void* y, const float* stats, int nparts_per_group, int groups,
                                          int C, int64_t rows, const float* gamma, const float* beta,
                                          float* running_mean, float* running_var, float momentum, float eps,
                                          float* coeffs, int act, float slope, int dtype, void* stream) {
    VG_CHECK_ARG(x && y && stats && coeffs && rows > 0 && C > 0, VG_EINVAL);
    VG_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), VG_EINVAL);
    VG_CHECK_ARG(vg_aligned16(x) && vg_aligned16(y), VG_EALIGN);
    if (!bn_fin_fwd_ok(nparts_per_group, groups, C, rows, dtype)) return VG_ENOSUP;
    const int64_t rpg = rows / groups;
    // ~512 workgroups in all; whole passes of 128 rows per workgroup
    const int slices = C / FF_CH;
    int64_t want = 512 / (slices * groups);
    if (want < 1) want = 1;
    int64_t rpb = (rpg + want - 1) / want;
    rpb = (rpb + 127) / 128 * 128;
    const int rb = (int)((rpg + rpb - 1) / rpb);
    vg_launch_timed(4, bn_fin_act_fwd_kernel, dim3(rb, slices, groups), dim3(FF_TH), 0, vg_stream(stream),
                       reinterpret_cast<const uint16_t*>(x), reinterpret_cast<uint16_t*>(y), stats, nparts_per_group, groups,
                       C, (double)rpg, gamma, beta, running_mean, running_var, momentum, eps, coeffs, rpg, (int)rpb, act,
                       slope);
    return VG_LAUNCH_RC();
}

extern "C" int vg_bn_backward_finalize_apply(const void* x, const void* dy, void* dx, const float* partial,
                                             int nparts_per_group, int groups, int C, int64_t rows, const float* gamma,
                                             const float* coeffs, float* dgamma, float* dbeta, int accumulate, int act,
                                             float slope, int dtype, void* stream) {
    VG_CHECK_ARG(x && dy && dx && partial && coeffs && rows > 0 && C > 0, VG_EINVAL);
    VG_CHECK_ARG(vg_aligned16(x) && vg_aligned16(dy) && vg_aligned16(dx), VG_EALIGN);
    if (!bn_fin_fwd_ok(nparts_per_group, groups, C, rows, dtype)) return VG_ENOSUP;
    const int64_t rpg = rows / groups;
    const int slices = C / FF_CH;
    int64_t want = 512 / (slices * groups);
    if (want < 1) want = 1;
    int64_t rpb = (rpg + want - 1) / want;
    rpb = (rpb + 127) / 128 * 128;
    const int rb = (int)((rpg + rpb - 1) / rpb);
    vg_launch_timed(4, bn_bwd_fin_apply_kernel, dim3(rb, slices, groups), dim3(FF_TH), 0, vg_stream(stream),
                       reinterpret_cast<const uint16_t*>(x), reinterpret_cast<const uint16_t*>(dy),
                       reinterpret_cast<uint16_t*>(dx), partial, nparts_per_group, groups, C, (double)rpg, gamma, coeffs,
                       dgamma, dbeta, accumulate, rpg, (int)rpb, act, slope);
    return VG_LAUNCH_RC();
}

extern "C" int vg_bn_act_forward_fp8(const void* x, void* y, void* y8, const float* scale, const float* shift, int64_t rows,
                                     int C, int act, float slope, int groups, int64_t gstride, int dtype, void* stream);

extern "C" int vg_bn_act_forward(const void* x, void* y, const float* scale, const float* shift, int64_t rows, int C,
                                 int act, float slope, int groups, int64_t gstride, int dtype, void* stream) {
    return vg_bn_act_forward_fp8(x, y, nullptr, scale, shift, rows, C, act, slope, groups, gstride, dtype, stream);
}

extern "C" int vg_bn_act_forward_fp8(const void* x, void* y, void* y8, const float* scale, const float* shift, int64_t rows,
                                     int C, int act, float slope, int groups, int64_t gstride, int dtype, void* stream) {
    int rc = check_rows_c(x, rows, C, dtype);
    if (rc) return rc;
    VG_CHECK_ARG(y != nullptr && (scale == nullptr) == (shift == nullptr), VG_EINVAL);
    VG_CHECK_ARG(groups >= 1 && rows % groups == 0, VG_EINVAL);
    VG_CHECK_ARG(y8 == nullptr || (dtype == VG_BF16 && (reinterpret_cast<uintptr_t>(y8) & 3u) == 0), VG_EINVAL);
    const int64_t nvec = rows * C / 4;
    const int64_t nvg = nvec / groups;
    if (dtype == VG_F32)
        vg_launch_timed(4, bn_act_fwd_kernel<VG_F32>, dim3(ew_blocks(nvec)), dim3(256), 0, vg_stream(stream), x, y,
                           scale, shift, nvec, C / 4, act, slope, nvg, gstride, (uint32_t*)nullptr);
    else
        vg_launch_timed(4, bn_act_fwd_kernel<VG_BF16>, dim3(ew_blocks(nvec)), dim3(256), 0, vg_stream(stream), x, y,
                           scale, shift, nvec, C / 4, act, slope, nvg, gstride, reinterpret_cast<uint32_t*>(y8));
    return VG_LAUNCH_RC();
}

extern "C" int vg_channel_stats(const void* x, int64_t rows, int C, float* stats, int stats_capacity,
                                int* nparts_out, int dtype, void* stream) {
    int rc = check_rows_c(x, rows, C, dtype);
    if (rc) return rc;
    return launch_reduce<0>(x, nullptr, nullptr, nullptr, nullptr, nullptr, rows, C, 0, 0.f, stats, stats_capacity,
                            nparts_out, dtype, vg_stream(stream));
}

extern "C" int vg_bn_act_backward_reduce(const void* x, const void* dy, const float* scale, const float* shift,
                                         const float* mean, const float* invstd, int64_t rows, int C, int act,
                                         float slope, float* partial, int partial_capacity, int* nparts_out,
                                         int groups, int64_t gstride, int dtype, void* stream) {
    int rc = check_rows_c(x, rows, C, dtype);
    if (rc) return rc;
    VG_CHECK_ARG(dy && scale && shift && mean && invstd, VG_EINVAL);
    return launch_reduce<1>(x, dy, scale, shift, mean, invstd, rows, C, act, slope, partial, partial_capacity,
                            nparts_out, dtype, vg_stream(stream), groups, gstride);
}

extern "C" int vg_bn_backward_finalize(const float* partial, int nparts, int C, int64_t count, const float* gamma,
                                       const float* invstd, float* dgamma, float* dbeta, int accumulate, float* coef,
                                       void* stream) {
    VG_CHECK_ARG(partial && nparts > 0 && C > 0 && count > 0 && invstd && coef, VG_EINVAL);
    vg_launch_timed(4, bn_bwd_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_PL), 0, vg_stream(stream), partial, nparts, C,
                       (double)count, gamma, invstd, dgamma, dbeta, accumulate, coef);
    return VG_LAUNCH_RC();
}

extern "C" int vg_bn_act_backward_apply(const void* x, const void* dy, void* dx, const float* scale,
                                        const float* shift, const float* mean, const float* invstd, const float* coef,
                                        int64_t rows, int C, int act, float slope, int groups, int64_t gstride,
                                        int64_t cstride, int dtype, void* stream) {
    int rc = check_rows_c(x, rows, C, dtype);
    if (rc) return rc;
    VG_CHECK_ARG(dy && dx && scale && shift && mean && invstd && coef, VG_EINVAL);
    VG_CHECK_ARG(groups >= 1 && rows % groups == 0, VG_EINVAL);
    const int64_t nvec = rows * C / 4;
    const int64_t nvg = nvec / groups;
    if (dtype == VG_F32)
        vg_launch_timed(4, bn_act_bwd_apply_kernel<VG_F32>, dim3(ew_blocks(nvec)), dim3(256), 0, vg_stream(stream), x,
                           dy, dx, scale, shift, mean, invstd, coef, nvec, C / 4, C, act, slope, nvg, gstride, cstride);
    else
        vg_launch_timed(4, bn_act_bwd_apply_kernel<VG_BF16>, dim3(ew_blocks(nvec)), dim3(256), 0, vg_stream(stream), x,
                           dy, dx, scale, shift, mean, invstd, coef, nvec, C / 4, C, act, slope, nvg, gstride, cstride);
    return VG_LAUNCH_RC();
}

extern "C" int vg_act_backward(const void* x, const void* dy, void* dx, int64_t n, int act, float slope, int dtype,
                               void* stream) {
    VG_CHECK_ARG(x && dy && dx && n > 0 && n % 4 == 0, VG_EINVAL);
    VG_CHECK_ARG(dtype == VG_F32 || dtype == VG_BF16, VG_ENOSUP);
    const int64_t nvec = n / 4;
    if (dtype == VG_F32)
        vg_launch_timed(4, act_bwd_kernel<VG_F32>, dim3(ew_blocks(nvec)), dim3(256), 0, vg_stream(stream), x, dy, dx,
                           nvec, act, slope);
    else
        vg_launch_timed(4, act_bwd_kernel<VG_BF16>, dim3(ew_blocks(nvec)), dim3(256), 0, vg_stream(stream), x, dy, dx,
                           nvec, act, slope);
    return VG_LAUNCH_RC();
}

extern "C" int vg_bias_grad(const void* dy, int64_t rows, int C, int NC, float* dbias, int accumulate, float* ws,
                            int ws_capacity, int dtype, void* stream) {
    int rc = check_rows_c(dy, rows, C, dtype);
    if (rc) return rc;
    VG_CHECK_ARG(dbias && ws && NC > 0 && NC <= C, VG_EINVAL);
    int nparts = 0;
    rc = launch_reduce<0>(dy, nullptr, nullptr, nullptr, nullptr, nullptr, rows, C, 0, 0.f, ws, ws_capacity, &nparts,
                          dtype, vg_stream(stream));
    if (rc) return rc;
    vg_launch_timed(4, bias_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_PL), 0, vg_stream(stream), ws, nparts, C, NC,
                       dbias, accumulate);
    return VG_LAUNCH_RC();
}

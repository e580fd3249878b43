/*
 * vaegan_hip.h -- C ABI of the MI355X (gfx950) VAE-GAN training-path kernels.
 *
 * Drop-in boundary: the reference has no FFI layer -- its hot path is
 * the nn.Module / Optimizer API used by vaegan_code.py:65-135, and every FLOP executes inside
 * third-party ATen kernels.  This library replaces those ATen kernels.  Each entry point
 * below names the reference call site(s) it serves.  Plain pointers + sizes only; all
 * pointers are DEVICE pointers unless stated; every call is asynchronous on `stream`
 * (a hipStream_t passed as void*) and performs no allocation and no host synchronisation,
 * so call sequences can be captured into a hipGraph.
 *
 * Return value: 0 on success, a negative VG_E* code for rejected arguments (shape/alignment
 * checks are done on the host BEFORE launch), or a positive hipError_t from the launch.
 *
 * Internal activation layout: NHWC ("pixel-major"), channel count padded to a multiple of
 * 16 bytes (pad channels are zero).  dtype: VG_F32 (exact f32 MFMA, the parity path) or
 * VG_BF16 (bf16 storage, f32 accumulate).  Parameters/gradients/optimizer state are f32 in
 * the reference layouts (OIHW / [Cin][Cout][kh][kw]); `vg_pack_weights` produces the
 * K-major operand copies the GEMM kernels read.
 */
#ifndef VAEGAN_HIP_H
#define VAEGAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VG_F32  0
#define VG_BF16 1
#define VG_FP8  2   /* OCP e4m3fn operands, f32 accumulate, bf16 output: vg_gather_gemm fprop only (BASELINE configs[4],
                     * a roofline run -- the reference has no fp8 semantics).  X: [..][IC] fp8, IC % 16 == 0; W: packed
                     * [nphase][N][Kp] fp8 holding weight * 2^VG_FP8_WSHIFT (N(0, 0.02) weights would sit in e4m3's
                     * subnormals); the block-scaled MFMA (v_mfma_scale_f32_16x16x128_f8f6f4) undoes the shift through
                     * its E8M0 operand scale.  Y, bias, stats as for VG_BF16 (Y is bf16). */
#define VG_FP8_WSHIFT 6

#define VG_EINVAL   (-1)   /* bad shape / size / flag                                  */
#define VG_EALIGN   (-2)   /* pointer or channel count violates the 16-byte contract   */
#define VG_ENOSUP   (-3)   /* unsupported configuration                                */

#define VG_MAX_PHASE 4

/* Activation codes for the fused BN/activation kernels. */
#define VG_ACT_NONE    0
#define VG_ACT_RELU    1   /* nn.ReLU(True)          gan_code.py:23-43            */
#define VG_ACT_LRELU   2   /* nn.LeakyReLU(slope)    main_vae.py:25, gan_code.py:62-82 */
#define VG_ACT_TANH    3   /* nn.Tanh()              gan_code.py:50 (vg_tnconv epilogue only) */

#define VG_ABI_VERSION 10  /* 2: vg_pack_desc.tile_start, SyncBN / WGAN / data-path entry points; 3: in-kernel noise (vg_*_rng);
                             4: vg_bn_finalize_act_forward, vg_bn_backward_finalize_apply;
                             7: vg_bce_pair_forward_backward;
                             8: vg_head_backward; round-3 prune -- the opt-in experiments of ABI 5 / 6 that measured slower (input prologue of
                                vg_tn_desc / vg_ew_desc, vg_gg_desc.bnb_*) are gone from the descriptors; vg_reload_switches;
                             9: vg_step_prologue, vg_adam_step(lr < 0);
                             10 (round 4): vg_adam_apply replaces the lr < 0 overload of vg_adam_step (which now rejects it) */
int vg_abi_version(void);
/* The library reads its optional kernel-selection switches (VG_* environment variables, DESIGN.md "Runtime switches")
 * ONCE, when it is loaded; nothing on a launch path calls getenv.  A process that changes one of them afterwards
 * (tests, A/B scripts) calls this to have them read again.  Returns 0. */
int vg_reload_switches(void);
/* Live kernel timing for the roofline report: while enabled, gather-GEMM (family 0) and wgrad (family 1)
 * launches carry a HIP start/stop event pair on their stream (hipExtLaunchKernelGGL); collect() synchronises,
 * sums the kernel times in ms, returns the launch count and resets the family.  Do not enable during capture. */
int vg_timing_enable(int on);
/* Kernel launches issued by this library since it was loaded (every hipLaunchKernelGGL / hipExtLaunchKernelGGL of csrc/;
 * memsets and copies are runtime calls, not counted).  bench.py: launches per training iteration. */
uint64_t vg_launch_count(void);
int vg_timing_collect(int family, double* total_ms /* host */, int* launches /* host */);
/* Which tile configuration the launcher would pick (for tests / bench reporting). */
const char* vg_build_info(void);

/* ------------------------------------------------------------------------------------------
 * Gather-GEMM: the one implicit-GEMM kernel behind
 *   nn.Conv2d forward            (main_vae.py:23, gan_code.py:61-84)
 *   nn.ConvTranspose2d forward   (gan_code.py:21-49)  -- 4-phase sub-pixel form for k4 s2 p1
 *   their data gradients (autograd convolution_backward dgrad of vaegan_code.py:104,133)
 *   nn.Linear forward / dgrad    (main_vae.py:47-48, 55-56) -- as a full-extent convolution
 *
 *   Y[b, gy*OSY+ooy[p], gx*OSX+oox[p], n] = bias[n] +
 *        sum_{a<TH, c<TW, ci<IC} X[b, gy*SY + y0[p] + DY*a, gx*SX + x0[p] + DX*c, ci]
 *                                 * W[p][n][(a*TW + c)*IC + ci]
 *   for p < nphase, b < B, gy < GH, gx < GW, n < N; out-of-range input pixels read as 0,
 *   out-of-range output pixels are skipped.  Optionally emits per-channel partial
 *   (sum y, sum y*y) slabs for train-mode BatchNorm (nn.BatchNorm2d, main_vae.py:24).
 * ---------------------------------------------------------------------------------------- */
typedef struct vg_gg_desc {
    const void* X;        /* [B][IH][IW][IC]                                   */
    const void* W;        /* packed [nphase][N][Kp] (Kp = padded TH*TW*IC)     */
    void*       Y;        /* [B][OH][OW][OC]                                   */
    const float* bias;    /* [N] or NULL                                       */
    float*      stats;    /* partial slabs [nparts][2][N] or NULL              */
    int32_t B, GH, GW;
    int32_t IH, IW, IC;
    int32_t SY, SX, DY, DX, TH, TW;
    int32_t y0[VG_MAX_PHASE], x0[VG_MAX_PHASE];
    int32_t N, Kp;
    int32_t OH, OW, OC, OSY, OSX;
    int32_t ooy[VG_MAX_PHASE], oox[VG_MAX_PHASE];
    int32_t nphase;
    int32_t stats_capacity;   /* number of [2][N] slabs `stats` can hold       */
    float*  ws;               /* optional split-K workspace (f32 partial tiles) */
    int64_t ws_bytes;
    const void* zeros;        /* >= 64 zero bytes, 16-byte aligned: source of out-of-image taps on the LDS-DMA path
                                 (NULL selects the register-staged path)       */
    int32_t act;              /* VG_ACT_*: activation applied in the epilogue (layers WITHOUT BatchNorm, e.g. the
                                 Discriminator's first Conv2d + LeakyReLU(0.2), gan_code.py:61-62); not with `stats` */
    float   act_slope;
    /* Fused activation BACKWARD of the layer below (optional; a data-gradient launch whose output Y is
     * dL/d(activated output) of a BatchNorm-less layer): Y is multiplied by act'(mask_x) on its way out, mask_x being
     * that layer's (activated) output in the same [B][OH][OW][OC] layout -- saves the separate vg_act_backward pass. */
    const void* mask_x;
    int32_t mask_act;
    float   mask_slope;
} vg_gg_desc;

/* Number of stats slabs vg_gather_gemm will write for this descriptor (host-only query). */
int vg_gather_gemm_nparts(const vg_gg_desc* d, int dtype);
/* Kernel family a launch of this descriptor belongs to (for the roofline accounting of vg_timing_*):
 * 0 = gather-GEMM (MFMA-bound), 2 = edge layer (3-channel image side, HBM-bound: conv_narrowk.hpp). */
int vg_gather_gemm_family(const vg_gg_desc* d, int dtype);
/* Rows (M) covered by one statistics slab = the M edge of the tile the launcher picks (host-only query). */
int vg_gather_gemm_tile_m(const vg_gg_desc* d, int dtype);
/* Bytes of split-K workspace this launch can use (0: the launcher will not split K).  Skinny problems -- few
 * output tiles but a long K, e.g. the data gradient of the 1x1-input ConvTranspose2d (M=B, K=16*1024) -- are
 * split along K over gridDim.z; the f32 partial tiles are summed in fixed order by a second kernel. */
int64_t vg_gather_gemm_ws_bytes(const vg_gg_desc* d, int dtype);
int vg_gather_gemm(const vg_gg_desc* d, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Weight gradient (autograd convolution_backward wgrad; Linear weight grad):
 *   dW[np*s_np + cq*s_cq + (a*TW+c)*s_t] (+)= sum_{b,gy,gx}
 *        P[b, gy, gx, np] * Q[b, gy*SY + y0 + DY*a, gx*SX + x0 + DX*c, cq]
 * P is dense over the (GH,GW) grid, Q is gathered.  Conv2d: P=dY, Q=X; ConvTranspose2d:
 * P=X, Q=dY.  Output f32 in the reference parameter layout.  Deterministic: split-M partial
 * slabs in `ws` are reduced in fixed order.  accumulate!=0 adds into dW (autograd semantics
 * of two D passes summed into one .grad, vaegan_code.py:99-104).
 * ---------------------------------------------------------------------------------------- */
typedef struct vg_wg_desc {
    const void* P;        /* [B][GH][GW][PC]                                   */
    const void* Q;        /* [B][QH][QW][QC]                                   */
    float*      dW;       /* f32, reference layout                             */
    float*      ws;       /* workspace, ws_bytes                               */
    int64_t     ws_bytes;
    int32_t B, GH, GW, PC, NP;        /* NP real rows (<= PC)                   */
    int32_t QH, QW, QC, NQ;           /* NQ real channels (<= QC)               */
    int32_t SY, SX, DY, DX, TH, TW, y0, x0;
    int32_t s_np, s_cq, s_t;          /* element strides into dW               */
    int32_t accumulate;
    const void* zeros;                /* >= 64 zero bytes (16-byte aligned) for the LDS-DMA path; NULL = register path */
} vg_wg_desc;

int64_t vg_wgrad_ws_bytes(const vg_wg_desc* d, int dtype);
int vg_wgrad(const vg_wg_desc* d, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Operand packing: f32 parameter tensor (reference layout) -> K-major GEMM operand
 *   dst[p][n][(a*TW+c)*IC + ci] = src[n*s_n + ci*s_c + kh(p,a)*KW + kw(p,c)]   (0 in padding)
 *   kh(p,a) = kh0[p] + kh_step*a,  kw(p,c) = kw0[p] + kw_step*c
 * tap_in_n != 0 selects the "taps are part of N" form used by the 1x1-input ConvTranspose2d
 * (gan_code.py:21): dst[(kh*KW+kw)*CO + co][ci] = src[ci*s_c + co*s_n + kh*KW+kw].
 * ---------------------------------------------------------------------------------------- */
typedef struct vg_pack_desc {
    const float* src;
    void*        dst;
    int32_t nphase, N, C, IC, TH, TW, Kp;
    int32_t s_n, s_c, KW;
    int32_t kh0[VG_MAX_PHASE], kw0[VG_MAX_PHASE], kh_step, kw_step;
    int32_t tap_in_n, KHW;          /* KHW = taps of the FULL kernel (KH*KW) */
    int32_t tile_start;             /* vg_pack_weights_multi only: first flat tile of this descriptor */
} vg_pack_desc;
int vg_pack_weights(const vg_pack_desc* d, int dtype, void* stream);
/* Same, for a whole network in one launch: `descs_dev` is an array of n <= 64 descriptors in DEVICE memory (their
 * src/dst pointers are stable: parameters live in the optimizer's flat buffer).  The launch is one workgroup
 * per tile: descriptor i must carry tile_start = sum of vg_pack_tile_count() of the descriptors before it, and
 * total_tiles is the sum over all of them. */
int vg_pack_tile_count(const vg_pack_desc* d);
int vg_pack_weights_multi(const vg_pack_desc* descs_dev, int n, int64_t total_tiles, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm (train / eval) + activation, NHWC rows = B*H*W, C channels (C % 4 == 0).
 * nn.BatchNorm2d semantics: biased batch variance for normalisation,
 * unbiased for running_var, running = (1-m)*running + m*batch, eps inside the sqrt.
 * ---------------------------------------------------------------------------------------- */
/* Reduce the conv epilogue's partial slabs -> mean/invstd, scale/shift; update running stats. */
int vg_bn_finalize(const float* stats, int nparts, int C, int64_t count,
                   const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float momentum, float eps,
                   float* mean, float* invstd, float* scale, float* shift, void* stream);
/* The same two reductions for `groups` independent row blocks in ONE launch (a Discriminator iteration's real and
 * fake batches run as one 2B-row pass): slabs of group g start at stats + g*nparts_per_group*2*C; coeffs is
 * [groups][4][C] = mean, invstd, scale, shift; coef is [groups][3][C]; running statistics and dgamma/dbeta are
 * updated group after group, the order in which the reference's separate calls would update them. */
int vg_bn_finalize_grouped(const float* stats, int nparts_per_group, int groups, int C, int64_t count_per_group,
                           const float* gamma, const float* beta, float* running_mean, float* running_var,
                           float momentum, float eps, float* coeffs, void* stream);
/* vg_bn_finalize_grouped + vg_bn_act_forward in ONE launch, for small layers (bf16, <= 200 slab rows per group, tensor <= 9 MB,
 * C % 64 == 0): every workgroup of the elementwise pass re-derives the coefficients of its 64
 * channels from the slab (same double-precision sums, another order: coefficients may differ from vg_bn_finalize_grouped
 * in the last bit).  Writes coeffs [groups][4][C] and the running statistics like vg_bn_finalize_grouped.
 * vg_bn_finalize_act_forward_supported() says whether a shape qualifies (VG_BN_FUSED_FWD=0 turns the path off); the
 * launch returns VG_ENOSUP otherwise.  rows = all groups' rows; x, y: [rows][C] bf16. */
int vg_bn_finalize_act_forward_supported(int nparts_per_group, int groups, int C, int64_t rows, int dtype);
int vg_bn_finalize_act_forward(const void* x, void* y, const float* stats, int nparts_per_group, int groups, int C,
                               int64_t rows, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, float momentum, float eps, float* coeffs, int act, float slope,
                               int dtype, void* stream);
/* Backward twin: vg_bn_backward_finalize_grouped + vg_bn_act_backward_apply in one launch, same eligibility
 * (vg_bn_finalize_act_forward_supported with the column-reduce partial count); VG_ENOSUP otherwise. */
int vg_bn_backward_finalize_apply(const void* x, const void* dy, void* dx, const float* partial, int nparts_per_group,
                                  int groups, int C, int64_t rows, const float* gamma, const float* coeffs,
                                  float* dgamma, float* dbeta, int accumulate, int act, float slope, int dtype,
                                  void* stream);
int vg_bn_backward_finalize_grouped(const float* partial, int nparts_per_group, int groups, int C,
                                    int64_t count_per_group, const float* gamma, const float* coeffs,
                                    float* dgamma, float* dbeta, int accumulate, float* coef, void* stream);
/* Synchronised BatchNorm (statistics over the global batch of a one-process-per-GPU job).
 * The reference is single-process (vaegan_code.py:29-35), so "the batch" of nn.BatchNorm2d is the whole batch;
 * these three calls let N ranks reproduce that: vg_slab_sums -> host all-reduce(SUM) of the f64 [2][C] vector
 * -> vg_bn_finalize_sums with the GLOBAL count.  Backward: vg_bn_act_backward_reduce -> vg_slab_sums ->
 * all-reduce -> vg_bn_backward_finalize_sums (dgamma/dbeta from the LOCAL sums: the gradient all-reduce
 * averages them afterwards; dx coefficients from the GLOBAL sums). */
int vg_slab_sums(const float* slabs, int nparts, int C, double* sums, void* stream);
int vg_bn_finalize_sums(const double* sums, int C, int64_t count, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float eps,
                        float* mean, float* invstd, float* scale, float* shift, void* stream);
int vg_bn_backward_finalize_sums(const double* global_sums, const double* local_sums, int C, int64_t count,
                                 const float* gamma, const float* invstd, float* dgamma, float* dbeta,
                                 int accumulate, float* coef, void* stream);
/* Eval mode: scale/shift from running statistics. */
int vg_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, int C,
                      float* scale, float* shift, void* stream);
/* y = act(scale[c]*x + shift[c]); scale/shift NULL -> pure activation.
 * groups > 1: the rows are `groups` equal, independent row blocks (e.g. the real and the fake batch of one
 * Discriminator iteration, vaegan_code.py:96-97, run as one launch) with their own coefficient sets:
 * group g reads scale[g*gstride + c] (same for shift / mean / invstd below). */
int vg_bn_act_forward(const void* x, void* y, const float* scale, const float* shift,
                      int64_t rows, int C, int act, float slope, int groups, int64_t gstride,
                      int dtype, void* stream);
/* Same, additionally writing y8 = e4m3(y) (same [rows][C] layout, one byte per element; VG_BF16 only): the operand of
 * the next layer's VG_FP8 forward GEMM, produced in the pass that produces the bf16 activation. */
int vg_bn_act_forward_fp8(const void* x, void* y, void* y8, const float* scale, const float* shift, int64_t rows,
                          int C, int act, float slope, int groups, int64_t gstride, int dtype, void* stream);
/* Standalone per-channel statistics of an NHWC tensor (used when no conv epilogue produced them). */
int vg_channel_stats(const void* x, int64_t rows, int C, float* stats, int stats_capacity,
                     int* nparts_out, int dtype, void* stream);
/* Backward pass 1: dz = dy_act * act'(z), z = scale*x+shift; partial sums of dz and dz*xhat.
 * With groups > 1 the slabs of group g are parts [g*nparts_out, (g+1)*nparts_out). */
int vg_bn_act_backward_reduce(const void* x, const void* dy, const float* scale, const float* shift,
                              const float* mean, const float* invstd,
                              int64_t rows, int C, int act, float slope,
                              float* partial, int partial_capacity, int* nparts_out,
                              int groups, int64_t gstride, int dtype, void* stream);
/* Backward finalize: dgamma, dbeta (accumulate optional) and the two per-channel coefficients. */
int vg_bn_backward_finalize(const float* partial, int nparts, int C, int64_t count,
                            const float* gamma, const float* invstd,
                            float* dgamma, float* dbeta, int accumulate,
                            float* coef /* [3][C]: a, b, c */, void* stream);
/* Backward pass 2: dx = a[c]*dz - b[c]*xhat - c[c]   (dz recomputed from dy, x); group g uses coef + g*cstride. */
int vg_bn_act_backward_apply(const void* x, const void* dy, void* dx,
                             const float* scale, const float* shift,
                             const float* mean, const float* invstd, const float* coef,
                             int64_t rows, int C, int act, float slope, int groups, int64_t gstride,
                             int64_t cstride, int dtype, void* stream);
/* Activation-only backward (first Discriminator layer has no BN, gan_code.py:61-62). */
int vg_act_backward(const void* x, const void* dy, void* dx, int64_t n, int act, float slope,
                    int dtype, void* stream);
/* dbias[c] (+)= sum over rows of dy[row][c]  (Conv2d bias grad, main_vae.py:23; Linear bias). */
int vg_bias_grad(const void* dy, int64_t rows, int C, int NC, float* dbias, int accumulate,
                 float* ws, int ws_capacity, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Edge layers (3-channel side at the image boundary; HBM-bound)
 * ---------------------------------------------------------------------------------------- */
/* Narrow-N transposed convolution, bf16:
 *     Y[b][oy][ox][n] = sum_{c,kh,kw} X[b][iy][ix][c] * W[c][n][kh][kw],  oy = iy*S - P + kh, ox = ix*S - P + kw
 * = nn.ConvTranspose2d(C, N, K, S, P).forward (the Generator's last layer, gan_code.py:49: C->3, k3 s1 p1) and the
 * data gradient of nn.Conv2d(N, C, K, S, P) (the image gradient below the Discriminator's first layer,
 * gan_code.py:61: 3->C, k4 s2 p1, reached by loss.backward() at vaegan_code.py:133).  N <= 4, K*K*N <= 64.
 * Computed as ONE GEMM per input pixel over the channels, Pm[pix][(kh,kw,n)] = X[pix][:] . Wp[(kh,kw,n)][:], followed
 * by the K*K-tap col2im sum inside the workgroup's LDS tile: every input byte is read once.
 * Wp: [K*K*N][Wpitch] bf16, row j = (kh*K + kw)*N + n, = vg_pack_weights with tap_in_n=1 of the [C][N][K][K]
 * (ConvTranspose2d) / [C][N][K][K]-viewed (Conv2d [Cout=C][Cin=N][K][K]) weight.
 * Outputs: Y NHWC bf16 [B][OH][OW][OC=8] (channels >= N zero) and/or Y_nchw f32 [B][N][OH][OW].
 * act = VG_ACT_TANH applies tanh (gan_code.py:50) to both; with noise (eps NCHW f32, or rng+draw: in-kernel
 * N(0,1)) Y becomes act(.) + sigma*noise while Y_nchw stays act(.) -- vaegan_code.py:83 and :92 in one pass. */
typedef struct vg_tn_desc {
    const void* X;          /* [B][IH][IW][C] bf16, C = 32 or 64 */
    const void* Wp;
    void* Y;                /* or NULL */
    float* Y_nchw;          /* or NULL */
    const float* eps;       /* or NULL */
    const uint64_t* rng;    /* or NULL */
    int32_t draw;
    float sigma;
    int32_t B, IH, IW, C, N, K, S, P, OH, OW, OC, Wpitch, act;
} vg_tn_desc;
/* Weight gradient of the edge layers, bf16 operands, f32 result:
 *     dW[c*s_c + n*s_n + kh*K + kw] (+)= sum_{b,py,px} Wd[b][py][px][c] * Nr[b][py*S - P + kh][px*S - P + kw][n]
 * Wd: the WIDE operand [B][WH][WW][C] (C = 32 | 64): dY of nn.Conv2d(N, C, K, S, P) (gan_code.py:61, main_vae.py:23 --
 * then dW is the [C][N][K][K] weight gradient, s_c = N*K*K, s_n = K*K) or the input of nn.ConvTranspose2d(C, N, K, S, P)
 * (gan_code.py:49 -- dW is [C][N][K][K] as well).  Nr: the 3-channel tensor on the other side, [B][NH][NW][8] bf16 with
 * channels >= N zero (the image, or the image gradient).  N <= 3, K = 3 | 4, S = 1 | 2.
 * ws: vg_edge_wgrad_ws_bytes() bytes of scratch (one [K*K*4 padded][C] f32 partial per workgroup, summed in fixed
 * order: bitwise reproducible).  zeros: >= 64 readable zero bytes. */
typedef struct vg_ew_desc {
    const void* Wd;
    const void* Nr;
    float* dW;
    float* ws;
    int64_t ws_bytes;
    const void* zeros;
    int32_t B, WH, WW, C, NH, NW, N, K, S, P, s_c, s_n, accumulate;
} vg_ew_desc;
int64_t vg_edge_wgrad_ws_bytes(const vg_ew_desc* d);
int vg_edge_wgrad(const vg_ew_desc* d, void* stream);
int vg_tnconv_supported(const vg_tn_desc* d);     /* 0 if vg_tnconv takes this shape, else the error code */
int vg_tnconv(const vg_tn_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Layout / pointwise / losses
 * ---------------------------------------------------------------------------------------- */
/* NCHW f32 [B][C][H][W] -> NHWC dtype [B][H][W][CP] (pad channels zero); optional fused
 * instance noise out = x + sigma*eps (vaegan_code.py:91-92), eps NCHW f32 or NULL. */
int vg_nchw_to_nhwc(const float* x, const float* eps, float sigma, void* y,
                    int B, int C, int H, int W, int CP, int dtype, void* stream);
/* Data path (dataset_code.py:137-178): the whole image set lives in HBM as u8 [N][H][W][C] (CelebA-HQ 256x256:
 * 5.9 GB of the 288 GB); a batch is assembled on the device from the sampler's indices with ToTensor +
 * Normalize((0.5,),(0.5,)) arithmetic (dataset_code.py:147-150): out[b][c][h][w] = (u/255 - 0.5)/0.5, f32 NCHW. */
int vg_gather_normalize_u8(const uint8_t* images, int64_t N, const int64_t* idx, int B, int C, int H, int W,
                           float* out, void* stream);
/* Denoise-evaluation input (vaegan_code.py:153-154): noisy = clamp(x + sigma*eps, lo, hi), written both as the
 * NHWC engine tensor and (optionally, y_nchw != NULL) as NCHW f32 for the caller. */
int vg_noisy_clamp_to_nhwc(const float* x, const float* eps, float sigma, float lo, float hi, void* y,
                           float* y_nchw, int B, int C, int H, int W, int CP, int dtype, void* stream);
/* NHWC dtype -> NCHW f32, optional tanh (gan_code.py:50). */
int vg_nhwc_to_nchw(const void* x, float* y, int B, int C, int H, int W, int CP,
                    int apply_tanh, int dtype, void* stream);
/* The Generator's output in one pass (vaegan_code.py:83,92): y_nchw = tanh(x) (NCHW f32) and
 * y_noisy_nhwc = tanh(x) + sigma*eps (NHWC dtype, padded channels 0), eps NCHW f32. */
int vg_nhwc_tanh_to_nchw_noisy(const void* x, float* y_nchw, const float* eps, float sigma, void* y_noisy_nhwc,
                               int B, int C, int H, int W, int CP, int dtype, void* stream);
/* Gradient of the above: dy NCHW f32 -> dx NHWC dtype, optionally * (1 - t*t) with t = tanh output (NCHW f32). */
int vg_nchw_grad_to_nhwc(const float* dy, const float* tanh_out, void* dx,
                         int B, int C, int H, int W, int CP, int dtype, void* stream);
/* Same with a second gradient branch that already sits in the engine layout: dx = (dy + add_nhwc) * (1 - t*t).
 * vaegan_code.py:117,133: the reconstruction's gradient is d(MSE)/d(recon) (NCHW f32) plus the Discriminator's
 * input gradient through the instance-noise add (NHWC) -- one pass instead of layout change + add + layout change. */
int vg_nchw_grad_add_to_nhwc(const float* dy, const void* add_nhwc, const float* tanh_out, void* dx,
                             int B, int C, int H, int W, int CP, int dtype, void* stream);
/* Reparameterisation (vaegan_code.py:75-77): lv=clamp(logvar,-10,10); z=mu+exp(.5 lv)*eps.
 * mulv: [B][MP] dtype, the fused fc_mu|fc_logvar output (main_vae.py:55-56): columns [0,L) = mu,
 * [L,2L) = logvar.  eps: [B][L] f32.  z: [B][ZP] dtype (pad = 0).  lv_clamped: [B][L] f32. */
int vg_reparam_forward(const void* mulv, const float* eps, void* z, float* lv_clamped,
                       int B, int L, int MP, int ZP, int dtype, void* stream);
/* KL (vaegan_code.py:114): out[0] = -0.5*sum(1+lv-mu^2-exp(lv)) / divisor. */
int vg_kl_forward(const void* mulv, const float* lv_clamped, int B, int L, int MP, float divisor,
                  float* out, int dtype, void* stream);
/* d(mu|logvar) [B][MP] dtype from dz [B][ZP] dtype and kl_scale = alpha_kl*min(1,epoch/50)/B
 * (vaegan_code.py:114,117); the clamp passes gradient on [-10,10] only. */
int vg_reparam_kl_backward(const void* mulv, const float* lv_clamped, const float* eps,
                           const void* dz, float kl_scale, void* dmulv,
                           int B, int L, int MP, int ZP, int dtype, void* stream);
/* Discriminator head (gan_code.py:84-85,89): p[b] = sigmoid(dot(x[b,:], w)), x NHWC-flattened. */
int vg_dot_sigmoid_forward(const void* x, const void* w, float* p, int B, int K, int dtype, void* stream);
/* dlogit[b] = dp[b]*p*(1-p); dx[b,k] = dlogit[b]*w[k]  (dx NULL -> skipped). */
int vg_dot_sigmoid_backward(const float* p, const float* dp, const void* w, void* dx, float* dlogit,
                            int B, int K, int dtype, void* stream);
/* dw[k] (+)= sum_b dlogit[b]*x[b,k] -> f32 in reference layout via perm (k_nhwc -> offset). */
int vg_dot_wgrad(const void* x, const float* dlogit, float* dw, int B, int K, int C, int HW,
                 int accumulate, int dtype, void* stream);
/* nn.BCELoss (vaegan_code.py:46): mean_b -(t*max(log p,-100)+(1-t)*max(log(1-p),-100));
 * loss[0] (+)= value when accumulate; dp[b] = gscale*(p-t)/max(p*(1-p),1e-12)/B (dp NULL ok). */
int vg_bce_forward_backward(const float* p, float target, int B, float gscale,
                            float* loss, int accumulate, float* dp, void* stream);
/* The Discriminator loss of one update, both halves in one launch (vaegan_code.py:98-103; ABI 7): p = [B real | B fake],
 * loss[0] (+)= BCE(p[:B], target0) + BCE(p[B:], target1), dp likewise [2B] -- bit-identical to two vg_bce_forward_backward
 * calls, the second accumulating. */
int vg_bce_pair_forward_backward(const float* p, float target0, float target1, int B, float gscale,
                                 float* loss, int accumulate, float* dp, void* stream);
/* Backward of BCE(sigmoid(head)) in ONE launch (round 3; vaegan_code.py:99-104 and :115,133 behind the Discriminator's last
 * Conv2d, gan_code.py:84-85): for p = [B rows with target0 | B rows with target1] (groups = 2) or B rows with target0
 * (groups = 1):  loss[0] (+)= sum of the groups' BCE means;  dlogit[r] = gscale * (p - t) / max(p (1 - p), 1e-12) / B * p (1 - p);
 * dx[r,k] = dlogit[r] * w[k] (dx NULL: skipped);  dw[k] (+)= sum_r dlogit[r] * x[r,k] in the reference layout (dw NULL:
 * skipped);  dlogit optional output.  Bit-identical to vg_bce[_pair]_forward_backward + vg_dot_sigmoid_backward +
 * vg_dot_wgrad (same arithmetic, lane mappings and summation orders).  B * groups <= 4096, K % 4 == 0. */
int vg_head_backward(const float* p, const void* x, const void* w, void* dx, float* dw, float* dlogit, int B, int groups,
                     float target0, float target1, float gscale, float* loss, int accumulate_loss, int accumulate_dw,
                     int K, int C, int HW, int dtype, void* stream);
/* Sibling loop train_wgan (gan_code.py:306-315, :328): loss[0] (+)= sign*mean_b p[b]; dp[b] = sign*gscale/B. */
int vg_mean_forward_backward(const float* p, float sign, int B, float gscale,
                             float* loss, int accumulate, float* dp, void* stream);
/* WGAN weight clipping `p.data.clamp_(-c, c)` (gan_code.py:320-321) over a flat parameter buffer. */
int vg_clamp(float* p, int64_t n, float lo, float hi, void* stream);
/* nn.MSELoss(mean) (vaegan_code.py:47) on NCHW f32 tensors; d_a = gscale*2*(a-b)/n (NULL ok). */
int vg_mse_forward_backward(const float* a, const float* b, int64_t n, float gscale,
                            float* loss, float* d_a, float* ws, int ws_capacity, void* stream);
/* Mean SSIM of two NCHW f32 image batches in [-1,1] (rescaled to [0,1] as vaegan_code.py:170-174 does):
 * gaussian 11x11, sigma 1.5, k1 .01, k2 .03, data_range 1, 5-pixel border cropped.  out[0] = mean. */
int vg_ssim(const float* a, const float* b, int B, int C, int H, int W, float* out, float* ws, int ws_capacity,
            void* stream);
/* ------------------------------------------------------------------------------------------
 * In-kernel N(0,1) noise: the three `torch.randn_like` draws of an iteration (vaegan_code.py:77 eps of the
 * reparameterisation, :91 instance noise on the real batch, :92 on the reconstruction) generated where they are
 * consumed instead of being materialised by a separate generator launch.
 * rng: device memory, two 64-bit words {seed, iteration counter}.  Counter-based Philox4x32-10 keyed by the seed;
 * counter = (element index >> 2 in the reference's NCHW / [B][L] order, draw id 0..255, iteration counter); one block
 * gives four normals (two Box-Muller pairs, cosine and sine of each): element i is component i & 3 of block i >> 2.
 * The `_rng` forms of the consuming kernels are identical to their eps-pointer forms with
 * eps[i] = N(seed, iteration, draw, i); vg_randn materialises exactly that tensor (tests, and callers that want
 * the draw itself).  vg_rng_advance (one thread) bumps the iteration counter: launch it once at the top of every
 * iteration, inside the captured graph, so that every replay draws fresh noise while forward and backward of one
 * iteration see the same eps.  Device streams can never equal the reference's CPU generator:
 * parity runs inject host noise through the eps-pointer forms.
 * ---------------------------------------------------------------------------------------- */
int vg_rng_advance(uint64_t* rng, void* stream);
int vg_randn(float* out, int64_t n, const uint64_t* rng, int draw, void* stream);
/* One pass over x (NCHW f32) -> y_noisy = x + sigma * noise AND y_plain = x, both NHWC bf16 with CP = 8 channels: the
 * Encoder's input and the Discriminator's noisy real batch (vaegan_code.py:74, :91).  Exactly one of eps (injected noise,
 * NCHW f32) / rng (+ draw).  VG_ENOSUP unless bf16, CP == 8, C <= 4, H * W % 4 == 0 (convert twice then). */
int vg_nchw_to_nhwc_pair(const float* x, const float* eps, const uint64_t* rng, int draw, float sigma, void* y_noisy,
                         void* y_plain, int B, int C, int H, int W, int CP, int dtype, void* stream);
/* vg_mse_forward_backward in two halves: the partial sums (+ gradient) here, the final sum inside the KL launch below
 * (vaegan_code.py:113-114 are evaluated back to back; the separate one-wave finalize launch was 4.7 us). */
int vg_mse_partial(const float* a, const float* b, int64_t n, float gscale, float* d_a, float* ws, int ws_capacity,
                   int* nparts_out, void* stream);
int vg_kl_forward_mse_final(const void* mulv, const float* lv_clamped, int B, int L, int MP, float divisor, float* out,
                            const float* mse_ws, int mse_nparts, int64_t mse_n, float* mse_loss, int dtype, void* stream);
int vg_nchw_to_nhwc_rng(const float* x, const uint64_t* rng, int draw, float sigma, void* y,
                        int B, int C, int H, int W, int CP, int dtype, void* stream);          /* vaegan_code.py:91 */
int vg_nhwc_tanh_to_nchw_noisy_rng(const void* x, float* y_nchw, const uint64_t* rng, int draw, float sigma,
                                   void* y_noisy_nhwc, int B, int C, int H, int W, int CP, int dtype,
                                   void* stream);                                             /* vaegan_code.py:83,92 */
int vg_reparam_forward_rng(const void* mulv, const uint64_t* rng, int draw, void* z, float* lv_clamped,
                           int B, int L, int MP, int ZP, int dtype, void* stream);            /* vaegan_code.py:75-78 */
int vg_reparam_kl_backward_rng(const void* mulv, const float* lv_clamped, const uint64_t* rng, int draw,
                               const void* dz, float kl_scale, void* dmulv, int B, int L, int MP, int ZP,
                               int dtype, void* stream);
/* hipMemsetAsync(p, 0, nbytes) on the stream: optimizer.zero_grad() (vaegan_code.py:103,131-132) over a flat buffer. */
int vg_memset_zero(void* p, int64_t nbytes, void* stream);
/* bf16 -> OCP e4m3fn, elementwise: y[i] = fp8(x[i] * 2^shift).  The fp8 copies of activations (shift 0) and of the
 * packed bf16 GEMM operands (shift VG_FP8_WSHIFT) that VG_FP8 launches of vg_gather_gemm read.  n % 8 == 0. */
int vg_cast_fp8(const void* x_bf16, void* y_fp8, int64_t n, int shift, void* stream);
/* BatchNorm(+activation) backward in ONE launch (csrc/bn_onepass.hip; bf16, C a power of two in [8, 1024], tensors of up to
 * 65 536 elements per CU): column sums, coefficients and dx = a*dz - b*xhat - c with the x / dy block of a workgroup held in
 * registers across a grid-wide exchange of the partial sums -- replaces vg_bn_act_backward_reduce +
 * vg_bn_backward_finalize_grouped + vg_bn_act_backward_apply (nn.BatchNorm2d / nn.LeakyReLU / nn.ReLU backward behind
 * loss.backward(), vaegan_code.py:104, :133; modules main_vae.py:24-25, gan_code.py:22-82).  coeffs: [groups][4][C] as
 * published by the forward pass (mean | invstd | scale | shift); dgamma / dbeta (+)= per `accumulate`, group after group.
 * slab: vg_bn_backward_onepass_ws_bytes() bytes of scratch; sync: 16 bytes that are ZERO before the first call and are left
 * zero by every call (sync[2] != 0 afterwards: a bounded grid-wide wait gave up -- results of that call are invalid).
 * Returns VG_ENOSUP where _supported() says 0 (the three-launch form then applies). */
int vg_bn_backward_onepass_supported(int64_t rows, int C, int groups, int dtype);
int64_t vg_bn_backward_onepass_ws_bytes(int64_t rows, int C, int groups, int dtype);
int vg_bn_backward_onepass(const void* x, const void* dy, void* dx, const float* coeffs, const float* gamma, float* dgamma,
                           float* dbeta, int accumulate, float* slab, unsigned* sync, int64_t rows, int C, int groups,
                           int act, float slope, int dtype, void* stream);
/* out = a + alpha*b (f32, n elements); used for gradient joins on NCHW images. */
int vg_axpy(const float* a, const float* b, float alpha, float* out, int64_t n, void* stream);
/* PSNR/SSIM support for the denoise path lives in vg_image_metrics. */

/* ------------------------------------------------------------------------------------------
 * Adam (torch.optim.Adam defaults as used at vaegan_code.py:42-44; arithmetic of torch 2.10
 * _single_tensor_adam): one launch over a flat f32 buffer.
 * state[0] = step count as float (incremented on device first), so the call is graph-replayable.
 * grad_scale multiplies g before use (1/world_size after an all-reduce SUM).
 * ---------------------------------------------------------------------------------------- */
int vg_adam_step(float* p, const float* g, float* m, float* v, int64_t n,
                 double lr, double beta1, double beta2, double eps, float grad_scale,
                 float* state /* [4] device */, void* stream);
/* A training iteration that steps several optimizers (vaegan_code.py:105, :134-135) can prepare all of them -- step
 * count + 1, bias corrections -- together with the noise generator's iteration counter (rng may be NULL) in ONE
 * single-thread launch at its top; vg_adam_apply then launches the update alone and uses `state` as it finds it.  Same
 * double arithmetic as the per-optimizer prepare: bit-identical updates.  (vg_adam_step rejects lr < 0: ABI 9 overloaded
 * a negative learning rate as "prepared", which a caller's typo could trigger silently.) */
int vg_adam_apply(float* p, const float* g, float* m, float* v, int64_t n, double beta1, double beta2, double eps,
                  float grad_scale, const float* state /* prepared by vg_step_prologue */, void* stream);
#define VG_PROLOGUE_MAX 4
int vg_step_prologue(uint64_t* rng, float* const* states, const double* lr, const double* beta1, const double* beta2,
                     int n, float* zero /* nzero floats set to 0 (the iteration's loss slots), or NULL */, int nzero,
                     void* stream);
/* The Encoder's and the Generator's optimizer step of one iteration (vaegan_code.py:134-135) in ONE launch: arrays of two
 * (p, g, m, v, n, betas, eps, grad_scale, prepared state); arithmetic and per-element work split of vg_adam_apply. */
int vg_adam_apply2(float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n,
                   const double* beta1, const double* beta2, const double* eps, const float* grad_scale,
                   const float* const* state, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VAEGAN_HIP_H */

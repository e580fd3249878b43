// Shared device/host helpers for the gfx950 VAE-GAN kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vaegan_hip.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define VG_CHECK_ARG(cond, code) do { if (!(cond)) return (code); } while (0)

// Every kernel launch of the library is counted (vg_launch_count(), C ABI): bench.py reports kernel launches per training
// iteration from it, live, next to the rocprofv3 trace that shows the same number.  hipLaunchKernelGGL is HIP's macro around
// the triple-chevron launch; it is re-defined here with the counter in front.
#include <atomic>
std::atomic<uint64_t>& vg_launch_counter();
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                       \
    do {                                                                                                        \
        vg_launch_counter().fetch_add(1, std::memory_order_relaxed);                                            \
        kernelName<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__);                      \
    } while (0)
#define VG_LAUNCH_RC() ((int)hipGetLastError())

static inline hipStream_t vg_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline bool vg_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- element traits -------------------------------------------------------------------------
template <int DT> struct ElemT;
template <> struct ElemT<VG_F32> {
    typedef float type;
    static constexpr int size = 4;
    static constexpr int per16 = 4;
    __device__ static __forceinline__ float to_f32(float v) { return v; }
    __device__ static __forceinline__ float from_f32(float v) { return v; }
};
template <> struct ElemT<VG_BF16> {
    typedef uint16_t type;
    static constexpr int size = 2;
    static constexpr int per16 = 8;
    __device__ static __forceinline__ float to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
    // plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32), MI355X_MICROARCH.md "Correctness boundaries"
    __device__ static __forceinline__ uint16_t from_f32(float v) {
        __bf16 b = (__bf16)v;
        return __builtin_bit_cast(uint16_t, b);
    }
};

template <> struct ElemT<VG_FP8> {          // OCP e4m3fn (gfx950); storage only: vg_gather_gemm's VG_FP8 operands
    typedef uint8_t type;
    static constexpr int size = 1;
    static constexpr int per16 = 16;
    __device__ static __forceinline__ float to_f32(uint8_t v) { return __builtin_amdgcn_cvt_f32_fp8((int)v, 0); }
    __device__ static __forceinline__ uint8_t from_f32(float v) {
        return (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false) & 0xff);
    }
};

// load / store a run of 4 consecutive elements as floats (16 B for f32, 8 B for bf16)
template <int DT> __device__ __forceinline__ float4 load4(const void* base, int64_t idx);
template <> __device__ __forceinline__ float4 load4<VG_F32>(const void* base, int64_t idx) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + idx);
}
template <> __device__ __forceinline__ float4 load4<VG_BF16>(const void* base, int64_t idx) {
    uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + idx);
    float4 o;
    o.x = __uint_as_float(r.x << 16);
    o.y = __uint_as_float(r.x & 0xffff0000u);
    o.z = __uint_as_float(r.y << 16);
    o.w = __uint_as_float(r.y & 0xffff0000u);
    return o;
}
template <int DT> __device__ __forceinline__ void store4(void* base, int64_t idx, float4 v);
template <> __device__ __forceinline__ void store4<VG_F32>(void* base, int64_t idx, float4 v) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + idx) = v;
}
template <> __device__ __forceinline__ void store4<VG_BF16>(void* base, int64_t idx, float4 v) {
    uint2 r;
    r.x = (uint32_t)ElemT<VG_BF16>::from_f32(v.x) | ((uint32_t)ElemT<VG_BF16>::from_f32(v.y) << 16);
    r.y = (uint32_t)ElemT<VG_BF16>::from_f32(v.z) | ((uint32_t)ElemT<VG_BF16>::from_f32(v.w) << 16);
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + idx) = r;
}
template <int DT> __device__ __forceinline__ float load1(const void* base, int64_t idx) {
    return ElemT<DT>::to_f32(reinterpret_cast<const typename ElemT<DT>::type*>(base)[idx]);
}
template <int DT> __device__ __forceinline__ void store1(void* base, int64_t idx, float v) {
    reinterpret_cast<typename ElemT<DT>::type*>(base)[idx] = ElemT<DT>::from_f32(v);
}

__device__ __forceinline__ float act_fwd(float z, int act, float slope) {
    if (act == VG_ACT_RELU) return z > 0.f ? z : 0.f;
    if (act == VG_ACT_LRELU) return z > 0.f ? z : z * slope;
    return z;
}
// derivative selected on the pre-activation z (torch: relu' = z>0, leaky' = z>0 ? 1 : slope)
__device__ __forceinline__ float act_bwd(float z, float g, int act, float slope) {
    if (act == VG_ACT_RELU) return z > 0.f ? g : 0.f;
    if (act == VG_ACT_LRELU) return z > 0.f ? g : g * slope;
    return g;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- optional live kernel timing (bench.py roofline leg) ------------------------------------------------------
// When enabled, the GEMM-class launches go through hipExtLaunchKernelGGL with a start/stop event pair, which
// brackets the kernel itself on its stream (no launch latency inside the bracket, unlike events recorded around
// the launch call).  Off by default and never on while a stream is being captured into a graph.
#include <hip/hip_ext.h>
#include <mutex>
#include <vector>
struct VgTiming {
    bool on = false;
    std::mutex mu;
    std::vector<hipEvent_t> pool;                                 // free events
    std::vector<std::pair<hipEvent_t, hipEvent_t>> rec[5];         // family 0: gather-GEMM, 1: wgrad, 2: edge layers (HBM-bound), 3: fp8 gather-GEMM, 4: BatchNorm / activation / bias-gradient passes (bn_act.hip, HBM-bound)
};
VgTiming& vg_timing();

template <typename K, typename... Args>
inline void vg_launch_timed(int family, K kernel, dim3 grid, dim3 block, size_t shm, hipStream_t s, Args... args) {
    VgTiming& t = vg_timing();
    if (!t.on) {
        hipLaunchKernelGGL(kernel, grid, block, shm, s, args...);
        return;
    }
    hipEvent_t e0, e1;
    {
        std::lock_guard<std::mutex> g(t.mu);
        auto get = [&]() { hipEvent_t e; if (t.pool.empty()) { (void)hipEventCreate(&e); } else { e = t.pool.back(); t.pool.pop_back(); } return e; };
        e0 = get(); e1 = get();
        t.rec[family].push_back({e0, e1});
    }
    vg_launch_counter().fetch_add(1, std::memory_order_relaxed);
    hipExtLaunchKernelGGL(kernel, grid, block, shm, s, e0, e1, 0, args...);
}

// ---- runtime switches -------------------------------------------------------------------------------------------
// Kernel-selection switches (all optional; the defaults are the measured optimum, DESIGN.md "Runtime switches").  They
// are read from the environment ONCE, when the library is loaded; vg_reload_switches() (C ABI) re-reads them -- tests
// and A/B scripts that flip a variable inside one process call it afterwards.  Nothing on a launch path calls getenv.
struct VgSwitches {
    int gg_dma;            // VG_GG_DMA            1: LDS-DMA staging in the gather-GEMM (0: register-staged kernels)
    int tile_min_wgs;      // VG_TILE_MIN_WGS      512: workgroups a launch must offer before a larger tile is chosen
    int gg_patch;          // VG_GG_PATCH          1: patch variant of the gather-GEMM
    int gg_patch64;        // VG_GG_PATCH64        1: its 64-column form
    int gg_patch32;        // VG_GG_PATCH32        1: its 32-column form
    int gg_phase4;         // VG_GG_PHASE4         1: all four phases per workgroup for transposed layers with <= 32 output channels (conv_phase4.hpp)
    int gg_phase4_min;     // VG_GG_PHASE4_MIN     512: ... where the launch has at least this many 256-pixel tiles
    int gg_patch_nr3;      // VG_GG_PATCH_NR3      1: 128 x 64 patch kernel with 3 patch rounds (4 workgroups per CU)
    int patch256_min;      // VG_PATCH256_MIN      256: least number of 256 x 128 tiles for the 8-wave patch kernel
    int patch256x64_min;   // VG_PATCH256X64_MIN   512: same for the 256 x 64 tile
    int splitk_max_tiles;  // VG_SPLITK_MAX_TILES  128: most output tiles for which the gather-GEMM splits K
    int splitk_general;    // VG_SPLITK_GENERAL    0: (1: split K also for 64 x 64-tile launches with sub-pixel phases and / or BatchNorm statistics -- measured: the slab reduces cost more than the main kernels gain)
    int splitk_bigk;       // VG_SPLITK_BIGK       1: few-row / long-K data gradients on 128 x 128 tiles with K slices (0: 64 x 64 tiles)
    int splitk_wgs;        // VG_SPLITK_WGS        1024: workgroups a split-K launch aims at (tiles x splits; at most 64 splits)
    int gg_nmajor;         // VG_GG_NMAJOR         1: XCD-major over n tiles where the weights are the larger operand (2: always)
    int edge;              // VG_EDGE              1: narrow-K direct convolution for the 3-channel image layers
    int wg_reduce_t;       // VG_WG_REDUCE_T       1: streaming transpose-reduce of the weight-gradient slabs
    int wg_target;         // VG_WG_TARGET         512: workgroups the wgrad split-M factor aims for
    int wg_spec;           // VG_WG_SPEC           3: wave-specialised 128 x 256 wgrad kernel (0: one role per wave)
    int wg_dma;            // VG_WG_DMA            1: LDS-DMA staging in wgrad
    int wg_xcd;            // VG_WG_XCD            1: XCD-aware workgroup order in wgrad (2: always)
    int bn_fused_fwd;      // VG_BN_FUSED_FWD      1: BatchNorm finalize folded into the elementwise passes of small layers
    int bn_onepass;        // VG_BN_ONEPASS        0: (1: BatchNorm backward in one launch with a grid-wide exchange, bn_onepass.hip -- measured slower)
};
const VgSwitches& vg_sw();

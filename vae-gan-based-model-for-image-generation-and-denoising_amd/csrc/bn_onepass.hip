// BatchNorm2d(+ReLU/LeakyReLU) backward in ONE launch (bf16): the backward of nn.BatchNorm2d / nn.LeakyReLU / nn.ReLU
// reached through loss.backward() at vaegan_code.py:104, :133 (modules of main_vae.py:24-25, gan_code.py:22-82).
//
// The three-launch form (bn_act.hip: column reduce -> finalize -> apply) reads x and dy twice (5 tensor-sized streams) and
// pays three kernel boundaries around a finalize whose only output is 3 C floats.  Here a workgroup owns a contiguous block
// of ROWS with all channels (fully coalesced 16-byte accesses), keeps its x / dy block IN REGISTERS across a grid-wide
// exchange of the per-channel partial sums, and applies dx = a*dz - b*xhat - c to the registers it still holds: 3 streams,
// one launch.  A workgroup holds 512 threads x 8 vectors x 2 tensors x 16 B = 128 KB in registers, i.e. tensors up to
// 256 CUs x 32 768 elements = 8.4 M elements (16.8 MB in bf16) qualify; larger ones keep the three-launch form.
//
// Grid-wide exchange (cdna_hip_programming.md section 6 Guideline 16, MI355X_MICROARCH.md "Valid forms", first table row):
//   producer  every partial-sum pair is stored by ONE 8-byte agent-scope (sc1, write-through) store; every storing wave
//             drains (s_waitcnt vmcnt(0)); workgroup barrier; ONE lane adds 1 to the arrival counter (agent-scope atomic);
//   consumer  ONE lane polls the counter with relaxed agent-scope (sc1) loads + s_sleep, BOUNDED (a give-up sets the sticky
//             error word sync[2] and the launch finishes with garbage instead of hanging the GPU); that lane then runs ONE
//             agent-scope acquire fence (buffer_inv sc1: drops this CU's stale L1 lines) + s_waitcnt vmcnt(0); workgroup
//             barrier; then all waves read the exchanged bytes with plain 16-byte loads.
// All workgroups are resident by construction: at most one 512-thread workgroup per CU is requested (grid <= CU count), and
// nothing else runs on the stream.  The counters return to zero at the end of every successful launch (the last workgroup
// to LEAVE resets them), so a captured graph replays without a memset node; they are allocated zeroed.
// Sums: f32 per thread and per workgroup (as col_reduce_kernel), double over the workgroups in fixed order -> results are
// deterministic; they differ from the three-launch form only by the association of those double additions.
#include "common.hpp"

namespace {

constexpr int OP_TH = 512;          // threads per workgroup: 8 waves, 2 per SIMD -> 256 registers per thread
constexpr int OP_MAXC = 1024;
constexpr int OP_MAXK = 8;           // 16-byte vectors per thread and tensor (K = 16 compiles to 256 registers + 560 bytes of scratch: not used)
constexpr unsigned OP_SPIN_LIMIT = 1u << 20;

typedef unsigned long long u64;

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack_bf(float a, float b) {
    return (uint32_t)ElemT<VG_BF16>::from_f32(a) | ((uint32_t)ElemT<VG_BF16>::from_f32(b) << 16);
}

// activation backward on the pre-activation z with ONE slope: 1 (no activation), 0 (ReLU), s (LeakyReLU) -- same values as
// common.hpp act_bwd for finite gradients, no branch on the activation kind
__device__ __forceinline__ float dact(float z, float g, float eslope) { return z > 0.f ? g : g * eslope; }

// grid-wide arrival: returns false when the bounded wait gave up.  ok_flag: one int of the kernel's (single, dynamic) LDS
// array -- a static __shared__ beside it would shift the dynamic base off its 16-byte alignment (Guideline 17).
__device__ __forceinline__ bool grid_arrive_and_wait(unsigned* sync, unsigned nwg, volatile int* ok_flag_p) {
    volatile int& ok_flag = *ok_flag_p;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // every storing wave drains its sc1 stores
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        bool ok = true;
        while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nwg) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > OP_SPIN_LIMIT) { ok = false; break; }
        }
        if (!ok) __hip_atomic_store(sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok_flag = ok ? 1 : 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");           // ONE acquire after the poll has matched
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // ... completed before the barrier releases the readers
    }
    __syncthreads();
    return ok_flag != 0;
}

// the last workgroup to leave puts the counters back to zero (every workgroup has passed its wait by then)
__device__ __forceinline__ void grid_depart(unsigned* sync, unsigned nwg) {
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == nwg - 1) {
            __hip_atomic_store(sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Per-channel sums of one group's slab rows: lane (pair, ln) adds the rows ln, ln + PL2, ... (NB 16-byte loads in flight,
// branch-free: rows past the end re-read the last row with weight 0), in double, into dred[ln][C][2]; the caller adds the
// PL2 lanes of a channel in order.  Ends with a workgroup barrier.
template <int NB>
__device__ __forceinline__ void group_sums(const float* base, int wgs, int C, double* dred, bool ok) {
    const int tid = threadIdx.x;
    const int NP2 = C >> 1;                                // channel pairs = 16-byte units of a slab row
    const int PL2 = NP2 >= OP_TH ? 1 : OP_TH / NP2;        // lanes per channel pair
    const int pr = PL2 > 1 ? tid % NP2 : tid;              // (NP2 <= OP_TH: C <= 1024)
    const int ln = PL2 > 1 ? tid / NP2 : 0;
    double a0 = 0.0, b0 = 0.0, a1 = 0.0, b1 = 0.0;
    if (ok) {
        for (int w0 = ln; w0 < wgs; w0 += NB * PL2) {
            f32x4 v[NB];
            float m[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int w = w0 + u * PL2;
                m[u] = w < wgs ? 1.f : 0.f;
                v[u] = *reinterpret_cast<const f32x4*>(base + ((int64_t)min(w, wgs - 1) * C + 2 * pr) * 2);
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {                 // fixed order: w ascending within a lane
                a0 += (double)(v[u][0] * m[u]); b0 += (double)(v[u][1] * m[u]);
                a1 += (double)(v[u][2] * m[u]); b1 += (double)(v[u][3] * m[u]);
            }
        }
    }
    double* d = dred + ((int64_t)ln * C + 2 * pr) * 2;
    d[0] = a0; d[1] = b0; d[2] = a1; d[3] = b1;
    __syncthreads();
}

// x, dy, dx: [rows][C] bf16; coeffs: [groups][4][C] = mean | invstd | scale | shift (what the forward pass published);
// slab: [gridDim.x][C][2] f32 workspace; sync: 3 zero-initialised words (+ padding).
template <int K>
__global__ __launch_bounds__(OP_TH) void bn_bwd_onepass_kernel(
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, uint16_t* __restrict__ dx,
    const float* __restrict__ coeffs, const float* __restrict__ gamma, float* __restrict__ dgamma,
    float* __restrict__ dbeta, int accumulate, float* slab, unsigned* sync, int C, int64_t rows_per_group, int groups,
    int wgs_per_group, double count, float eslope) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int* s_flag = reinterpret_cast<int*>(smem_raw);                                    // 16 bytes: the barrier's verdict
    float* s_co = reinterpret_cast<float*>(smem_raw + 16);                             // [7][C]: mean invstd scale shift a b c
    float* red = s_co + 7 * C;                                                          // [RPP][C][2] f32, later [PL2][C][2] f64
    const int tid = threadIdx.x;
    const int VPR = C >> 3, RPP = OP_TH / VPR;
    const int cv = tid % VPR, tr = tid / VPR;
    const int c8 = cv * 8;
    const int grp = blockIdx.x / wgs_per_group, wi = blockIdx.x - grp * wgs_per_group;
    const int64_t gbase = (int64_t)grp * rows_per_group;
    const int64_t r0 = gbase + (int64_t)wi * (K * RPP);
    const int64_t r1 = min(gbase + rows_per_group, r0 + (int64_t)K * RPP);
    const float* co = coeffs + (int64_t)grp * 4 * C;

    // ---- phase 1: the block of rows into registers (2 K loads in flight per thread) ----
    // addresses = uniform base (SGPRs) + one 32-bit per-thread offset + a uniform step per k: no per-k address registers
    const int nrows = (int)(r1 - r0);                       // rows of this workgroup (uniform)
    const uint32_t toff = (uint32_t)(tr * C + c8);
    const uint16_t* xb = x + r0 * C;
    const uint16_t* gb = dy + r0 * C;
    uint16_t* ob = dx + r0 * C;
    const int kstep = RPP * C;
    // every load is issued unconditionally (a row past the block's end re-reads the block's last row and is masked out of
    // the sums and never stored): 2 K independent loads in flight per thread, no exec-mask branches between them
    u32x4 xv[K], gv[K];
    float vm[K];                                            // 1 for a row of this block, 0 for the clamped re-read
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int row = k * RPP + tr;
        const bool valid = row < nrows;
        vm[k] = valid ? 1.f : 0.f;
        const uint32_t off = valid ? (uint32_t)k * (uint32_t)kstep + toff : (uint32_t)(nrows - 1) * (uint32_t)C + (uint32_t)c8;
        xv[k] = *reinterpret_cast<const u32x4*>(xb + off);
        gv[k] = *reinterpret_cast<const u32x4*>(gb + off);
    }
    for (int i = tid; i < 4 * C; i += OP_TH) s_co[i] = co[i];
    __syncthreads();

    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
    for (int h = 0; h < 4; ++h) {                          // channel pairs: coefficients of two channels live at a time
        const int ca = c8 + 2 * h, cb = ca + 1;
        const float mu0 = s_co[ca], mu1 = s_co[cb], is0 = s_co[C + ca], is1 = s_co[C + cb];
        const float sc0 = s_co[2 * C + ca], sc1_ = s_co[2 * C + cb], sh0 = s_co[3 * C + ca], sh1 = s_co[3 * C + cb];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t xw = xv[k][h], gw = gv[k][h];
            const float v0 = bf_lo(xw), v1 = bf_hi(xw);
            const float dz0 = vm[k] * dact(sc0 * v0 + sh0, bf_lo(gw), eslope);
            const float dz1 = vm[k] * dact(sc1_ * v1 + sh1, bf_hi(gw), eslope);
            s1[2 * h] += dz0; s1[2 * h + 1] += dz1;
            s2[2 * h] += dz0 * ((v0 - mu0) * is0);
            s2[2 * h + 1] += dz1 * ((v1 - mu1) * is1);
            if (K >= 16 && (k & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // bound the live temporaries (128 data registers)
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    {
        float* rr = red + ((int64_t)tr * C + c8) * 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) { rr[2 * j] = s1[j]; rr[2 * j + 1] = s2[j]; }
    }
    __syncthreads();
    float* myrow = slab + (int64_t)blockIdx.x * C * 2;
    for (int c = tid; c < C; c += OP_TH) {
        float a = 0.f, b = 0.f;
        for (int t = 0; t < RPP; ++t) { a += red[((int64_t)t * C + c) * 2]; b += red[((int64_t)t * C + c) * 2 + 1]; }
        const u64 bits = (u64)__float_as_uint(a) | ((u64)__float_as_uint(b) << 32);
        __hip_atomic_store(reinterpret_cast<u64*>(myrow + 2 * c), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // What phase 2 needs of phase 1 is the PACKED block only: without this the compiler keeps the unpacked values, z and xhat of
    // every element alive across the exchange (common subexpressions of the apply pass) and spills.
#pragma unroll
    for (int k = 0; k < K; ++k) asm volatile("" : "+v"(xv[k]), "+v"(gv[k]));

    // ---- grid-wide exchange ----
    const unsigned nwg = gridDim.x;
    const bool ok = grid_arrive_and_wait(sync, nwg, s_flag);

    // ---- phase 2: sums over the workgroups of this group, coefficients ----
    double* dred = reinterpret_cast<double*>(red);         // [PL2][C][2]
    const bool writer = blockIdx.x == 0;                   // (a workgroup of group 0)
    group_sums<(K >= 16 ? 8 : 16)>(slab + (int64_t)grp * wgs_per_group * C * 2, wgs_per_group, C, dred, ok);
    {
        const int PL2 = (C >> 1) >= OP_TH ? 1 : OP_TH / (C >> 1);
        for (int c = tid; c < C; c += OP_TH) {
            double t1 = 0.0, t2 = 0.0;
            for (int l = 0; l < PL2; ++l) { t1 += dred[((int64_t)l * C + c) * 2]; t2 += dred[((int64_t)l * C + c) * 2 + 1]; }
            if (writer) {
                const float fs1 = (float)t1, fs2 = (float)t2;
                if (dgamma) dgamma[c] = accumulate ? dgamma[c] + fs2 : fs2;
                if (dbeta) dbeta[c] = accumulate ? dbeta[c] + fs1 : fs1;
            }
            const float a = (gamma ? gamma[c] : 1.f) * s_co[C + c];                      // gamma * invstd
            s_co[4 * C + c] = a;
            s_co[5 * C + c] = (float)((double)a * t2 / count);
            s_co[6 * C + c] = (float)((double)a * t1 / count);
        }
    }
    __syncthreads();

    // apply, channel pair by channel pair (coefficients of two channels live at a time); the result overwrites the x word
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int ca = c8 + 2 * h, cb = ca + 1;
        const float mu0 = s_co[ca], mu1 = s_co[cb], is0 = s_co[C + ca], is1 = s_co[C + cb];
        const float sc0 = s_co[2 * C + ca], sc1_ = s_co[2 * C + cb], sh0 = s_co[3 * C + ca], sh1 = s_co[3 * C + cb];
        const float a0 = s_co[4 * C + ca], a1 = s_co[4 * C + cb], b0 = s_co[5 * C + ca], b1 = s_co[5 * C + cb];
        const float k0 = s_co[6 * C + ca], k1 = s_co[6 * C + cb];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t xw = xv[k][h], gw = gv[k][h];
            const float v0 = bf_lo(xw), v1 = bf_hi(xw);
            const float o0 = a0 * dact(sc0 * v0 + sh0, bf_lo(gw), eslope) - b0 * ((v0 - mu0) * is0) - k0;
            const float o1 = a1 * dact(sc1_ * v1 + sh1, bf_hi(gw), eslope) - b1 * ((v1 - mu1) * is1) - k1;
            xv[k][h] = pack_bf(o0, o1);
            if (K >= 16 && (k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);                 // keep the next pair's coefficient reads behind this pair's arithmetic
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (k * RPP + tr < nrows) *reinterpret_cast<u32x4*>(ob + (int64_t)k * kstep + toff) = xv[k];

    // dgamma / dbeta of the other groups (a grouped Discriminator pass: real and fake rows), accumulated in group order by
    // the same first workgroup, now that its registers are free
    if (writer) {
        const int PL2 = (C >> 1) >= OP_TH ? 1 : OP_TH / (C >> 1);
        for (int g = 1; g < groups; ++g) {
            __syncthreads();
            group_sums<16>(slab + (int64_t)g * wgs_per_group * C * 2, wgs_per_group, C, dred, ok);
            for (int c = tid; c < C; c += OP_TH) {
                double t1 = 0.0, t2 = 0.0;
                for (int l = 0; l < PL2; ++l) { t1 += dred[((int64_t)l * C + c) * 2]; t2 += dred[((int64_t)l * C + c) * 2 + 1]; }
                if (dgamma) dgamma[c] += (float)t2;
                if (dbeta) dbeta[c] += (float)t1;
            }
        }
    }
    grid_depart(sync, nwg);            // after the LAST read of the slab: the next launch may overwrite it
}

struct OnePlan { int K, wgs_per_group, rpp; };

inline int device_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 1;
    }
    return n;
}

// K = 0: not supported (C not a power of two in [8, 1024], tensor too large for the register files, ...)
inline OnePlan plan_onepass(int64_t rows, int C, int groups, int dtype) {
    OnePlan p{0, 0, 0};
    if (vg_sw().bn_onepass == 0) return p;
    if (dtype != VG_BF16 || C < 8 || C > OP_MAXC || (C & (C - 1)) != 0 || groups < 1 || rows <= 0 || rows % groups != 0) return p;
    const int64_t rpg = rows / groups;
    const int vpr = C / 8, rpp = OP_TH / vpr;
    const int np2 = C / 2, pl2 = np2 >= OP_TH ? 1 : OP_TH / np2;
    const int ncu = device_cus();
    for (int K = 1; K <= OP_MAXK; K *= 2) {
        const int64_t wpg = (rpg + (int64_t)K * rpp - 1) / ((int64_t)K * rpp);
        if (wpg * groups <= ncu && wpg <= 64 * pl2) {     // at most 64 slab rows per lane in the exchange (4 rounds of 16 loads)
            p.K = K; p.wgs_per_group = (int)wpg; p.rpp = rpp;
            return p;
        }
    }
    return p;
}

}  // namespace

extern "C" int vg_bn_backward_onepass_supported(int64_t rows, int C, int groups, int dtype) {
    return plan_onepass(rows, C, groups, dtype).K != 0 ? 1 : 0;
}

extern "C" int64_t vg_bn_backward_onepass_ws_bytes(int64_t rows, int C, int groups, int dtype) {
    const OnePlan p = plan_onepass(rows, C, groups, dtype);
    if (p.K == 0) return 0;
    return (int64_t)p.wgs_per_group * groups * C * 2 * 4;
}

extern "C" int vg_bn_backward_onepass(const void* x, const void* dy, void* dx, const float* coeffs, const float* gamma,
                                      float* dgamma, float* dbeta, int accumulate, float* slab, unsigned* sync,
                                      int64_t rows, int C, int groups, int act, float slope, int dtype, void* stream) {
    VG_CHECK_ARG(x && dy && dx && coeffs && slab && sync && rows > 0, VG_EINVAL);
    VG_CHECK_ARG(vg_aligned16(x) && vg_aligned16(dy) && vg_aligned16(dx) && vg_aligned16(slab), VG_EALIGN);
    const OnePlan p = plan_onepass(rows, C, groups, dtype);
    if (p.K == 0) return VG_ENOSUP;
    const size_t shm = 16 + (size_t)7 * C * 4 + (size_t)OP_TH * 8 * 2 * 4;  // flag + coefficients + [RPP][C][2] f32 ( >= [PL2][C][2] f64 )
    const dim3 grid(p.wgs_per_group * groups), block(OP_TH);
    const double count = (double)(rows / groups);
    VG_CHECK_ARG(act == VG_ACT_NONE || act == VG_ACT_RELU || act == VG_ACT_LRELU, VG_EINVAL);
    const float eslope = act == VG_ACT_NONE ? 1.f : (act == VG_ACT_RELU ? 0.f : slope);
    hipStream_t s = vg_stream(stream);
    const uint16_t* xx = reinterpret_cast<const uint16_t*>(x);
    const uint16_t* dd = reinterpret_cast<const uint16_t*>(dy);
    uint16_t* oo = reinterpret_cast<uint16_t*>(dx);
#define VG_OP_LAUNCH(KK)                                                                                                  \
    vg_launch_timed(4, bn_bwd_onepass_kernel<KK>, grid, block, shm, s, xx, dd, oo, coeffs, gamma, dgamma, dbeta, accumulate, \
                    slab, sync, C, rows / groups, groups, p.wgs_per_group, count, eslope)
    switch (p.K) {
        case 1: VG_OP_LAUNCH(1); break;
        case 2: VG_OP_LAUNCH(2); break;
        case 4: VG_OP_LAUNCH(4); break;
        case 8: VG_OP_LAUNCH(8); break;
        default: return VG_ENOSUP;
    }
#undef VG_OP_LAUNCH
    return VG_LAUNCH_RC();
}

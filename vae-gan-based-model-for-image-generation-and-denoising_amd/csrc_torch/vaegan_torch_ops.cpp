// PyTorch-ROCm custom-op face of libvaegan_hip.so: TORCH_LIBRARY(vaegan, ...) registrations that forward to the SAME
// C entry points (include/vaegan_hip.h) -- a thin shim, no second kernel path.  BASELINE.json north_star: "bound into
// Python via PyTorch-ROCm custom ops"; SURVEY.md section 8(b).  Tensors are device-memory handles: every op checks
// device / dtype / contiguity with TORCH_CHECK (RuntimeError, the reference's error convention), fills the C
// descriptor from an int list in the struct's field order, and launches on at::hip's current stream.
// Schemas mark the written tensors (Tensor(a!)); no op allocates, synchronises or returns a tensor.
#include <torch/library.h>
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <cstring>
#include "vaegan_hip.h"

namespace {

void* cur_stream() { return reinterpret_cast<void*>(c10::hip::getCurrentHIPStream().stream()); }

void check_dev(const at::Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda(), "vaegan::", name, ": tensor must live on the MI355X ('cuda'); there is no CPU path");
    TORCH_CHECK(t.is_contiguous(), "vaegan::", name, ": tensor must be contiguous");
}
const void* cptr(const c10::optional<at::Tensor>& t, const char* name) {
    if (!t.has_value() || !t->defined()) return nullptr;
    check_dev(*t, name);
    return t->data_ptr();
}
void rc_check(int rc, const char* what) {
    TORCH_CHECK(rc == 0, "vaegan::", what, ": rejected by libvaegan_hip (",
                rc == VG_EINVAL ? "VG_EINVAL bad shape/size/flag" : rc == VG_EALIGN ? "VG_EALIGN 16-byte contract" :
                rc == VG_ENOSUP ? "VG_ENOSUP unsupported configuration" : "HIP launch error", ", code ", rc, ")");
}

// geom: the int32 fields of vg_gg_desc from B to nphase in declaration order (y0/x0/ooy/oox are 4 each),
// then stats_capacity, act, mask_act (39 values in all)
int64_t gather_gemm(const at::Tensor& X, const at::Tensor& W, at::Tensor& Y, const c10::optional<at::Tensor>& bias,
                    const c10::optional<at::Tensor>& stats, const c10::optional<at::Tensor>& ws, const at::Tensor& zeros,
                    const c10::optional<at::Tensor>& mask_x, at::IntArrayRef geom, double act_slope, double mask_slope,
                    int64_t dtype) {
    check_dev(X, "gather_gemm"); check_dev(W, "gather_gemm"); check_dev(Y, "gather_gemm"); check_dev(zeros, "gather_gemm");
    TORCH_CHECK(geom.size() == 39, "vaegan::gather_gemm: geom must hold 39 integers, got ", geom.size());
    vg_gg_desc d;
    std::memset(&d, 0, sizeof(d));
    d.X = X.data_ptr(); d.W = W.data_ptr(); d.Y = Y.data_ptr();
    d.bias = static_cast<const float*>(cptr(bias, "gather_gemm"));
    d.stats = static_cast<float*>(const_cast<void*>(cptr(stats, "gather_gemm")));
    d.zeros = zeros.data_ptr();
    d.mask_x = cptr(mask_x, "gather_gemm");
    if (ws.has_value() && ws->defined()) { check_dev(*ws, "gather_gemm"); d.ws = static_cast<float*>(ws->data_ptr()); d.ws_bytes = ws->numel() * ws->element_size(); }
    int k = 0;
    auto nx = [&]() { return static_cast<int32_t>(geom[k++]); };
    d.B = nx(); d.GH = nx(); d.GW = nx(); d.IH = nx(); d.IW = nx(); d.IC = nx();
    d.SY = nx(); d.SX = nx(); d.DY = nx(); d.DX = nx(); d.TH = nx(); d.TW = nx();
    for (int i = 0; i < 4; ++i) d.y0[i] = nx();
    for (int i = 0; i < 4; ++i) d.x0[i] = nx();
    d.N = nx(); d.Kp = nx(); d.OH = nx(); d.OW = nx(); d.OC = nx(); d.OSY = nx(); d.OSX = nx();
    for (int i = 0; i < 4; ++i) d.ooy[i] = nx();
    for (int i = 0; i < 4; ++i) d.oox[i] = nx();
    d.nphase = nx(); d.stats_capacity = nx(); d.act = nx(); d.mask_act = nx();
    d.act_slope = static_cast<float>(act_slope); d.mask_slope = static_cast<float>(mask_slope);
    const int nparts = d.stats ? vg_gather_gemm_nparts(&d, static_cast<int>(dtype)) : 0;
    rc_check(vg_gather_gemm(&d, static_cast<int>(dtype), cur_stream()), "gather_gemm");
    return nparts;
}

// geom: B GH GW PC NP QH QW QC NQ SY SX DY DX TH TW y0 x0 s_np s_cq s_t accumulate (21 values)
void wgrad(const at::Tensor& P, const at::Tensor& Q, at::Tensor& dW, at::Tensor& ws, const at::Tensor& zeros,
           at::IntArrayRef geom, int64_t dtype) {
    check_dev(P, "wgrad"); check_dev(Q, "wgrad"); check_dev(dW, "wgrad"); check_dev(ws, "wgrad"); check_dev(zeros, "wgrad");
    TORCH_CHECK(geom.size() == 21, "vaegan::wgrad: geom must hold 21 integers, got ", geom.size());
    TORCH_CHECK(dW.scalar_type() == at::kFloat && ws.scalar_type() == at::kFloat, "vaegan::wgrad: dW and ws are float32");
    vg_wg_desc d;
    std::memset(&d, 0, sizeof(d));
    d.P = P.data_ptr(); d.Q = Q.data_ptr(); d.dW = static_cast<float*>(dW.data_ptr());
    d.ws = static_cast<float*>(ws.data_ptr()); d.ws_bytes = ws.numel() * 4; d.zeros = zeros.data_ptr();
    int k = 0;
    auto nx = [&]() { return static_cast<int32_t>(geom[k++]); };
    d.B = nx(); d.GH = nx(); d.GW = nx(); d.PC = nx(); d.NP = nx(); d.QH = nx(); d.QW = nx(); d.QC = nx(); d.NQ = nx();
    d.SY = nx(); d.SX = nx(); d.DY = nx(); d.DX = nx(); d.TH = nx(); d.TW = nx(); d.y0 = nx(); d.x0 = nx();
    d.s_np = nx(); d.s_cq = nx(); d.s_t = nx(); d.accumulate = nx();
    rc_check(vg_wgrad(&d, static_cast<int>(dtype), cur_stream()), "wgrad");
}

int64_t wgrad_ws_bytes(at::IntArrayRef geom, int64_t dtype) {
    TORCH_CHECK(geom.size() == 21, "vaegan::wgrad_ws_bytes: geom must hold 21 integers");
    vg_wg_desc d;
    std::memset(&d, 0, sizeof(d));
    int k = 0;
    auto nx = [&]() { return static_cast<int32_t>(geom[k++]); };
    d.B = nx(); d.GH = nx(); d.GW = nx(); d.PC = nx(); d.NP = nx(); d.QH = nx(); d.QW = nx(); d.QC = nx(); d.NQ = nx();
    d.SY = nx(); d.SX = nx(); d.DY = nx(); d.DX = nx(); d.TH = nx(); d.TW = nx(); d.y0 = nx(); d.x0 = nx();
    d.s_np = nx(); d.s_cq = nx(); d.s_t = nx(); d.accumulate = nx();
    d.zeros = reinterpret_cast<const void*>(16);        // only its presence selects the LDS-DMA plan; never dereferenced here
    return vg_wgrad_ws_bytes(&d, static_cast<int>(dtype));
}

void adam_step(at::Tensor& p, const at::Tensor& g, at::Tensor& m, at::Tensor& v, double lr, double beta1, double beta2,
               double eps, double grad_scale, at::Tensor& state, bool prepared) {
    const at::Tensor* all[5] = {&p, &g, &m, &v, &state};
    for (const at::Tensor* t : all) {
        check_dev(*t, "adam_step");
        TORCH_CHECK(t->scalar_type() == at::kFloat, "vaegan::adam_step: float32 buffers only (fp32 master weights)");
    }
    TORCH_CHECK(g.numel() == p.numel() && m.numel() == p.numel() && v.numel() == p.numel() && state.numel() >= 3,
                "vaegan::adam_step: buffer sizes differ");
    if (prepared) {                                     // the iteration's vg_step_prologue has advanced `state` already
        rc_check(vg_adam_apply(static_cast<float*>(p.data_ptr()), static_cast<const float*>(g.data_ptr()),
                               static_cast<float*>(m.data_ptr()), static_cast<float*>(v.data_ptr()), p.numel(), beta1, beta2, eps,
                               static_cast<float>(grad_scale), static_cast<const float*>(state.data_ptr()), cur_stream()),
                 "adam_step");
        return;
    }
    rc_check(vg_adam_step(static_cast<float*>(p.data_ptr()), static_cast<const float*>(g.data_ptr()),
                          static_cast<float*>(m.data_ptr()), static_cast<float*>(v.data_ptr()), p.numel(), lr, beta1,
                          beta2, eps, static_cast<float>(grad_scale), static_cast<float*>(state.data_ptr()), cur_stream()),
             "adam_step");
}

void bn_act_forward(const at::Tensor& x, at::Tensor& y, const c10::optional<at::Tensor>& scale,
                    const c10::optional<at::Tensor>& shift, int64_t rows, int64_t C, int64_t act, double slope,
                    int64_t groups, int64_t gstride, int64_t dtype) {
    check_dev(x, "bn_act_forward"); check_dev(y, "bn_act_forward");
    TORCH_CHECK(x.numel() == rows * C && y.numel() == x.numel(), "vaegan::bn_act_forward: size mismatch");
    rc_check(vg_bn_act_forward(x.data_ptr(), y.data_ptr(), static_cast<const float*>(cptr(scale, "bn_act_forward")),
                               static_cast<const float*>(cptr(shift, "bn_act_forward")), rows, static_cast<int>(C),
                               static_cast<int>(act), static_cast<float>(slope), static_cast<int>(groups), gstride,
                               static_cast<int>(dtype), cur_stream()), "bn_act_forward");
}

void pack_weights_multi(const at::Tensor& table, int64_t n, int64_t total_tiles, int64_t dtype) {
    check_dev(table, "pack_weights_multi");
    rc_check(vg_pack_weights_multi(reinterpret_cast<const vg_pack_desc*>(table.data_ptr()), static_cast<int>(n), total_tiles,
                                   static_cast<int>(dtype), cur_stream()), "pack_weights_multi");
}

int64_t abi_version() { return vg_abi_version(); }

}  // namespace

TORCH_LIBRARY(vaegan, m) {
    m.def("gather_gemm(Tensor X, Tensor W, Tensor(a!) Y, Tensor? bias, Tensor(b!)? stats, Tensor(c!)? ws, Tensor zeros, "
          "Tensor? mask_x, int[] geom, float act_slope, float mask_slope, int dtype) -> int");
    m.def("wgrad(Tensor P, Tensor Q, Tensor(a!) dW, Tensor(b!) ws, Tensor zeros, int[] geom, int dtype) -> ()");
    m.def("wgrad_ws_bytes(int[] geom, int dtype) -> int");
    m.def("adam_step(Tensor(a!) p, Tensor g, Tensor(b!) m, Tensor(c!) v, float lr, float beta1, float beta2, float eps, "
          "float grad_scale, Tensor(d!) state, bool prepared=False) -> ()");
    m.def("bn_act_forward(Tensor x, Tensor(a!) y, Tensor? scale, Tensor? shift, int rows, int C, int act, float slope, "
          "int groups, int gstride, int dtype) -> ()");
    m.def("pack_weights_multi(Tensor table, int n, int total_tiles, int dtype) -> ()");
    m.def("abi_version() -> int");
}

TORCH_LIBRARY_IMPL(vaegan, CUDA, m) {          // the HIP device is PyTorch-ROCm's "CUDA" dispatch key
    m.impl("gather_gemm", &gather_gemm);
    m.impl("wgrad", &wgrad);
    m.impl("adam_step", &adam_step);
    m.impl("bn_act_forward", &bn_act_forward);
    m.impl("pack_weights_multi", &pack_weights_multi);
}

TORCH_LIBRARY_IMPL(vaegan, CompositeExplicitAutograd, m) {   // host-only queries (no tensor argument to dispatch on)
    m.impl("wgrad_ws_bytes", &wgrad_ws_bytes);
    m.impl("abi_version", &abi_version);
}

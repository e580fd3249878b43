"""ctypes binding of libvaegan_hip.so (the C ABI declared in include/vaegan_hip.h).

The shared library is the product; this file only marshals device pointers and sizes.  There is
no CPU fallback: if the library is missing or a call is rejected, a RuntimeError is raised.
torch is imported first so that the library's DT_NEEDED libamdhip64.so.7 resolves to the HIP
runtime torch already loaded (one runtime per process).
"""
import ctypes
import os
from ctypes import (POINTER, Structure, byref, c_char_p, c_double, c_float, c_int, c_int32, c_int64,
                    c_void_p)

import torch  # noqa: F401  (must be loaded before the HIP library, see module docstring)

VG_F32, VG_BF16, VG_FP8 = 0, 1, 2
VG_FP8_WSHIFT = 6
VG_ACT_NONE, VG_ACT_RELU, VG_ACT_LRELU, VG_ACT_TANH = 0, 1, 2, 3
VG_MAX_PHASE = 4
ABI_VERSION = 10

VG_ENOSUP = -3
_ERR = {-1: "VG_EINVAL (bad shape/size/flag)", -2: "VG_EALIGN (16-byte contract violated)",
        -3: "VG_ENOSUP (unsupported configuration)"}

_I4 = c_int32 * VG_MAX_PHASE


class GGDesc(Structure):
    """vg_gg_desc (include/vaegan_hip.h)."""
    _fields_ = [("X", c_void_p), ("W", c_void_p), ("Y", c_void_p), ("bias", c_void_p), ("stats", c_void_p),
                ("B", c_int32), ("GH", c_int32), ("GW", c_int32),
                ("IH", c_int32), ("IW", c_int32), ("IC", c_int32),
                ("SY", c_int32), ("SX", c_int32), ("DY", c_int32), ("DX", c_int32), ("TH", c_int32), ("TW", c_int32),
                ("y0", _I4), ("x0", _I4),
                ("N", c_int32), ("Kp", c_int32),
                ("OH", c_int32), ("OW", c_int32), ("OC", c_int32), ("OSY", c_int32), ("OSX", c_int32),
                ("ooy", _I4), ("oox", _I4),
                ("nphase", c_int32), ("stats_capacity", c_int32), ("ws", c_void_p), ("ws_bytes", c_int64), ("zeros", c_void_p),
                ("act", c_int32), ("act_slope", c_float),
                ("mask_x", c_void_p), ("mask_act", c_int32), ("mask_slope", c_float)]


class WGDesc(Structure):
    """vg_wg_desc."""
    _fields_ = [("P", c_void_p), ("Q", c_void_p), ("dW", c_void_p), ("ws", c_void_p), ("ws_bytes", c_int64),
                ("B", c_int32), ("GH", c_int32), ("GW", c_int32), ("PC", c_int32), ("NP", c_int32),
                ("QH", c_int32), ("QW", c_int32), ("QC", c_int32), ("NQ", c_int32),
                ("SY", c_int32), ("SX", c_int32), ("DY", c_int32), ("DX", c_int32), ("TH", c_int32), ("TW", c_int32),
                ("y0", c_int32), ("x0", c_int32),
                ("s_np", c_int32), ("s_cq", c_int32), ("s_t", c_int32), ("accumulate", c_int32), ("zeros", c_void_p)]


class TNDesc(Structure):
    """vg_tn_desc."""
    _fields_ = [("X", c_void_p), ("Wp", c_void_p), ("Y", c_void_p), ("Y_nchw", c_void_p), ("eps", c_void_p),
                ("rng", c_void_p), ("draw", c_int32), ("sigma", c_float),
                ("B", c_int32), ("IH", c_int32), ("IW", c_int32), ("C", c_int32), ("N", c_int32), ("K", c_int32),
                ("S", c_int32), ("P", c_int32), ("OH", c_int32), ("OW", c_int32), ("OC", c_int32),
                ("Wpitch", c_int32), ("act", c_int32)]


class EWDesc(Structure):
    """vg_ew_desc."""
    _fields_ = [("Wd", c_void_p), ("Nr", c_void_p), ("dW", c_void_p), ("ws", c_void_p), ("ws_bytes", c_int64),
                ("zeros", c_void_p),
                ("B", c_int32), ("WH", c_int32), ("WW", c_int32), ("C", c_int32), ("NH", c_int32), ("NW", c_int32),
                ("N", c_int32), ("K", c_int32), ("S", c_int32), ("P", c_int32), ("s_c", c_int32), ("s_n", c_int32),
                ("accumulate", c_int32)]


class PackDesc(Structure):
    """vg_pack_desc."""
    _fields_ = [("src", c_void_p), ("dst", c_void_p),
                ("nphase", c_int32), ("N", c_int32), ("C", c_int32), ("IC", c_int32), ("TH", c_int32),
                ("TW", c_int32), ("Kp", c_int32),
                ("s_n", c_int32), ("s_c", c_int32), ("KW", c_int32),
                ("kh0", _I4), ("kw0", _I4), ("kh_step", c_int32), ("kw_step", c_int32),
                ("tap_in_n", c_int32), ("KHW", c_int32), ("tile_start", c_int32)]


# name -> (restype, argtypes); every symbol include/vaegan_hip.h declares
_P, _F, _I, _L, _D = c_void_p, c_float, c_int, c_int64, c_double
SIGNATURES = {
    "vg_abi_version": (c_int, []),
    "vg_launch_count": (ctypes.c_uint64, []),
    "vg_reload_switches": (c_int, []),
    "vg_build_info": (c_char_p, []),
    "vg_timing_enable": (c_int, [_I]),
    "vg_timing_collect": (c_int, [_I, POINTER(c_double), POINTER(c_int)]),
    "vg_gather_gemm_nparts": (c_int, [POINTER(GGDesc), _I]),
    "vg_gather_gemm_tile_m": (c_int, [POINTER(GGDesc), _I]),
    "vg_gather_gemm_family": (c_int, [POINTER(GGDesc), _I]),
    "vg_gather_gemm_ws_bytes": (c_int64, [POINTER(GGDesc), _I]),
    "vg_gather_gemm": (c_int, [POINTER(GGDesc), _I, _P]),
    "vg_wgrad_ws_bytes": (c_int64, [POINTER(WGDesc), _I]),
    "vg_wgrad": (c_int, [POINTER(WGDesc), _I, _P]),
    "vg_pack_weights": (c_int, [POINTER(PackDesc), _I, _P]),
    "vg_pack_tile_count": (c_int, [POINTER(PackDesc)]),
    "vg_pack_weights_multi": (c_int, [_P, _I, _L, _I, _P]),
    "vg_bn_finalize": (c_int, [_P, _I, _I, _L, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P]),
    "vg_bn_finalize_grouped": (c_int, [_P, _I, _I, _I, _L, _P, _P, _P, _P, _F, _F, _P, _P]),
    "vg_bn_backward_finalize_grouped": (c_int, [_P, _I, _I, _I, _L, _P, _P, _P, _P, _I, _P, _P]),
    "vg_slab_sums": (c_int, [_P, _I, _I, _P, _P]),
    "vg_bn_finalize_sums": (c_int, [_P, _I, _L, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P]),
    "vg_bn_backward_finalize_sums": (c_int, [_P, _P, _I, _L, _P, _P, _P, _P, _I, _P, _P]),
    "vg_bn_eval_coeffs": (c_int, [_P, _P, _P, _P, _F, _I, _P, _P, _P]),
    "vg_bn_act_forward": (c_int, [_P, _P, _P, _P, _L, _I, _I, _F, _I, _L, _I, _P]),
    "vg_bn_act_forward_fp8": (c_int, [_P, _P, _P, _P, _P, _L, _I, _I, _F, _I, _L, _I, _P]),
    "vg_channel_stats": (c_int, [_P, _L, _I, _P, _I, POINTER(c_int), _I, _P]),
    "vg_bn_act_backward_reduce": (c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _P, _I, POINTER(c_int), _I, _L,
                                          _I, _P]),
    "vg_bn_backward_finalize": (c_int, [_P, _I, _I, _L, _P, _P, _P, _P, _I, _P, _P]),
    "vg_bn_finalize_act_forward_supported": (c_int, [_I, _I, _I, _L, _I]),
    "vg_bn_finalize_act_forward": (c_int, [_P, _P, _P, _I, _I, _I, _L, _P, _P, _P, _P, _F, _F, _P, _I, _F, _I, _P]),
    "vg_bn_backward_finalize_apply": (c_int, [_P, _P, _P, _P, _I, _I, _I, _L, _P, _P, _P, _P, _I, _I, _F, _I, _P]),
    "vg_bn_act_backward_apply": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _I, _L, _L, _I, _P]),
    "vg_act_backward": (c_int, [_P, _P, _P, _L, _I, _F, _I, _P]),
    "vg_bias_grad": (c_int, [_P, _L, _I, _I, _P, _I, _P, _I, _I, _P]),
    "vg_nchw_to_nhwc": (c_int, [_P, _P, _F, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vg_gather_normalize_u8": (c_int, [_P, _L, _P, _I, _I, _I, _I, _P, _P]),
    "vg_noisy_clamp_to_nhwc": (c_int, [_P, _P, _F, _F, _F, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vg_nhwc_to_nchw": (c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "vg_nchw_grad_to_nhwc": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vg_nhwc_tanh_to_nchw_noisy": (c_int, [_P, _P, _P, _F, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vg_nchw_grad_add_to_nhwc": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vg_reparam_forward": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "vg_kl_forward": (c_int, [_P, _P, _I, _I, _I, _F, _P, _I, _P]),
    "vg_reparam_kl_backward": (c_int, [_P, _P, _P, _P, _F, _P, _I, _I, _I, _I, _I, _P]),
    "vg_dot_sigmoid_forward": (c_int, [_P, _P, _P, _I, _I, _I, _P]),
    "vg_dot_sigmoid_backward": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "vg_dot_wgrad": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vg_bce_forward_backward": (c_int, [_P, _F, _I, _F, _P, _I, _P, _P]),
    "vg_bce_pair_forward_backward": (c_int, [_P, _F, _F, _I, _F, _P, _I, _P, _P]),
    "vg_head_backward": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _F, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vg_mean_forward_backward": (c_int, [_P, _F, _I, _F, _P, _I, _P, _P]),
    "vg_clamp": (c_int, [_P, _L, _F, _F, _P]),
    "vg_mse_forward_backward": (c_int, [_P, _P, _L, _F, _P, _P, _P, _I, _P]),
    "vg_ssim": (c_int, [_P, _P, _I, _I, _I, _I, _P, _P, _I, _P]),
    "vg_axpy": (c_int, [_P, _P, _F, _P, _L, _P]),
    "vg_bn_backward_onepass_supported": (c_int, [_L, _I, _I, _I]),
    "vg_bn_backward_onepass_ws_bytes": (c_int64, [_L, _I, _I, _I]),
    "vg_bn_backward_onepass": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _L, _I, _I, _I, _F, _I, _P]),
    "vg_adam_step": (c_int, [_P, _P, _P, _P, _L, _D, _D, _D, _D, _F, _P, _P]),
    "vg_adam_apply": (c_int, [_P, _P, _P, _P, _L, _D, _D, _D, _F, _P, _P]),
    "vg_rng_advance": (c_int, [_P, _P]),
    "vg_step_prologue": (c_int, [_P, _P, _P, _P, _P, _I, _P, _I, _P]),
    "vg_adam_apply2": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "vg_nchw_to_nhwc_pair": (c_int, [_P, _P, _P, _I, _F, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vg_mse_partial": (c_int, [_P, _P, _L, _F, _P, _P, _I, POINTER(c_int), _P]),
    "vg_kl_forward_mse_final": (c_int, [_P, _P, _I, _I, _I, _F, _P, _P, _I, _L, _P, _I, _P]),
    "vg_randn": (c_int, [_P, _L, _P, _I, _P]),
    "vg_nchw_to_nhwc_rng": (c_int, [_P, _P, _I, _F, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vg_nhwc_tanh_to_nchw_noisy_rng": (c_int, [_P, _P, _P, _I, _F, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vg_reparam_forward_rng": (c_int, [_P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "vg_reparam_kl_backward_rng": (c_int, [_P, _P, _P, _I, _P, _F, _P, _I, _I, _I, _I, _I, _P]),
    "vg_memset_zero": (c_int, [_P, _L, _P]),
    "vg_cast_fp8": (c_int, [_P, _P, _L, _I, _P]),
    "vg_edge_wgrad_ws_bytes": (c_int64, [POINTER(EWDesc)]),
    "vg_edge_wgrad": (c_int, [POINTER(EWDesc), _P]),
    "vg_tnconv_supported": (c_int, [POINTER(TNDesc)]),
    "vg_tnconv": (c_int, [POINTER(TNDesc), _P]),
}

LIB_PATH = os.environ.get("VG_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvaegan_hip.so")
_lib = None


def load():
    """Load the HIP library (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C <package>/csrc).  This package has no CPU or PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"libvaegan_hip.so does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    if lib.vg_abi_version() != ABI_VERSION:
        raise RuntimeError("libvaegan_hip.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc < 0:
        raise RuntimeError(f"{what}: rejected by libvaegan_hip: {_ERR.get(rc, rc)}")
    raise RuntimeError(f"{what}: HIP launch failed with hipError_t {rc}")


def ptr(t) -> c_void_p:
    """Device pointer of a tensor (None -> NULL)."""
    return c_void_p(0) if t is None else c_void_p(t.data_ptr())


def stream_ptr() -> c_void_p:
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def i4(vals):
    vals = list(vals) + [0] * (VG_MAX_PHASE - len(vals))
    return _I4(*vals[:VG_MAX_PHASE])

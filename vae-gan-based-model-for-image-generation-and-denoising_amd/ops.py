"""Thin tensor-level wrappers over the C ABI (libvaegan_hip.so).

Each function takes torch CUDA tensors (used only as device-memory handles), fills the
descriptor struct, and launches on torch's current HIP stream.  No computation happens in
Python/ATen here.  All functions raise RuntimeError when the library rejects the arguments.
"""
import os
from ctypes import byref, c_int

import torch

from . import _lib as L
from .geometry import BF16, F32, FP8, EWSpec, GGSpec, PackSpec, TNSpec, WGSpec, esize

TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16, FP8: torch.uint8}      # e4m3 bytes travel as uint8


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("vaegan_amd ops need tensors on the MI355X (cuda) device; there is no CPU path")
        if t is not None and not t.is_contiguous():
            raise RuntimeError("vaegan_amd ops need contiguous tensors")


class _Workspace:
    """Grow-only scratch buffers, one per (device, tag); all users are ordered on one stream.
    A buffer that is outgrown is RETIRED, never freed: captured hipGraphs (trainer.train_step_graphed, the sibling
    trainers) hold raw pointers into the buffers that existed when they were captured, and a later, larger eager
    call (another trainer, a bigger validation batch) must not hand that memory back to the allocator under them.
    Growth is geometric, so the retired buffers sum to less than four times the live one."""

    def __init__(self):
        self.bufs = {}
        self.retired = []

    def get(self, tag: str, nbytes: int, device) -> torch.Tensor:
        key = (tag, str(device))
        buf = self.bufs.get(key)
        if buf is None or buf.numel() * 4 < nbytes:
            n = max(int(nbytes * 1.25) // 4 + 64, 1 << 16)
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError(f"workspace '{tag}' would grow during graph capture; run a warm-up step first")
            if buf is not None:
                self.retired.append(buf)
            buf = torch.empty(n, dtype=torch.float32, device=device)
            self.bufs[key] = buf
        return buf


WS = _Workspace()


def reload_switches() -> None:
    """Re-read the library's optional VG_* kernel-selection switches from the environment (they are read once, when the
    library is loaded: include/vaegan_hip.h vg_reload_switches).  For tests / A-B scripts that flip one in-process."""
    L.check(L.load().vg_reload_switches(), "vg_reload_switches")


class KernelTimer:
    """Live timing of the GEMM-class launches (bench.py's roofline leg): the library brackets each kernel with a
    HIP start/stop event pair on its launch stream (vg_timing_*); this object only adds up the algorithmic
    FLOP / bytes of the same launches.  Off by default."""
    # "bn": every launch of csrc/bn_act.hip -- BatchNorm finalize / normalise + activation / backward reduce + apply,
    # plain activation backward, bias-gradient column sums: HBM-bound passes over the activation tensors
    FAMILIES = {"gather_gemm": 0, "wgrad": 1, "edge": 2, "gather_gemm_fp8": 3, "bn": 4}

    def __init__(self):
        self.acc = {k: dict(flops=0, bytes=0) for k in self.FAMILIES}
        L.load().vg_timing_enable(1)

    def begin(self, family, flops, nbytes):
        self.acc[family]["flops"] += flops
        self.acc[family]["bytes"] += nbytes
        return None

    def end(self, tok):
        pass

    def summary(self):
        """family -> dict(launches, ms, flops, bytes); synchronises the recorded events and stops timing."""
        import ctypes
        out = {}
        lib = L.load()
        for fam, idx in self.FAMILIES.items():
            ms, n = ctypes.c_double(0.0), ctypes.c_int(0)
            L.check(lib.vg_timing_collect(idx, ctypes.byref(ms), ctypes.byref(n)), "vg_timing_collect")
            out[fam] = dict(launches=n.value, ms=ms.value, **self.acc[fam])
        lib.vg_timing_enable(0)
        return out


TIMER = None


def _bn_bytes(passes: int, numel: int, dtype: int) -> None:
    """Algorithmic HBM bytes of a BatchNorm / activation pass for bench.py's roofline_bn: `passes` tensor-sized streams
    (forward: read the raw output, write the activated one = 2; backward: reduce reads x and dy, apply reads x and dy and
    writes dx = 5; the small statistics / coefficient vectors are not counted)."""
    if TIMER is not None:
        TIMER.begin("bn", 0, passes * numel * (4 if dtype == F32 else 2))

# ---- binding ---------------------------------------------------------------------------------------------------------
# The kernels live behind the C ABI (include/vaegan_hip.h).  Two faces reach it from Python:
#   "ctypes"   (default)  _lib.py marshals pointers and descriptor structs;
#   "torchops"            the PyTorch-ROCm custom-op face: torch.ops.vaegan.* registered by libvaegan_torch_ops.so
#                         (csrc_torch/vaegan_torch_ops.cpp, TORCH_LIBRARY + TORCH_LIBRARY_IMPL(CUDA)), a thin shim over
#                         the SAME entry points.  Covers the GEMM-class and optimizer launches (gather_gemm, wgrad,
#                         adam_step, bn_act_forward, pack_weights_multi); host-only queries stay on ctypes.
# Select with VG_BINDING=torchops or ops.set_binding("torchops"); results are bit-identical (tests/test_gpu_binding.py).
BINDING = "ctypes"
_TORCH_OPS = None


def torch_ops():
    """torch.ops.vaegan, loading libvaegan_torch_ops.so on first use (RuntimeError if it was not built)."""
    global _TORCH_OPS
    if _TORCH_OPS is None:
        import os
        path = os.path.join(os.path.dirname(L.LIB_PATH), "libvaegan_torch_ops.so")
        if not os.path.isfile(path):
            raise RuntimeError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        L.load()                                   # libvaegan_hip.so first: the shim links against it
        torch.ops.load_library(path)
        if torch.ops.vaegan.abi_version() != L.ABI_VERSION:
            raise RuntimeError("libvaegan_torch_ops.so was built against another libvaegan_hip ABI; rebuild")
        _TORCH_OPS = torch.ops.vaegan
    return _TORCH_OPS


def set_binding(name: str) -> None:
    global BINDING
    if name not in ("ctypes", "torchops"):
        raise ValueError("binding must be 'ctypes' or 'torchops'")
    if name == "torchops":
        torch_ops()
    BINDING = name


def _gg_geom(d) -> list:
    return ([d.B, d.GH, d.GW, d.IH, d.IW, d.IC, d.SY, d.SX, d.DY, d.DX, d.TH, d.TW] + list(d.y0) + list(d.x0) +
            [d.N, d.Kp, d.OH, d.OW, d.OC, d.OSY, d.OSX] + list(d.ooy) + list(d.oox) +
            [d.nphase, d.stats_capacity, d.act, d.mask_act])


def _wg_geom(d) -> list:
    return [d.B, d.GH, d.GW, d.PC, d.NP, d.QH, d.QW, d.QC, d.NQ, d.SY, d.SX, d.DY, d.DX, d.TH, d.TW, d.y0, d.x0,
            d.s_np, d.s_cq, d.s_t, d.accumulate]


def set_timer(t) -> None:
    global TIMER
    TIMER = t


def empty_act(shape, dtype: int, device) -> torch.Tensor:
    return torch.empty(shape, dtype=TORCH_DT[dtype], device=device)


def zeros_f32(n: int, device) -> torch.Tensor:
    """f32 zeros through hipMemsetAsync (no ATen fill kernel)."""
    t = torch.empty(n, dtype=torch.float32, device=device)
    L.check(L.load().vg_memset_zero(t.data_ptr(), n * 4, L.stream_ptr()), "vg_memset_zero")
    return t


def memset_zero(t: torch.Tensor) -> None:
    _need_cuda(t)
    L.check(L.load().vg_memset_zero(t.data_ptr(), t.numel() * t.element_size(), L.stream_ptr()), "vg_memset_zero")


class NoiseDraw:
    """One of the N(0,1) draws of an iteration, generated inside the kernel that consumes it (vg_*_rng)."""
    __slots__ = ("state", "draw")

    def __init__(self, state: torch.Tensor, draw: int):
        self.state, self.draw = state, draw


class NoiseStream:
    """Device-resident generator for the in-kernel draws (include/vaegan_hip.h "In-kernel N(0,1) noise"):
    int64[2] = {seed, iteration counter}.  advance() is a one-thread kernel, captured with the iteration."""

    def __init__(self, device, seed: int):
        self.seed = int(seed)
        self.state = torch.tensor([self.seed & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=device)
        if not self.state.is_cuda:
            raise RuntimeError("NoiseStream lives on the MI355X ('cuda'); there is no CPU path")

    def advance(self) -> None:
        L.check(L.load().vg_rng_advance(self.state.data_ptr(), L.stream_ptr()), "vg_rng_advance")

    def draw(self, k: int) -> NoiseDraw:
        return NoiseDraw(self.state, k)

    def randn(self, shape, k: int) -> torch.Tensor:
        """Materialise draw k of the current iteration (exactly what the *_rng kernels consume)."""
        out = torch.empty(shape, dtype=torch.float32, device=self.state.device)
        L.check(L.load().vg_randn(out.data_ptr(), out.numel(), self.state.data_ptr(), k, L.stream_ptr()), "vg_randn")
        return out

    def get_state(self):
        return self.state.clone()

    def set_state(self, st: torch.Tensor) -> None:
        self.state.copy_(st.to(self.state.device))


_NOISE = {}


def reset_noise() -> None:
    """Forget the per-device default noise streams: the next draw starts a fresh stream at iteration 0, keyed by
    torch's current device seed (utils.configure_seed calls this)."""
    _NOISE.clear()


def default_noise(device) -> NoiseStream:
    """Per-device stream seeded from torch's device generator seed (utils.configure_seed / torch.cuda.manual_seed
    govern it, as they govern torch.randn_like in the reference); re-seeding torch starts a new stream."""
    key = str(device)
    seed = torch.cuda.initial_seed()
    ns = _NOISE.get(key)
    if ns is None or ns.seed != seed:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the noise stream cannot be (re)created during graph capture; run a warm-up step first")
        ns = NoiseStream(device, seed)
        _NOISE[key] = ns
    return ns


# ---------------------------------------------------------------------------------------------
def pack_weights(pk: PackSpec, w: torch.Tensor, dtype: int, out: torch.Tensor = None) -> torch.Tensor:
    _need_cuda(w, out)
    if w.dtype != torch.float32:
        raise RuntimeError("parameters must be float32 (fp32 master weights)")
    if out is None:
        out = torch.empty(pk.numel(), dtype=TORCH_DT[dtype], device=w.device)
    d = L.PackDesc(src=w.data_ptr(), dst=out.data_ptr(), nphase=pk.nphase, N=pk.N, C=pk.C, IC=pk.IC, TH=pk.TH,
                   TW=pk.TW, Kp=pk.Kp, s_n=pk.s_n, s_c=pk.s_c, KW=pk.KW, kh0=L.i4(pk.kh0), kw0=L.i4(pk.kw0),
                   kh_step=pk.kh_step, kw_step=pk.kw_step, tap_in_n=pk.tap_in_n, KHW=pk.KHW)
    L.check(L.load().vg_pack_weights(byref(d), dtype, L.stream_ptr()), "vg_pack_weights")
    return out


def pack_desc(pk: PackSpec, w: torch.Tensor, out: torch.Tensor) -> L.PackDesc:
    return L.PackDesc(src=w.data_ptr(), dst=out.data_ptr(), nphase=pk.nphase, N=pk.N, C=pk.C, IC=pk.IC, TH=pk.TH,
                      TW=pk.TW, Kp=pk.Kp, s_n=pk.s_n, s_c=pk.s_c, KW=pk.KW, kh0=L.i4(pk.kh0), kw0=L.i4(pk.kw0),
                      kh_step=pk.kh_step, kw_step=pk.kw_step, tap_in_n=pk.tap_in_n, KHW=pk.KHW)


def pack_table(descs, device):
    """Descriptor table in device memory for pack_weights_multi (built once per network).
    Returns (table, total_tiles); fills every descriptor's tile_start (flat one-workgroup-per-tile launch)."""
    import ctypes
    lib = L.load()
    total = 0
    for d in descs:
        d.tile_start = total
        nt = lib.vg_pack_tile_count(byref(d))
        if nt <= 0:
            L.check(nt if nt < 0 else -1, "vg_pack_tile_count")
        total += nt
    arr = (L.PackDesc * len(descs))(*descs)
    raw = bytes(ctypes.string_at(ctypes.addressof(arr), ctypes.sizeof(arr)))
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device), total


def pack_weights_multi(table: torch.Tensor, n: int, total_tiles: int, dtype: int) -> None:
    if BINDING == "torchops":
        torch_ops().pack_weights_multi(table, n, total_tiles, dtype)
        return
    L.check(L.load().vg_pack_weights_multi(table.data_ptr(), n, total_tiles, dtype, L.stream_ptr()),
            "vg_pack_weights_multi")


_ZEROS = {}


def zero_page(device) -> torch.Tensor:
    z = _ZEROS.get(str(device))
    if z is None:
        z = zeros_f32(64, device)                                     # 256 zero bytes
        _ZEROS[str(device)] = z
    return z


def _gg_desc(g: GGSpec, X, Wp, Y, bias, stats, cap) -> L.GGDesc:
    return L.GGDesc(zeros=zero_page(X.device).data_ptr(), X=X.data_ptr(), W=Wp.data_ptr(), Y=Y.data_ptr(),
                    bias=0 if bias is None else bias.data_ptr(), stats=0 if stats is None else stats.data_ptr(),
                    B=g.B, GH=g.GH, GW=g.GW, IH=g.IH, IW=g.IW, IC=g.IC, SY=g.SY, SX=g.SX, DY=g.DY, DX=g.DX,
                    TH=g.TH, TW=g.TW, y0=L.i4(g.y0), x0=L.i4(g.x0), N=g.N, Kp=g.Kp, OH=g.OH, OW=g.OW, OC=g.OC,
                    OSY=g.OSY, OSX=g.OSX, ooy=L.i4(g.ooy), oox=L.i4(g.oox), nphase=g.nphase, stats_capacity=cap)


def gather_gemm(g: GGSpec, X: torch.Tensor, Wp: torch.Tensor, dtype: int, bias: torch.Tensor = None,
                want_stats: bool = False, out: torch.Tensor = None, alg=None, act=None, mask=None):
    """Returns (Y [B,OH,OW,OC], stats slabs or None, nparts).  alg = (flops, bytes) of the layer for the timer."""
    _need_cuda(X, Wp, bias, out)
    if X.dtype != TORCH_DT[dtype] or Wp.dtype != TORCH_DT[dtype]:
        raise RuntimeError(f"gather_gemm: operand dtype {X.dtype}/{Wp.dtype} does not match engine dtype")
    if X.numel() != g.B * g.IH * g.IW * g.IC:
        raise RuntimeError(f"gather_gemm: input has {X.numel()} elements, geometry expects "
                           f"{(g.B, g.IH, g.IW, g.IC)}")
    if Wp.numel() != g.nphase * g.N * g.Kp:
        raise RuntimeError("gather_gemm: packed weight size mismatch")
    lib = L.load()
    Y = out if out is not None else empty_act((g.B, g.OH, g.OW, g.OC), BF16 if dtype == FP8 else dtype, X.device)
    if Y.numel() != g.B * g.OH * g.OW * g.OC:
        raise RuntimeError("gather_gemm: output size mismatch")
    stats, nparts = None, 0
    if want_stats:
        probe = _gg_desc(g, X, Wp, Y, bias, None, 0)
        probe.stats = X.data_ptr()           # the tile choice of the launch below, which emits statistics (query only)
        nparts = lib.vg_gather_gemm_nparts(byref(probe), dtype)
        if nparts < 0:
            L.check(nparts, "vg_gather_gemm_nparts")
        stats = WS.get("stats", nparts * 2 * g.N * 4, X.device)
    d = _gg_desc(g, X, Wp, Y, bias, stats, nparts)
    if act is not None:                      # (code, slope): activation of a BatchNorm-less layer, fused in the epilogue
        d.act, d.act_slope = act
    if mask is not None:                     # (activated output of the layer below, code, slope): its activation backward, fused
        mx, mact, mslope = mask
        if mx.numel() != Y.numel() or mx.dtype != Y.dtype:
            raise RuntimeError("gather_gemm: mask tensor must be shaped and typed like the output")
        d.mask_x, d.mask_act, d.mask_slope = mx.data_ptr(), mact, mslope
    wsb = lib.vg_gather_gemm_ws_bytes(byref(d), dtype)
    ws = None
    if wsb > 0:
        ws = WS.get("splitk", wsb, X.device)
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel() * 4
    tok = None
    if TIMER is not None:
        fam = {0: "gather_gemm", 2: "edge", 3: "gather_gemm_fp8"}[lib.vg_gather_gemm_family(byref(d), dtype)]
        tok = TIMER.begin(fam, *(alg or (g.flops(), 0)))
    if BINDING == "torchops":
        torch_ops().gather_gemm(X, Wp, Y, bias, stats, ws if wsb > 0 else None, zero_page(X.device),
                                mask[0] if mask is not None else None, _gg_geom(d), float(d.act_slope),
                                float(d.mask_slope), dtype)
    else:
        L.check(lib.vg_gather_gemm(byref(d), dtype, L.stream_ptr()), "vg_gather_gemm")
    if tok is not None:
        TIMER.end(tok)
    return Y, stats, nparts


def edge_wgrad(ew: EWSpec, wide: torch.Tensor, narrow: torch.Tensor, dW: torch.Tensor, accumulate: bool, alg=None) -> None:
    """Weight gradient of an edge layer (vg_edge_wgrad): wide [B,WH,WW,C] bf16, narrow [B,NH,NW,8] bf16, dW f32."""
    _need_cuda(wide, narrow, dW)
    if wide.dtype != torch.bfloat16 or narrow.dtype != torch.bfloat16 or dW.dtype != torch.float32:
        raise RuntimeError("edge_wgrad: bf16 operands, float32 gradient")
    if wide.numel() != ew.B * ew.WH * ew.WW * ew.C or narrow.numel() != ew.B * ew.NH * ew.NW * 8:
        raise RuntimeError("edge_wgrad: operand size mismatch")
    lib = L.load()
    d = L.EWDesc(Wd=wide.data_ptr(), Nr=narrow.data_ptr(), dW=dW.data_ptr(), ws=0, ws_bytes=0,
                 zeros=zero_page(wide.device).data_ptr(), B=ew.B, WH=ew.WH, WW=ew.WW, C=ew.C, NH=ew.NH, NW=ew.NW, N=ew.N,
                 K=ew.K, S=ew.S, P=ew.P, s_c=ew.s_c, s_n=ew.s_n, accumulate=1 if accumulate else 0)
    nbytes = lib.vg_edge_wgrad_ws_bytes(byref(d))
    if nbytes < 0:
        L.check(int(nbytes), "vg_edge_wgrad_ws_bytes")
    ws = WS.get("wgrad", nbytes, wide.device)
    d.ws, d.ws_bytes = ws.data_ptr(), ws.numel() * 4
    tok = TIMER.begin("wgrad", *(alg or (ew.flops(), 0))) if TIMER is not None else None
    L.check(lib.vg_edge_wgrad(byref(d), L.stream_ptr()), "vg_edge_wgrad")
    if tok is not None:
        TIMER.end(tok)


def cast_fp8(x: torch.Tensor, shift: int = 0, out: torch.Tensor = None) -> torch.Tensor:
    """bf16 -> e4m3 bytes (uint8 tensor of the same shape), y = fp8(x * 2^shift)."""
    _need_cuda(x, out)
    if x.dtype != torch.bfloat16 or x.numel() % 8:
        raise RuntimeError("cast_fp8: bf16 input with a multiple of 8 elements")
    y = out if out is not None else torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    L.check(L.load().vg_cast_fp8(x.data_ptr(), y.data_ptr(), x.numel(), shift, L.stream_ptr()), "vg_cast_fp8")
    return y


def tnconv(tn: TNSpec, X: torch.Tensor, Wp: torch.Tensor, want_nhwc: bool = True, want_nchw: bool = False,
           act: int = 0, noise=None, sigma: float = 0.0, out_nhwc: torch.Tensor = None, alg=None):
    """Narrow-N transposed convolution (vg_tnconv) -> (Y NHWC bf16 [B,OH,OW,OC] or None, Y NCHW f32 or None).
    noise: None, an NCHW f32 tensor [B,N,OH,OW] or a NoiseDraw; with noise, Y = act(.) + sigma*noise."""
    rng = isinstance(noise, NoiseDraw)
    _need_cuda(X, Wp, None if rng else noise, out_nhwc)
    if X.dtype != torch.bfloat16 or Wp.dtype != torch.bfloat16:
        raise RuntimeError("tnconv: bf16 operands only")
    if X.numel() != tn.B * tn.IH * tn.IW * tn.C or Wp.numel() != tn.K * tn.K * tn.N * tn.Wpitch:
        raise RuntimeError("tnconv: operand size mismatch")
    Y = None
    if want_nhwc:
        Y = out_nhwc if out_nhwc is not None else torch.empty(tn.B, tn.OH, tn.OW, tn.OC, dtype=torch.bfloat16, device=X.device)
        if Y.numel() != tn.B * tn.OH * tn.OW * tn.OC:
            raise RuntimeError("tnconv: output size mismatch")
    Yc = torch.empty(tn.B, tn.N, tn.OH, tn.OW, dtype=torch.float32, device=X.device) if want_nchw else None
    if noise is not None and not rng and noise.numel() != tn.B * tn.N * tn.OH * tn.OW:
        raise RuntimeError("tnconv: noise tensor must be [B,N,OH,OW]")
    d = L.TNDesc(X=X.data_ptr(), Wp=Wp.data_ptr(), Y=0 if Y is None else Y.data_ptr(),
                 Y_nchw=0 if Yc is None else Yc.data_ptr(),
                 eps=noise.data_ptr() if (noise is not None and not rng) else 0,
                 rng=noise.state.data_ptr() if rng else 0, draw=noise.draw if rng else 0, sigma=sigma,
                 B=tn.B, IH=tn.IH, IW=tn.IW, C=tn.C, N=tn.N, K=tn.K, S=tn.S, P=tn.P, OH=tn.OH, OW=tn.OW, OC=tn.OC,
                 Wpitch=tn.Wpitch, act=act)
    tok = TIMER.begin("edge", *(alg or (tn.flops(), 0))) if TIMER is not None else None
    L.check(L.load().vg_tnconv(byref(d), L.stream_ptr()), "vg_tnconv")
    if tok is not None:
        TIMER.end(tok)
    return Y, Yc


def wgrad(wg: WGSpec, P: torch.Tensor, Q: torch.Tensor, dW: torch.Tensor, accumulate: bool, dtype: int,
          alg=None) -> None:
    _need_cuda(P, Q, dW)
    if dW.dtype != torch.float32:
        raise RuntimeError("weight gradients are float32")
    if P.numel() != wg.B * wg.GH * wg.GW * wg.PC or Q.numel() != wg.B * wg.QH * wg.QW * wg.QC:
        raise RuntimeError("wgrad: operand size mismatch")
    lib = L.load()
    d = L.WGDesc(P=P.data_ptr(), Q=Q.data_ptr(), dW=dW.data_ptr(), ws=0, ws_bytes=0, B=wg.B, GH=wg.GH, GW=wg.GW,
                 PC=wg.PC, NP=wg.NP, QH=wg.QH, QW=wg.QW, QC=wg.QC, NQ=wg.NQ, SY=wg.SY, SX=wg.SX, DY=wg.DY,
                 DX=wg.DX, TH=wg.TH, TW=wg.TW, y0=wg.y0, x0=wg.x0, s_np=wg.s_np, s_cq=wg.s_cq, s_t=wg.s_t,
                 accumulate=1 if accumulate else 0, zeros=zero_page(P.device).data_ptr())
    nbytes = lib.vg_wgrad_ws_bytes(byref(d), dtype)
    if nbytes < 0:
        L.check(int(nbytes), "vg_wgrad_ws_bytes")
    # the slab workspace is shared by all launches of one stream; work forked onto another stream gets its own
    ws = WS.get("wgrad", nbytes, P.device)
    d.ws = ws.data_ptr()
    d.ws_bytes = ws.numel() * 4
    tok = TIMER.begin("wgrad", *(alg or (0, 0))) if TIMER is not None else None
    if BINDING == "torchops":
        torch_ops().wgrad(P, Q, dW, ws, zero_page(P.device), _wg_geom(d), dtype)
    else:
        L.check(lib.vg_wgrad(byref(d), dtype, L.stream_ptr()), "vg_wgrad")
    if tok is not None:
        TIMER.end(tok)


# ---------------------------------------------------------------------------------------------
def gather_gemm_tile_m(g: GGSpec, X, Wp, dtype: int) -> int:
    """M edge of the tile the launcher will pick = rows covered by one BatchNorm statistics slab."""
    probe = _gg_desc(g, X, Wp, X, None, None, 0)
    probe.stats = X.data_ptr()               # a launch that emits statistics (never dereferenced by the query): its tile choice
    r = L.load().vg_gather_gemm_tile_m(byref(probe), dtype)
    if r < 0:
        L.check(r, "vg_gather_gemm_tile_m")
    return r


def bn_finalize(stats, nparts, C, count, gamma, beta, running_mean, running_var, momentum, eps, device, groups=1,
                sync=None):
    """-> coeffs tensor [groups][4][C]: mean, invstd, scale, shift.  With groups > 1 the slabs of group g are
    parts [g*nparts/groups, ...) and the running statistics are updated group after group (the order in which
    the reference runs its separate forward passes).  sync: None, or an object with .world and
    .all_reduce_sum(tensor) (ddp.GradReducer) -> statistics over the global batch of all ranks."""
    co = torch.empty(groups, 4, C, dtype=torch.float32, device=device)
    npg = nparts // groups
    if sync is not None:
        # synchronised BatchNorm: per-group f64 sums -> ONE all-reduce -> coefficients from the global sums
        sums = torch.empty(groups, 2, C, dtype=torch.float64, device=device)
        for g in range(groups):
            L.check(L.load().vg_slab_sums(stats.data_ptr() + g * npg * 2 * C * 4, npg, C, sums[g].data_ptr(),
                                          L.stream_ptr()), "vg_slab_sums")
        sync.all_reduce_sum(sums)
        for g in range(groups):
            L.check(L.load().vg_bn_finalize_sums(sums[g].data_ptr(), C, (count // groups) * sync.world,
                                                 L.ptr(gamma), L.ptr(beta), L.ptr(running_mean), L.ptr(running_var),
                                                 momentum, eps, co[g, 0].data_ptr(), co[g, 1].data_ptr(),
                                                 co[g, 2].data_ptr(), co[g, 3].data_ptr(), L.stream_ptr()),
                    "vg_bn_finalize_sums")
        return co
    L.check(L.load().vg_bn_finalize_grouped(stats.data_ptr(), npg, groups, C, count // groups, L.ptr(gamma), L.ptr(beta),
                                            L.ptr(running_mean), L.ptr(running_var), momentum, eps, co.data_ptr(),
                                            L.stream_ptr()), "vg_bn_finalize_grouped")
    return co


def bn_eval_coeffs(gamma, beta, running_mean, running_var, eps):
    C = running_mean.numel()
    co = torch.empty(1, 4, C, dtype=torch.float32, device=running_mean.device)
    L.check(L.load().vg_bn_eval_coeffs(L.ptr(gamma), L.ptr(beta), L.ptr(running_mean), L.ptr(running_var), eps, C,
                                       co[0, 2].data_ptr(), co[0, 3].data_ptr(), L.stream_ptr()), "vg_bn_eval_coeffs")
    return co


def bn_act_forward(x, coeffs, rows, C, act, slope, dtype, out=None, want_fp8=False):
    """coeffs: [groups][4][C] (or None for a pure activation).  want_fp8: -> (y, e4m3 copy of y as uint8)."""
    _need_cuda(x, coeffs)
    y = out if out is not None else torch.empty_like(x)
    groups = coeffs.shape[0] if coeffs is not None else 1
    _bn_bytes(2, x.numel(), dtype)
    if want_fp8:
        y8 = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        L.check(L.load().vg_bn_act_forward_fp8(x.data_ptr(), y.data_ptr(), y8.data_ptr(),
                                               coeffs[0, 2].data_ptr() if coeffs is not None else 0,
                                               coeffs[0, 3].data_ptr() if coeffs is not None else 0, rows, C, act, slope,
                                               groups, 4 * C, dtype, L.stream_ptr()), "vg_bn_act_forward_fp8")
        return y, y8
    if BINDING == "torchops":
        torch_ops().bn_act_forward(x, y, coeffs[0, 2] if coeffs is not None else None,
                                   coeffs[0, 3] if coeffs is not None else None, rows, C, act, float(slope), groups, 4 * C, dtype)
        return y
    sc = coeffs[0, 2].data_ptr() if coeffs is not None else 0
    sh = coeffs[0, 3].data_ptr() if coeffs is not None else 0
    L.check(L.load().vg_bn_act_forward(x.data_ptr(), y.data_ptr(), sc, sh, rows, C, act, slope, groups, 4 * C, dtype,
                                       L.stream_ptr()), "vg_bn_act_forward")
    return y


def bn_finalize_act_forward(x, stats, nparts, C, rows, gamma, beta, running_mean, running_var, momentum, eps, act, slope,
                            dtype, groups=1):
    """Train-mode finalize + normalise + activation in one launch where the statistics slab is small
    (bn_act.hip bn_fin_act_fwd_kernel).  Returns (coeffs [groups][4][C], y), or None when the shape does not qualify
    (the caller then runs bn_finalize + bn_act_forward)."""
    lib = L.load()
    if x.shape[-1] != C or not lib.vg_bn_finalize_act_forward_supported(nparts // groups, groups, C, rows, dtype):
        return None
    _need_cuda(x, stats)
    co = torch.empty(groups, 4, C, dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    _bn_bytes(2, x.numel(), dtype)
    L.check(lib.vg_bn_finalize_act_forward(x.data_ptr(), y.data_ptr(), stats.data_ptr(), nparts // groups, groups, C, rows,
                                           L.ptr(gamma), L.ptr(beta), L.ptr(running_mean), L.ptr(running_var), momentum, eps,
                                           co.data_ptr(), act, slope, dtype, L.stream_ptr()), "vg_bn_finalize_act_forward")
    return co, y


def channel_stats(x, rows, C, dtype):
    n = c_int(0)
    cap = 1024
    stats = WS.get("stats", cap * 2 * C * 4, x.device)
    _bn_bytes(1, x.numel(), dtype)
    L.check(L.load().vg_channel_stats(x.data_ptr(), rows, C, stats.data_ptr(), cap, byref(n), dtype, L.stream_ptr()),
            "vg_channel_stats")
    return stats, n.value


_GRID_SYNC = {}


def grid_sync_words(device) -> torch.Tensor:
    """The zero-initialised counter words of the kernels that synchronise their whole grid (csrc/bn_onepass.hip): int32[16],
    [0] arrivals, [1] departures (both back at zero after every launch), [2] sticky error (a bounded wait gave up)."""
    key = str(device)
    t = _GRID_SYNC.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("grid-sync words would be allocated during graph capture; run a warm-up step first")
        t = torch.zeros(16, dtype=torch.int32, device=device)
        _GRID_SYNC[key] = t
    return t


def grid_sync_error(device) -> bool:
    """True when a grid-wide wait of some earlier launch gave up (results of that launch are invalid).  Synchronises."""
    t = _GRID_SYNC.get(str(device))
    return bool(t is not None and int(t[2].item()) != 0)


def bn_act_backward(x, dy, coeffs, rows, C, count, gamma, act, slope, dgamma, dbeta, accumulate, dtype, sync=None):
    """Full BN(+act) backward: returns dx (gradient w.r.t. the raw conv output).  coeffs: [groups][4][C]."""
    _need_cuda(x, dy, coeffs)
    lib = L.load()
    groups = coeffs.shape[0]
    if sync is None and x.shape[-1] == C and lib.vg_bn_backward_onepass_supported(rows, C, groups, dtype):
        # one launch: the workgroups keep their x / dy rows in registers across a grid-wide exchange of the partial sums
        # (csrc/bn_onepass.hip): 3 tensor-sized streams instead of 5, no finalize launch
        _bn_bytes(3, x.numel(), dtype)
        slab = WS.get("bn1pass", lib.vg_bn_backward_onepass_ws_bytes(rows, C, groups, dtype), x.device)
        dx = torch.empty_like(x)
        L.check(lib.vg_bn_backward_onepass(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), coeffs.data_ptr(), L.ptr(gamma),
                                           L.ptr(dgamma), L.ptr(dbeta), 1 if accumulate else 0, slab.data_ptr(),
                                           grid_sync_words(x.device).data_ptr(), rows, C, groups, act, slope, dtype,
                                           L.stream_ptr()), "vg_bn_backward_onepass")
        return dx
    n = c_int(0)
    cap = 2048
    _bn_bytes(5, x.numel(), dtype)
    partial = WS.get("bnbwd", cap * 2 * C * 4, x.device)
    L.check(lib.vg_bn_act_backward_reduce(x.data_ptr(), dy.data_ptr(), coeffs[0, 2].data_ptr(),
                                          coeffs[0, 3].data_ptr(), coeffs[0, 0].data_ptr(), coeffs[0, 1].data_ptr(),
                                          rows, C, act, slope, partial.data_ptr(), cap, byref(n), groups, 4 * C,
                                          dtype, L.stream_ptr()), "vg_bn_act_backward_reduce")
    if sync is None and x.shape[-1] == C and \
            lib.vg_bn_finalize_act_forward_supported(n.value, groups, C, rows, dtype):
        # small layer: finalize + apply in one launch (bn_act.hip bn_bwd_fin_apply_kernel); n = partial rows PER GROUP
        dx = torch.empty_like(x)
        L.check(lib.vg_bn_backward_finalize_apply(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), partial.data_ptr(),
                                                  n.value, groups, C, rows, L.ptr(gamma), coeffs.data_ptr(),
                                                  L.ptr(dgamma), L.ptr(dbeta), 1 if accumulate else 0, act, slope, dtype,
                                                  L.stream_ptr()), "vg_bn_backward_finalize_apply")
        return dx
    coef = torch.empty(groups, 3, C, dtype=torch.float32, device=x.device)
    if sync is not None:
        lsums = torch.empty(groups, 2, C, dtype=torch.float64, device=x.device)
        for g in range(groups):
            L.check(lib.vg_slab_sums(partial.data_ptr() + g * n.value * 2 * C * 4, n.value, C, lsums[g].data_ptr(),
                                     L.stream_ptr()), "vg_slab_sums")
        gsums = lsums.clone()
        sync.all_reduce_sum(gsums)
        for g in range(groups):
            L.check(lib.vg_bn_backward_finalize_sums(gsums[g].data_ptr(), lsums[g].data_ptr(), C,
                                                     (count // groups) * sync.world, L.ptr(gamma),
                                                     coeffs[g, 1].data_ptr(), L.ptr(dgamma), L.ptr(dbeta),
                                                     1 if (accumulate or g > 0) else 0, coef[g].data_ptr(),
                                                     L.stream_ptr()), "vg_bn_backward_finalize_sums")
    if sync is None:
        L.check(lib.vg_bn_backward_finalize_grouped(partial.data_ptr(), n.value, groups, C, count // groups, L.ptr(gamma),
                                                    coeffs.data_ptr(), L.ptr(dgamma), L.ptr(dbeta),
                                                    1 if accumulate else 0, coef.data_ptr(), L.stream_ptr()),
                "vg_bn_backward_finalize_grouped")
    dx = torch.empty_like(x)
    L.check(lib.vg_bn_act_backward_apply(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), coeffs[0, 2].data_ptr(),
                                         coeffs[0, 3].data_ptr(), coeffs[0, 0].data_ptr(), coeffs[0, 1].data_ptr(),
                                         coef.data_ptr(), rows, C, act, slope, groups, 4 * C, 3 * C, dtype,
                                         L.stream_ptr()), "vg_bn_act_backward_apply")
    return dx


def act_backward(x, dy, act, slope, dtype):
    dx = torch.empty_like(x)
    _bn_bytes(3, x.numel(), dtype)
    L.check(L.load().vg_act_backward(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), act, slope, dtype,
                                     L.stream_ptr()), "vg_act_backward")
    return dx


def bias_grad(dy, rows, C, NC, dbias, accumulate, dtype):
    cap = 1024
    _bn_bytes(1, rows * C, dtype)
    ws = WS.get("biasgrad", cap * 2 * C * 4, dy.device)
    L.check(L.load().vg_bias_grad(dy.data_ptr(), rows, C, NC, dbias.data_ptr(), 1 if accumulate else 0,
                                  ws.data_ptr(), cap, dtype, L.stream_ptr()), "vg_bias_grad")


# ---------------------------------------------------------------------------------------------
def nchw_to_nhwc(x, CP, dtype, eps=None, sigma=0.0, out=None):
    """eps: None, an NCHW f32 noise tensor, or a NoiseDraw (generated inside the kernel)."""
    _need_cuda(x, None if isinstance(eps, NoiseDraw) else eps, out)
    B, C, H, W = x.shape
    y = out if out is not None else empty_act((B, H, W, CP), dtype, x.device)
    if isinstance(eps, NoiseDraw):
        L.check(L.load().vg_nchw_to_nhwc_rng(x.data_ptr(), eps.state.data_ptr(), eps.draw, sigma, y.data_ptr(), B, C, H,
                                             W, CP, dtype, L.stream_ptr()), "vg_nchw_to_nhwc_rng")
        return y
    L.check(L.load().vg_nchw_to_nhwc(x.data_ptr(), L.ptr(eps), sigma, y.data_ptr(), B, C, H, W, CP, dtype,
                                     L.stream_ptr()), "vg_nchw_to_nhwc")
    return y


def nchw_to_nhwc_pair(x, CP, dtype, eps, sigma, out_noisy):
    """(x as NHWC, x + sigma * eps as NHWC written into out_noisy) in ONE pass over x (vg_nchw_to_nhwc_pair), or None where
    the four-pixel kernel does not apply.  eps: NCHW f32 noise tensor or a NoiseDraw."""
    rng = isinstance(eps, NoiseDraw)
    _need_cuda(x, None if rng else eps, out_noisy)
    B, C, H, W = x.shape
    if dtype != 1 or CP != 8 or C > 4 or (H * W) % 4 != 0:
        return None
    plain = empty_act((B, H, W, CP), dtype, x.device)
    rc = L.load().vg_nchw_to_nhwc_pair(x.data_ptr(), None if rng else eps.data_ptr(), eps.state.data_ptr() if rng else None,
                                       eps.draw if rng else 0, sigma, out_noisy.data_ptr(), plain.data_ptr(), B, C, H, W, CP,
                                       dtype, L.stream_ptr())
    if rc == L.VG_ENOSUP:
        return None
    L.check(rc, "vg_nchw_to_nhwc_pair")
    return plain, out_noisy


def gather_normalize_u8(images_u8, idx, out=None):
    """images_u8 [N,H,W,C] uint8 on the device, idx [B] int64 on the device -> [B,C,H,W] f32 in [-1,1]
    (ToTensor + Normalize(0.5,0.5), dataset_code.py:147-150)."""
    _need_cuda(images_u8, idx, out)
    if images_u8.dtype != torch.uint8 or idx.dtype != torch.int64 or images_u8.dim() != 4:
        raise RuntimeError("gather_normalize_u8: images must be uint8 [N,H,W,C], idx int64 [B]")
    N, H, W, C = images_u8.shape
    B = idx.numel()
    y = out if out is not None else torch.empty(B, C, H, W, dtype=torch.float32, device=images_u8.device)
    L.check(L.load().vg_gather_normalize_u8(images_u8.data_ptr(), N, idx.data_ptr(), B, C, H, W, y.data_ptr(),
                                            L.stream_ptr()), "vg_gather_normalize_u8")
    return y


def noisy_clamp_to_nhwc(x, eps, sigma, CP, dtype, lo=-1.0, hi=1.0):
    """-> (noisy NHWC engine tensor, noisy NCHW f32)."""
    _need_cuda(x, eps)
    B, C, H, W = x.shape
    y = empty_act((B, H, W, CP), dtype, x.device)
    yn = torch.empty_like(x)
    L.check(L.load().vg_noisy_clamp_to_nhwc(x.data_ptr(), eps.data_ptr(), sigma, lo, hi, y.data_ptr(), yn.data_ptr(),
                                            B, C, H, W, CP, dtype, L.stream_ptr()), "vg_noisy_clamp_to_nhwc")
    return y, yn


def ssim(a, b):
    """Mean SSIM of two NCHW f32 batches in [-1,1] -> device scalar tensor [1]."""
    _need_cuda(a, b)
    B, C, H, W = a.shape
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    ws = WS.get("mse", 1024 * 4, a.device)
    L.check(L.load().vg_ssim(a.data_ptr(), b.data_ptr(), B, C, H, W, out.data_ptr(), ws.data_ptr(), 1024,
                             L.stream_ptr()), "vg_ssim")
    return out


def nhwc_to_nchw(x, C, dtype, apply_tanh=False):
    _need_cuda(x)
    B, H, W, CP = x.shape
    y = torch.empty(B, C, H, W, dtype=torch.float32, device=x.device)
    L.check(L.load().vg_nhwc_to_nchw(x.data_ptr(), y.data_ptr(), B, C, H, W, CP, 1 if apply_tanh else 0, dtype,
                                     L.stream_ptr()), "vg_nhwc_to_nchw")
    return y


def nhwc_tanh_to_nchw_noisy(x, C, eps, sigma, out_noisy, dtype):
    """x NHWC (pre-tanh) -> (tanh(x) as NCHW f32, tanh(x) + sigma*eps written into out_noisy [B,H,W,CP])."""
    rng = isinstance(eps, NoiseDraw)
    _need_cuda(x, None if rng else eps, out_noisy)
    B, H, W, CP = x.shape
    if out_noisy.numel() != x.numel() or (not rng and eps.numel() != B * C * H * W):
        raise RuntimeError("nhwc_tanh_to_nchw_noisy: shape mismatch")
    y = torch.empty(B, C, H, W, dtype=torch.float32, device=x.device)
    if rng:
        L.check(L.load().vg_nhwc_tanh_to_nchw_noisy_rng(x.data_ptr(), y.data_ptr(), eps.state.data_ptr(), eps.draw, sigma,
                                                        out_noisy.data_ptr(), B, C, H, W, CP, dtype, L.stream_ptr()),
                "vg_nhwc_tanh_to_nchw_noisy_rng")
        return y
    L.check(L.load().vg_nhwc_tanh_to_nchw_noisy(x.data_ptr(), y.data_ptr(), eps.data_ptr(), sigma, out_noisy.data_ptr(),
                                                B, C, H, W, CP, dtype, L.stream_ptr()), "vg_nhwc_tanh_to_nchw_noisy")
    return y


def nchw_grad_add_to_nhwc(dy, add_nhwc, tanh_out, CP, dtype):
    """dx NHWC = (dy NCHW f32 + add_nhwc) * (1 - tanh_out^2): two gradient branches of the reconstruction in one pass."""
    _need_cuda(dy, add_nhwc, tanh_out)
    B, C, H, W = dy.shape
    if add_nhwc.numel() != B * H * W * CP:
        raise RuntimeError("nchw_grad_add_to_nhwc: add_nhwc must be [B,H,W,CP]")
    dx = empty_act((B, H, W, CP), dtype, dy.device)
    L.check(L.load().vg_nchw_grad_add_to_nhwc(dy.data_ptr(), add_nhwc.data_ptr(), L.ptr(tanh_out), dx.data_ptr(), B, C, H, W,
                                              CP, dtype, L.stream_ptr()), "vg_nchw_grad_add_to_nhwc")
    return dx


def nchw_grad_to_nhwc(dy, tanh_out, CP, dtype):
    _need_cuda(dy, tanh_out)
    B, C, H, W = dy.shape
    dx = empty_act((B, H, W, CP), dtype, dy.device)
    L.check(L.load().vg_nchw_grad_to_nhwc(dy.data_ptr(), L.ptr(tanh_out), dx.data_ptr(), B, C, H, W, CP, dtype,
                                          L.stream_ptr()), "vg_nchw_grad_to_nhwc")
    return dx


def reparam_forward(mulv, eps, L_dim, ZP, dtype):
    rng = isinstance(eps, NoiseDraw)
    _need_cuda(mulv, None if rng else eps)
    B, MP = mulv.shape[0], mulv.shape[-1]
    z = empty_act((B, 1, 1, ZP), dtype, mulv.device)
    lvc = torch.empty(B, L_dim, dtype=torch.float32, device=mulv.device)
    if rng:
        L.check(L.load().vg_reparam_forward_rng(mulv.data_ptr(), eps.state.data_ptr(), eps.draw, z.data_ptr(),
                                                lvc.data_ptr(), B, L_dim, MP, ZP, dtype, L.stream_ptr()),
                "vg_reparam_forward_rng")
        return z, lvc
    L.check(L.load().vg_reparam_forward(mulv.data_ptr(), eps.data_ptr(), z.data_ptr(), lvc.data_ptr(), B, L_dim, MP,
                                        ZP, dtype, L.stream_ptr()), "vg_reparam_forward")
    return z, lvc


def kl_forward(mulv, lvc, L_dim, divisor, dtype, out=None, mse=None):
    """mse = (workspace, nparts, n, loss slot) from mse_forward_backward(..., defer_final=True): the MSE's final sum is
    formed by this launch (vg_kl_forward_mse_final), same arithmetic as its own finalize launch."""
    B, MP = mulv.shape[0], mulv.shape[-1]
    out = out if out is not None else torch.empty(1, dtype=torch.float32, device=mulv.device)
    if mse is not None:
        ws, nparts, n, slot = mse
        L.check(L.load().vg_kl_forward_mse_final(mulv.data_ptr(), lvc.data_ptr(), B, L_dim, MP, divisor, out.data_ptr(),
                                                 ws.data_ptr(), nparts, n, slot.data_ptr(), dtype, L.stream_ptr()),
                "vg_kl_forward_mse_final")
        return out
    L.check(L.load().vg_kl_forward(mulv.data_ptr(), lvc.data_ptr(), B, L_dim, MP, divisor, out.data_ptr(), dtype,
                                   L.stream_ptr()), "vg_kl_forward")
    return out


def reparam_kl_backward(mulv, lvc, eps, dz, kl_scale, L_dim, dtype):
    B, MP = mulv.shape[0], mulv.shape[-1]
    ZP = dz.shape[-1]
    dmulv = torch.empty_like(mulv)
    if isinstance(eps, NoiseDraw):
        L.check(L.load().vg_reparam_kl_backward_rng(mulv.data_ptr(), lvc.data_ptr(), eps.state.data_ptr(), eps.draw,
                                                    dz.data_ptr(), kl_scale, dmulv.data_ptr(), B, L_dim, MP, ZP, dtype,
                                                    L.stream_ptr()), "vg_reparam_kl_backward_rng")
        return dmulv
    L.check(L.load().vg_reparam_kl_backward(mulv.data_ptr(), lvc.data_ptr(), eps.data_ptr(), dz.data_ptr(), kl_scale,
                                            dmulv.data_ptr(), B, L_dim, MP, ZP, dtype, L.stream_ptr()),
            "vg_reparam_kl_backward")
    return dmulv


def dot_sigmoid_forward(x, w, B, K, dtype):
    p = torch.empty(B, dtype=torch.float32, device=x.device)
    L.check(L.load().vg_dot_sigmoid_forward(x.data_ptr(), w.data_ptr(), p.data_ptr(), B, K, dtype, L.stream_ptr()),
            "vg_dot_sigmoid_forward")
    return p


def dot_sigmoid_backward(p, dp, w, B, K, dtype, need_dx, like):
    dlogit = torch.empty(B, dtype=torch.float32, device=p.device)
    dx = torch.empty_like(like) if need_dx else None
    L.check(L.load().vg_dot_sigmoid_backward(p.data_ptr(), dp.data_ptr(), w.data_ptr(), L.ptr(dx), dlogit.data_ptr(),
                                             B, K, dtype, L.stream_ptr()), "vg_dot_sigmoid_backward")
    return dx, dlogit


def dot_wgrad(x, dlogit, dw, B, K, C, HW, accumulate, dtype):
    L.check(L.load().vg_dot_wgrad(x.data_ptr(), dlogit.data_ptr(), dw.data_ptr(), B, K, C, HW,
                                  1 if accumulate else 0, dtype, L.stream_ptr()), "vg_dot_wgrad")


HEAD_BWD_MAXROWS = 4096


def head_backward(p, x, w, B, groups, target0, target1, gscale, loss, accumulate_loss, dw, accumulate_dw, K, C, HW, dtype,
                  need_dx):
    """BCE(sigmoid(head)) backward in one launch (vg_head_backward): p [groups*B] probabilities, x the head's input
    [groups*B, K] (NHWC-flattened), w its packed weight [K]; loss[0] (+)= the groups' BCE means; returns dx (or None);
    dw (f32, reference layout, or None) (+)= the weight gradient.  Bit-identical to bce[_pair]_forward_backward +
    dot_sigmoid_backward + dot_wgrad."""
    _need_cuda(p, x, w, loss, dw)
    dx = torch.empty_like(x) if need_dx else None
    L.check(L.load().vg_head_backward(p.data_ptr(), x.data_ptr(), w.data_ptr(), L.ptr(dx), L.ptr(dw), 0, B, groups,
                                      target0, target1, gscale, loss.data_ptr(), 1 if accumulate_loss else 0,
                                      1 if accumulate_dw else 0, K, C, HW, dtype, L.stream_ptr()), "vg_head_backward")
    return dx


def bce_forward_backward(p, target, gscale, loss, accumulate, want_grad, out=None):
    _need_cuda(p, loss, out)
    B = p.numel()
    dp = out if out is not None else (torch.empty_like(p) if want_grad else None)
    L.check(L.load().vg_bce_forward_backward(p.data_ptr(), target, B, gscale, loss.data_ptr(),
                                             1 if accumulate else 0, L.ptr(dp), L.stream_ptr()),
            "vg_bce_forward_backward")
    return dp


def bce_pair_forward_backward(p_both, target0, target1, gscale, loss, out):
    """loss[0] = BCE(p_both[:B], target0) + BCE(p_both[B:], target1); out [2B] <- the gradients.  One launch."""
    _need_cuda(p_both, loss, out)
    B = p_both.numel() // 2
    L.check(L.load().vg_bce_pair_forward_backward(p_both.data_ptr(), target0, target1, B, gscale, loss.data_ptr(), 0,
                                                  out.data_ptr(), L.stream_ptr()), "vg_bce_pair_forward_backward")
    return out


def mean_forward_backward(p, sign, gscale, loss, accumulate, want_grad, out=None):
    """loss (+)= sign*mean(p); returns dp = sign*gscale/B (WGAN losses, gan_code.py:306-315, :328)."""
    _need_cuda(p, loss, out)
    dp = out if out is not None else (torch.empty_like(p) if want_grad else None)
    L.check(L.load().vg_mean_forward_backward(p.data_ptr(), sign, p.numel(), gscale, loss.data_ptr(),
                                              1 if accumulate else 0, L.ptr(dp), L.stream_ptr()),
            "vg_mean_forward_backward")
    return dp


def clamp_(flat, lo, hi):
    """In-place clamp of a flat f32 buffer (WGAN weight clipping, gan_code.py:320-321)."""
    _need_cuda(flat)
    L.check(L.load().vg_clamp(flat.data_ptr(), flat.numel(), lo, hi, L.stream_ptr()), "vg_clamp")


def mse_forward_backward(a, b, gscale, loss, want_grad, defer_final=False):
    """defer_final=True: only the partial sums (+ gradient) are launched; returns (da, (ws, nparts, n, loss)) for
    kl_forward(..., mse=...) to finish."""
    _need_cuda(a, b, loss)
    n = a.numel()
    da = torch.empty_like(a) if want_grad else None
    ws = WS.get("mse", 1024 * 4, a.device)
    if defer_final:
        npo = c_int(0)
        L.check(L.load().vg_mse_partial(a.data_ptr(), b.data_ptr(), n, gscale, L.ptr(da), ws.data_ptr(), 1024, byref(npo),
                                        L.stream_ptr()), "vg_mse_partial")
        return da, (ws, npo.value, n, loss)
    L.check(L.load().vg_mse_forward_backward(a.data_ptr(), b.data_ptr(), n, gscale, loss.data_ptr(), L.ptr(da),
                                             ws.data_ptr(), 1024, L.stream_ptr()), "vg_mse_forward_backward")
    return da


def axpy(a, b, alpha, out=None):
    out = out if out is not None else torch.empty_like(a)
    L.check(L.load().vg_axpy(a.data_ptr(), b.data_ptr(), alpha, out.data_ptr(), a.numel(), L.stream_ptr()), "vg_axpy")
    return out


def step_prologue(noise, optimizers, zero=None) -> None:
    """Top of an iteration, one single-thread launch: noise.advance() (noise may be None), the step counters / bias
    corrections of `optimizers` (optim.Adam, at most 4) -- their step(prepared=True) then launches the update alone --
    and, when given, the iteration's loss slots `zero` (f32, <= 64 elements) set to 0 (instead of a memset node)."""
    import ctypes
    n = len(optimizers)
    states = (ctypes.c_void_p * max(n, 1))(*[o.state_dev.data_ptr() for o in optimizers])
    lr = (ctypes.c_double * max(n, 1))(*[o.lr for o in optimizers])
    b1 = (ctypes.c_double * max(n, 1))(*[o.betas[0] for o in optimizers])
    b2 = (ctypes.c_double * max(n, 1))(*[o.betas[1] for o in optimizers])
    L.check(L.load().vg_step_prologue(noise.state.data_ptr() if noise is not None else None, states, lr, b1, b2, n,
                                      L.ptr(zero), 0 if zero is None else zero.numel(), L.stream_ptr()), "vg_step_prologue")


def adam_apply2(opt_a, opt_b) -> None:
    """The prepared update of TWO optim.Adam instances in one launch (vg_adam_apply2); bit-identical to two adam_step(...,
    prepared=True) launches."""
    import ctypes
    oo = (opt_a, opt_b)
    arr = lambda f: (ctypes.c_void_p * 2)(*[f(o) for o in oo])       # noqa: E731
    n = (ctypes.c_int64 * 2)(*[o.flat_p.numel() for o in oo])
    b1 = (ctypes.c_double * 2)(*[o.betas[0] for o in oo])
    b2 = (ctypes.c_double * 2)(*[o.betas[1] for o in oo])
    eps = (ctypes.c_double * 2)(*[o.eps for o in oo])
    gs = (ctypes.c_float * 2)(*[o.grad_scale for o in oo])
    L.check(L.load().vg_adam_apply2(arr(lambda o: o.flat_p.data_ptr()), arr(lambda o: o.flat_g.data_ptr()),
                                    arr(lambda o: o.exp_avg.data_ptr()), arr(lambda o: o.exp_avg_sq.data_ptr()), n, b1, b2,
                                    eps, gs, arr(lambda o: o.state_dev.data_ptr()), L.stream_ptr()), "vg_adam_apply2")


def launch_count() -> int:
    """Kernel launches libvaegan_hip.so has issued so far in this process (vg_launch_count)."""
    return int(L.load().vg_launch_count())


def adam_step(p, g, m, v, lr, beta1, beta2, eps, grad_scale, state, prepared: bool = False):
    """prepared=True: `state` (step count, bias corrections) was advanced by this iteration's step_prologue; only the
    update kernel is launched (vg_adam_apply)."""
    _need_cuda(p, g, m, v, state)
    if lr < 0:
        raise ValueError(f"Invalid learning rate: {lr}")
    if BINDING == "torchops":
        torch_ops().adam_step(p, g, m, v, lr, beta1, beta2, eps, grad_scale, state, bool(prepared))
        return
    if prepared:
        L.check(L.load().vg_adam_apply(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), beta1, beta2, eps,
                                       grad_scale, state.data_ptr(), L.stream_ptr()), "vg_adam_apply")
        return
    L.check(L.load().vg_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1,
                                  beta2, eps, grad_scale, state.data_ptr(), L.stream_ptr()), "vg_adam_step")


import os as _os
if _os.environ.get("VG_BINDING"):
    set_binding(_os.environ["VG_BINDING"])

"""Host-side geometry: turns a layer (Conv2d / ConvTranspose2d / Linear, forward, data-gradient
or weight-gradient) into the integer descriptors of the gather-GEMM, weight-gradient and
operand-pack kernels (vg_gg_desc / vg_wg_desc / vg_pack_desc in include/vaegan_hip.h).

Pure Python, no torch tensors, no GPU: unit-tested on CPU against torch's own convolution
semantics by an index-level emulator in tests/test_geometry_cpu.py.

Layer conventions follow the reference: nn.Conv2d weight [Cout][Cin][k][k] (main_vae.py:23,
gan_code.py:61-84), nn.ConvTranspose2d weight [Cin][Cout][k][k] (gan_code.py:21-49),
nn.Linear weight [N][C*H*W] applied to the NCHW-flattened feature map (main_vae.py:47-56).

Transposed convolutions (and the data gradient of strided convolutions) use the sub-pixel
decomposition: for stride 2 every output parity class (py, px) is a dense (k/2 x k/2)-tap
convolution of the small tensor, so no zero-inserted MACs are issued.
"""
from dataclasses import dataclass
from typing import List, Tuple

F32, BF16 = 0, 1
FP8 = 2          # e4m3 operands of the fp8 forward GEMMs (storage dtype of those engines stays BF16)


def esize(dtype: int) -> int:
    return {F32: 4, BF16: 2, FP8: 1}[dtype]


def per16(dtype: int) -> int:
    return 16 // esize(dtype)


def padc(c: int, dtype: int) -> int:
    """Channel count padded so that one pixel is a whole number of 16-byte units."""
    q = per16(dtype)
    return (c + q - 1) // q * q


def kpad(k: int, dtype: int) -> int:
    """K padded to the 64-byte chunk the GEMM main loop consumes."""
    q = 64 // esize(dtype)
    return (k + q - 1) // q * q


def conv_out(h: int, k: int, s: int, p: int) -> int:
    if h + 2 * p < k:
        # same message the reference relies on for wrong image sizes
        raise RuntimeError(f"Calculated padded input size per channel: ({h + 2 * p} x {h + 2 * p}). "
                           f"Kernel size: ({k} x {k}). Kernel size can't be greater than actual input size")
    return (h + 2 * p - k) // s + 1


def convT_out(h: int, k: int, s: int, p: int) -> int:
    return (h - 1) * s - 2 * p + k


@dataclass
class GGSpec:
    B: int; GH: int; GW: int
    IH: int; IW: int; IC: int
    SY: int; SX: int; DY: int; DX: int; TH: int; TW: int
    y0: List[int]; x0: List[int]
    N: int; Kp: int
    OH: int; OW: int; OC: int; OSY: int; OSX: int
    ooy: List[int]; oox: List[int]
    nphase: int

    @property
    def M(self) -> int:
        return self.B * self.GH * self.GW

    def flops(self) -> int:
        """Useful MACs*2 actually issued (padding channels excluded is the caller's business)."""
        return 2 * self.nphase * self.M * self.N * self.TH * self.TW * self.IC


@dataclass
class PackSpec:
    nphase: int; N: int; C: int; IC: int; TH: int; TW: int; Kp: int
    s_n: int; s_c: int; KW: int
    kh0: List[int]; kw0: List[int]; kh_step: int; kw_step: int
    tap_in_n: int = 0
    KHW: int = 0

    def numel(self) -> int:
        return self.nphase * self.N * self.Kp


@dataclass
class WGSpec:
    B: int; GH: int; GW: int; PC: int; NP: int
    QH: int; QW: int; QC: int; NQ: int
    SY: int; SX: int; DY: int; DX: int; TH: int; TW: int; y0: int; x0: int
    s_np: int; s_cq: int; s_t: int


# ------------------------------------------------------------------------------------------------
# Dense (direct) form:  out[gy, gx] = sum_{a,c} in[gy*s + a - p, gx*s + c - p] * w[., ., a, c]
# used by Conv2d forward and by the data gradient of ConvTranspose2d.
# ------------------------------------------------------------------------------------------------
def _direct(B, H, W, C, N, k, s, p, dtype, s_n, s_c) -> Tuple[GGSpec, PackSpec]:
    OH, OW = conv_out(H, k, s, p), conv_out(W, k, s, p)
    IC = padc(C, dtype)
    Kp = kpad(k * k * IC, dtype)
    gg = GGSpec(B=B, GH=OH, GW=OW, IH=H, IW=W, IC=IC, SY=s, SX=s, DY=1, DX=1, TH=k, TW=k,
                y0=[-p], x0=[-p], N=N, Kp=Kp, OH=OH, OW=OW, OC=padc(N, dtype), OSY=1, OSX=1,
                ooy=[0], oox=[0], nphase=1)
    pk = PackSpec(nphase=1, N=N, C=C, IC=IC, TH=k, TW=k, Kp=Kp, s_n=s_n, s_c=s_c, KW=k,
                  kh0=[0], kw0=[0], kh_step=1, kw_step=1, KHW=k * k)
    return gg, pk


# ------------------------------------------------------------------------------------------------
# Transposed form: big[y, x] = sum small[iy, ix] * w[., ., kh, kw]  with  y = s*iy - p + kh.
# used by ConvTranspose2d forward and by the data gradient of Conv2d.
# (h, w) = small tensor, (BH, BW) = big tensor.
# ------------------------------------------------------------------------------------------------
def _transposed(B, h, w, C, BH, BW, N, k, s, p, dtype, s_n, s_c) -> Tuple[GGSpec, PackSpec]:
    IC = padc(C, dtype)
    if s == 1:
        Kp = kpad(k * k * IC, dtype)
        gg = GGSpec(B=B, GH=BH, GW=BW, IH=h, IW=w, IC=IC, SY=1, SX=1, DY=-1, DX=-1, TH=k, TW=k,
                    y0=[p], x0=[p], N=N, Kp=Kp, OH=BH, OW=BW, OC=padc(N, dtype), OSY=1, OSX=1,
                    ooy=[0], oox=[0], nphase=1)
        pk = PackSpec(nphase=1, N=N, C=C, IC=IC, TH=k, TW=k, Kp=Kp, s_n=s_n, s_c=s_c, KW=k,
                      kh0=[0], kw0=[0], kh_step=1, kw_step=1, KHW=k * k)
        return gg, pk
    if s != 2 or k % 2 != 0:
        raise NotImplementedError(f"transposed geometry supports stride 1, or stride 2 with even k (got k={k}, s={s})")
    T = k // 2
    Kp = kpad(T * T * IC, dtype)
    y0, x0, kh0, kw0, ooy, oox = [], [], [], [], [], []
    for py in range(2):
        for px in range(2):
            y0.append((py + p) // 2)
            x0.append((px + p) // 2)
            kh0.append((py + p) % 2)
            kw0.append((px + p) % 2)
            ooy.append(py)
            oox.append(px)
    gg = GGSpec(B=B, GH=(BH + 1) // 2, GW=(BW + 1) // 2, IH=h, IW=w, IC=IC, SY=1, SX=1, DY=-1, DX=-1,
                TH=T, TW=T, y0=y0, x0=x0, N=N, Kp=Kp, OH=BH, OW=BW, OC=padc(N, dtype), OSY=2, OSX=2,
                ooy=ooy, oox=oox, nphase=4)
    pk = PackSpec(nphase=4, N=N, C=C, IC=IC, TH=T, TW=T, Kp=Kp, s_n=s_n, s_c=s_c, KW=k,
                  kh0=kh0, kw0=kw0, kh_step=2, kw_step=2, KHW=k * k)
    return gg, pk


def _tap_in_n(B, C, N_out, khw, dtype, s_n, s_c) -> Tuple[GGSpec, PackSpec]:
    """1x1 small tensor, stride 1, no padding: out[b, tap, co] = sum_ci x[b, ci] * w[ci, co, tap]
    as ONE plain GEMM with N = taps*Cout (no zero taps)."""
    IC = padc(C, dtype)
    Kp = kpad(IC, dtype)
    N = khw * N_out
    gg = GGSpec(B=B, GH=1, GW=1, IH=1, IW=1, IC=IC, SY=1, SX=1, DY=1, DX=1, TH=1, TW=1, y0=[0], x0=[0],
                N=N, Kp=Kp, OH=1, OW=1, OC=N, OSY=1, OSX=1, ooy=[0], oox=[0], nphase=1)
    pk = PackSpec(nphase=1, N=N, C=C, IC=IC, TH=1, TW=1, Kp=Kp, s_n=s_n, s_c=s_c, KW=1,
                  kh0=[0], kw0=[0], kh_step=1, kw_step=1, tap_in_n=1, KHW=khw)
    return gg, pk


# ------------------------------------------------------------------------------------------------
# Narrow-N transposed form (vg_tnconv, csrc/edge_conv.hip): big[b][s*iy - p + kh][..][n] += small[b][iy][ix][:] . w[:][n][kh][kw]
# as one GEMM per input pixel over the channels + col2im.  For the 3-channel image side of the networks.
# ------------------------------------------------------------------------------------------------
@dataclass
class TNSpec:
    B: int; IH: int; IW: int; C: int; N: int; K: int; S: int; P: int; OH: int; OW: int; OC: int; Wpitch: int

    def flops(self) -> int:
        return 2 * self.B * self.IH * self.IW * self.C * self.N * self.K * self.K


def tn_spec(B, h, w, C, N, k, s, p, dtype, s_n, s_c):
    """-> (TNSpec, PackSpec) or None when vg_tnconv does not take the shape (the gather-GEMM form is used then).
    s_n / s_c: element strides of the n / c index in the [c][n][k][k]-ordered weight."""
    if dtype != BF16 or C not in (16, 32, 64) or N > 4 or k * k * N > 64 or (k, s) not in ((3, 1), (4, 2)):
        return None
    npix = 512 if (k * k * N + 15) // 16 <= 2 else 256     # input pixels per workgroup tile (csrc/edge_conv.hip: tn_plan)
    if w % 16 != 0 or h != w:
        return None
    OH, OW = convT_out(h, k, s, p), convT_out(w, k, s, p)
    rows_needed = (s - 1 + k - 1) // s + 1                 # input rows behind the smallest output row block
    if rows_needed * 16 > npix:                            # (maps wider than the tile are cut into column blocks: tn_plan)
        return None
    Kp = kpad(C, dtype)
    tn = TNSpec(B=B, IH=h, IW=w, C=C, N=N, K=k, S=s, P=p, OH=OH, OW=OW, OC=padc(N, dtype), Wpitch=Kp)
    pk = PackSpec(nphase=1, N=k * k * N, C=C, IC=C, TH=1, TW=1, Kp=Kp, s_n=s_n, s_c=s_c, KW=1,
                  kh0=[0], kw0=[0], kh_step=1, kw_step=1, tap_in_n=1, KHW=k * k)
    return tn, pk


def convT_fprop_tn(B, H, W, Cin, Cout, k, s, p, dtype):
    return tn_spec(B, H, W, Cin, Cout, k, s, p, dtype, s_n=k * k, s_c=Cout * k * k)


def conv_dgrad_tn(B, H, W, Cin, Cout, k, s, p, dtype):
    """dx [B,H,W,Cin] from dy [B,OH,OW,Cout] (OH = conv_out(H)); only when the transposed map covers H exactly."""
    OH, OW = conv_out(H, k, s, p), conv_out(W, k, s, p)
    if convT_out(OH, k, s, p) != H or convT_out(OW, k, s, p) != W:
        return None
    return tn_spec(B, OH, OW, Cout, Cin, k, s, p, dtype, s_n=k * k, s_c=Cin * k * k)


@dataclass
class EWSpec:
    """vg_edge_wgrad (csrc/edge_conv.hip): wide operand [B][WH][WW][C], narrow operand [B][NH][NW][8] (N real channels)."""
    B: int; WH: int; WW: int; C: int; NH: int; NW: int; N: int; K: int; S: int; P: int; s_c: int; s_n: int

    def flops(self) -> int:
        return 2 * self.B * self.WH * self.WW * self.C * self.N * self.K * self.K


def edge_wgrad_spec(B, wh, ww, C, nh, nw, N, k, s, p, dtype, s_c, s_n):
    """-> EWSpec or None when the edge-layer weight-gradient kernel does not take the shape."""
    if dtype != BF16 or C not in (16, 32, 64) or N > 3 or k not in (3, 4) or s not in (1, 2) or ww > 256:
        return None
    R = 256 // ww
    while R > 1 and (((R - 1) * s + k) * ((ww - 1) * s + k) + 1) * 16 > 22 * 1024:
        R -= 1
    if (((R - 1) * s + k) * ((ww - 1) * s + k) + 1) * 16 > 22 * 1024:
        return None
    return EWSpec(B=B, WH=wh, WW=ww, C=C, NH=nh, NW=nw, N=N, K=k, S=s, P=p, s_c=s_c, s_n=s_n)


def conv_wgrad_edge(B, H, W, Cin, Cout, k, s, p, dtype):
    """nn.Conv2d(Cin <= 3, Cout): wide = dY [B,OH,OW,Cout], narrow = the input image; dW [Cout][Cin][k][k]."""
    OH, OW = conv_out(H, k, s, p), conv_out(W, k, s, p)
    return edge_wgrad_spec(B, OH, OW, Cout, H, W, Cin, k, s, p, dtype, s_c=Cin * k * k, s_n=k * k)


def convT_wgrad_edge(B, H, W, Cin, Cout, k, s, p, dtype):
    """nn.ConvTranspose2d(Cin, Cout <= 3): wide = the input [B,H,W,Cin], narrow = dY [B,OH,OW,8]; dW [Cin][Cout][k][k]."""
    OH, OW = convT_out(H, k, s, p), convT_out(W, k, s, p)
    return edge_wgrad_spec(B, H, W, Cin, OH, OW, Cout, k, s, p, dtype, s_c=Cout * k * k, s_n=k * k)


# ---- nn.Conv2d(Cin, Cout, k, s, p): weight [Cout][Cin][k][k] ---------------------------------------
def conv_fprop(B, H, W, Cin, Cout, k, s, p, dtype):
    return _direct(B, H, W, Cin, Cout, k, s, p, dtype, s_n=Cin * k * k, s_c=k * k)


def conv_dgrad(B, H, W, Cin, Cout, k, s, p, dtype):
    """dx [B,H,W,Cin] from dy [B,OH,OW,Cout]."""
    OH, OW = conv_out(H, k, s, p), conv_out(W, k, s, p)
    return _transposed(B, OH, OW, Cout, H, W, Cin, k, s, p, dtype, s_n=k * k, s_c=Cin * k * k)


def conv_wgrad(B, H, W, Cin, Cout, k, s, p, dtype) -> WGSpec:
    OH, OW = conv_out(H, k, s, p), conv_out(W, k, s, p)
    return WGSpec(B=B, GH=OH, GW=OW, PC=padc(Cout, dtype), NP=Cout, QH=H, QW=W, QC=padc(Cin, dtype), NQ=Cin,
                  SY=s, SX=s, DY=1, DX=1, TH=k, TW=k, y0=-p, x0=-p, s_np=Cin * k * k, s_cq=k * k, s_t=1)


# ---- nn.ConvTranspose2d(Cin, Cout, k, s, p): weight [Cin][Cout][k][k] ------------------------------
def convT_fprop(B, H, W, Cin, Cout, k, s, p, dtype):
    if H == 1 and W == 1 and s == 1 and p == 0 and Cout % per16(dtype) == 0:
        return _tap_in_n(B, Cin, Cout, k * k, dtype, s_n=k * k, s_c=Cout * k * k)
    OH, OW = convT_out(H, k, s, p), convT_out(W, k, s, p)
    return _transposed(B, H, W, Cin, OH, OW, Cout, k, s, p, dtype, s_n=k * k, s_c=Cout * k * k)


def convT_dgrad(B, H, W, Cin, Cout, k, s, p, dtype):
    """dx [B,H,W,Cin] from dy [B,OH,OW,Cout]: a direct convolution of dy."""
    OH, OW = convT_out(H, k, s, p), convT_out(W, k, s, p)
    gg, pk = _direct(B, OH, OW, Cout, Cin, k, s, p, dtype, s_n=Cout * k * k, s_c=k * k)
    assert gg.GH == H and gg.GW == W, (gg.GH, H)
    return gg, pk


def convT_wgrad(B, H, W, Cin, Cout, k, s, p, dtype) -> WGSpec:
    OH, OW = convT_out(H, k, s, p), convT_out(W, k, s, p)
    return WGSpec(B=B, GH=H, GW=W, PC=padc(Cin, dtype), NP=Cin, QH=OH, QW=OW, QC=padc(Cout, dtype), NQ=Cout,
                  SY=s, SX=s, DY=1, DX=1, TH=k, TW=k, y0=-p, x0=-p, s_np=Cout * k * k, s_cq=k * k, s_t=1)


# ---- nn.Linear(C*H*W, N) on the NCHW-flattened map == Conv2d(C, N, k=(H,W)) on the NHWC map --------
def linear_fprop(B, H, W, C, N, dtype):
    assert H == W, "square feature maps only"
    return _direct(B, H, W, C, N, H, 1, 0, dtype, s_n=C * H * W, s_c=H * W)


def linear_dgrad(B, H, W, C, N, dtype):
    """dh [B,H,W,C] from dout [B,N]: GEMM with the spatial taps folded into N (no zero taps)."""
    assert C % per16(dtype) == 0
    return _tap_in_n(B, N, C, H * W, dtype, s_n=H * W, s_c=C * H * W)


def linear_wgrad(B, H, W, C, N, dtype) -> WGSpec:
    return WGSpec(B=B, GH=1, GW=1, PC=padc(N, dtype), NP=N, QH=H, QW=W, QC=padc(C, dtype), NQ=C,
                  SY=1, SX=1, DY=1, DX=1, TH=H, TW=W, y0=0, x0=0, s_np=C * H * W, s_cq=H * W, s_t=1)

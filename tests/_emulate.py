"""Index-level CPU emulation of the kernel descriptor semantics (include/vaegan_hip.h).

Test infrastructure: implements, with plain torch indexing on CPU, exactly the formulas the
header documents for vg_pack_weights / vg_gather_gemm / vg_wgrad, so that (a) the host-side
geometry builders can be validated against torch's convolution semantics without a GPU and
(b) the HIP kernels can be validated against the same formulas on the GPU box.
"""
import torch


def emulate_pack(pk, w):
    """w: flat f32 parameter tensor (reference layout).  -> [nphase][N][Kp] float64."""
    w = w.detach().double().flatten()
    out = torch.zeros(pk.nphase, pk.N, pk.Kp, dtype=torch.float64)
    T = pk.TH * pk.TW
    for p in range(pk.nphase):
        for t in range(T):
            a, c = divmod(t, pk.TW)
            for ci in range(pk.C):
                k = t * pk.IC + ci
                n = torch.arange(pk.N)
                if pk.tap_in_n:
                    CO = pk.N // pk.KHW
                    tap, co = n // CO, n % CO
                    out[p, :, k] = w[ci * pk.s_c + co * pk.s_n + tap]
                else:
                    kh = pk.kh0[p] + pk.kh_step * a
                    kw = pk.kw0[p] + pk.kw_step * c
                    out[p, :, k] = w[n * pk.s_n + ci * pk.s_c + kh * pk.KW + kw]
    return out


def emulate_gg(g, X, Wp, bias=None):
    """X: [B][IH][IW][IC]; Wp: [nphase][N][Kp] -> Y [B][OH][OW][OC] (float64), written mask."""
    X = X.double()
    Y = torch.zeros(g.B, g.OH, g.OW, g.OC, dtype=torch.float64)
    written = torch.zeros(g.B, g.OH, g.OW, dtype=torch.int32)
    for p in range(g.nphase):
        for gy in range(g.GH):
            oy = gy * g.OSY + g.ooy[p]
            if oy >= g.OH:
                continue
            for gx in range(g.GW):
                ox = gx * g.OSX + g.oox[p]
                if ox >= g.OW:
                    continue
                acc = torch.zeros(g.B, g.N, dtype=torch.float64)
                for a in range(g.TH):
                    iy = gy * g.SY + g.y0[p] + g.DY * a
                    if iy < 0 or iy >= g.IH:
                        continue
                    for c in range(g.TW):
                        ix = gx * g.SX + g.x0[p] + g.DX * c
                        if ix < 0 or ix >= g.IW:
                            continue
                        t = a * g.TW + c
                        wk = Wp[p, :, t * g.IC:(t + 1) * g.IC]            # [N][IC]
                        acc += X[:, iy, ix, :] @ wk.T
                if bias is not None:
                    acc = acc + bias.double()[None, :]
                Y[:, oy, ox, :g.N] = acc
                written[:, oy, ox] += 1
    return Y, written


def emulate_wg(wg, P, Q, numel):
    """P: [B][GH][GW][PC], Q: [B][QH][QW][QC] -> flat dW (float64) of `numel` elements."""
    P, Q = P.double(), Q.double()
    dW = torch.zeros(numel, dtype=torch.float64)
    for a in range(wg.TH):
        for c in range(wg.TW):
            t = a * wg.TW + c
            acc = torch.zeros(wg.NP, wg.NQ, dtype=torch.float64)
            for gy in range(wg.GH):
                iy = gy * wg.SY + wg.y0 + wg.DY * a
                if iy < 0 or iy >= wg.QH:
                    continue
                for gx in range(wg.GW):
                    ix = gx * wg.SX + wg.x0 + wg.DX * c
                    if ix < 0 or ix >= wg.QW:
                        continue
                    acc += P[:, gy, gx, :wg.NP].T @ Q[:, iy, ix, :wg.NQ]
            np_i = torch.arange(wg.NP)[:, None]
            cq_i = torch.arange(wg.NQ)[None, :]
            dW[(np_i * wg.s_np + cq_i * wg.s_cq + t * wg.s_t).flatten()] = acc.flatten()
    return dW


def to_nhwc(x, cp):
    """NCHW -> NHWC with channels zero-padded to cp."""
    B, C, H, W = x.shape
    y = torch.zeros(B, H, W, cp, dtype=x.dtype)
    y[..., :C] = x.permute(0, 2, 3, 1)
    return y


def from_nhwc(y, c):
    return y[..., :c].permute(0, 3, 1, 2).contiguous()

"""GPU: every C-ABI kernel against an independent CPU reference (torch fp64 / the oracle's ops),
called through the shared library exactly as the product calls it."""
import importlib
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _emulate import from_nhwc, to_nhwc

pytestmark = pytest.mark.gpu

PKG = "vae-gan-based-model-for-image-generation-and-denoising_amd"
G = importlib.import_module(PKG + ".geometry")


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return importlib.import_module(PKG + ".ops")


DEV = "cuda"
TOL = {G.F32: dict(rtol=2e-5, atol=2e-5), G.BF16: dict(rtol=3e-2, atol=3e-2)}


def close(actual, ref, dtype, f32=(1e-4, 4e-6), bf16=(3e-2, 1.5e-2)):
    """assert_close with the absolute tolerance scaled by the reference magnitude: f32 MFMA is an
    exact fma chain, so the error is ~1e-7 * sum|a*b| (cdna_hip_programming.md section 3)."""
    rtol, arel = f32 if dtype == G.F32 else bf16
    scale = max(1.0, float(ref.abs().max()))
    torch.testing.assert_close(actual, ref, rtol=rtol, atol=arel * scale)


def _q(t, dtype):
    """Round to the storage dtype the kernel sees (so the reference isolates kernel error)."""
    return t.to(torch.bfloat16).double() if dtype == G.BF16 else t.float().double()


def _dev(t, dtype, ops):
    return t.to(ops.TORCH_DT[dtype]).contiguous().to(DEV)


CONV_CASES = [  # B, H, Cin, Cout, k, s, p
    (2, 16, 3, 8, 4, 2, 1), (2, 15, 3, 8, 4, 2, 0), (3, 9, 8, 4, 4, 2, 0), (2, 6, 4, 8, 4, 2, 0),
    (2, 8, 4, 4, 3, 1, 1),
    (8, 64, 3, 32, 4, 2, 0), (8, 31, 32, 64, 4, 2, 0), (8, 14, 64, 128, 4, 2, 0), (16, 6, 128, 256, 4, 2, 0),
    (8, 64, 3, 64, 4, 2, 1), (8, 32, 64, 128, 4, 2, 1), (8, 8, 256, 512, 4, 2, 1),
]


@pytest.mark.parametrize("dtype", [G.F32, G.BF16])
@pytest.mark.parametrize("B,H,Cin,Cout,k,s,p", CONV_CASES)
def test_conv2d_fprop_dgrad_wgrad(ops, dtype, B, H, Cin, Cout, k, s, p):
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + Cin)
    x = _q(torch.randn(B, Cin, H, H, generator=g), dtype)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.2
    b = torch.randn(Cout, generator=g)
    wq = _q(w, dtype)
    y_ref = F.conv2d(x, wq, b.double(), stride=s, padding=p)
    gg, pk = G.conv_fprop(B, H, H, Cin, Cout, k, s, p, dtype)
    Wp = ops.pack_weights(pk, w.to(DEV), dtype)
    X = _dev(to_nhwc(x, gg.IC), dtype, ops)
    Y, stats, nparts = ops.gather_gemm(gg, X, Wp, dtype, bias=b.to(DEV), want_stats=True)
    y = from_nhwc(Y.double().cpu(), Cout)
    close(y, y_ref, dtype)
    assert (Y[..., Cout:] == 0).all()
    # BN statistics emitted by the epilogue (on the f32 accumulators)
    st = stats[: nparts * 2 * Cout].view(nparts, 2, Cout).double().sum(0).cpu()
    torch.testing.assert_close(st[0], y_ref.sum((0, 2, 3)), rtol=1e-3, atol=1e-2 * (1 if dtype == G.F32 else 30))
    torch.testing.assert_close(st[1], (y_ref ** 2).sum((0, 2, 3)), rtol=2e-3 if dtype == G.F32 else 3e-2, atol=1e-2)
    # dgrad
    dy = _q(torch.randn(y_ref.shape, generator=g), dtype)
    dx_ref = torch.nn.grad.conv2d_input(x.shape, wq, dy, stride=s, padding=p)
    gg, pk = G.conv_dgrad(B, H, H, Cin, Cout, k, s, p, dtype)
    Wd = ops.pack_weights(pk, w.to(DEV), dtype)
    DY = _dev(to_nhwc(dy, gg.IC), dtype, ops)
    DX, _, _ = ops.gather_gemm(gg, DY, Wd, dtype)
    close(from_nhwc(DX.double().cpu(), Cin), dx_ref, dtype)
    # wgrad (+ accumulate)
    dw_ref = torch.nn.grad.conv2d_weight(x, w.shape, dy, stride=s, padding=p)
    wg = G.conv_wgrad(B, H, H, Cin, Cout, k, s, p, dtype)
    dW = torch.full(w.shape, 7.0, device=DEV)
    ops.wgrad(wg, DY, X, dW, False, dtype)
    close(dW.double().cpu(), dw_ref, dtype, f32=(1e-4, 2e-5), bf16=(3e-2, 3e-2))
    ops.wgrad(wg, DY, X, dW, True, dtype)
    close(dW.double().cpu(), 2 * dw_ref, dtype, f32=(1e-4, 2e-5), bf16=(3e-2, 3e-2))


CONVT_CASES = [  # B, H, Cin, Cout, k, s, p
    (2, 1, 12, 8, 4, 1, 0), (2, 4, 8, 8, 4, 2, 1), (2, 5, 8, 16, 4, 2, 1), (2, 8, 8, 3, 3, 1, 1),
    (8, 1, 100, 1024, 4, 1, 0), (8, 4, 1024, 512, 4, 2, 1), (8, 32, 128, 64, 4, 2, 1), (4, 64, 64, 3, 3, 1, 1),
]


@pytest.mark.parametrize("dtype", [G.F32, G.BF16])
@pytest.mark.parametrize("B,H,Cin,Cout,k,s,p", CONVT_CASES)
def test_conv_transpose2d_fprop_dgrad_wgrad(ops, dtype, B, H, Cin, Cout, k, s, p):
    g = torch.Generator().manual_seed(B * 77 + H * 10 + Cin)
    x = _q(torch.randn(B, Cin, H, H, generator=g), dtype).requires_grad_(True)
    w = torch.randn(Cin, Cout, k, k, generator=g) * 0.1
    wq = _q(w, dtype).requires_grad_(True)
    y_ref = F.conv_transpose2d(x, wq, None, stride=s, padding=p)
    gg, pk = G.convT_fprop(B, H, H, Cin, Cout, k, s, p, dtype)
    Wp = ops.pack_weights(pk, w.to(DEV), dtype)
    X = _dev(to_nhwc(x.detach(), gg.IC), dtype, ops)
    Y, stats, nparts = ops.gather_gemm(gg, X, Wp, dtype, want_stats=True)
    OH = y_ref.shape[-1]
    Yv = Y.view(B, OH, OH, -1)
    close(from_nhwc(Yv.double().cpu(), Cout), y_ref.detach(), dtype)
    if not pk.tap_in_n:
        st = stats[: nparts * 2 * Cout].view(nparts, 2, Cout).double().sum(0).cpu()
        torch.testing.assert_close(st[0], y_ref.detach().sum((0, 2, 3)), rtol=1e-3,
                                   atol=1e-2 * (1 if dtype == G.F32 else 30))
    dy = _q(torch.randn(y_ref.shape, generator=g), dtype)
    dx_ref, dw_ref = torch.autograd.grad(y_ref, (x, wq), dy)
    gg, pk = G.convT_dgrad(B, H, H, Cin, Cout, k, s, p, dtype)
    Wd = ops.pack_weights(pk, w.to(DEV), dtype)
    DY = _dev(to_nhwc(dy, gg.IC), dtype, ops)
    DX, _, _ = ops.gather_gemm(gg, DY, Wd, dtype)
    close(from_nhwc(DX.double().cpu(), Cin), dx_ref, dtype)
    wg = G.convT_wgrad(B, H, H, Cin, Cout, k, s, p, dtype)
    dW = torch.zeros(w.shape, device=DEV)
    ops.wgrad(wg, X, DY, dW, False, dtype)
    close(dW.double().cpu(), dw_ref, dtype, f32=(1e-4, 2e-5), bf16=(3e-2, 3e-2))


@pytest.mark.parametrize("B,H,Cin,Cout", [(64, 4, 1024, 512), (128, 4, 1024, 512), (128, 4, 512, 256), (96, 4, 512, 256)])
def test_few_row_long_k_data_gradient_on_128x128_tiles_with_dma_k_slices(ops, vg_switch, B, H, Cin, Cout):
    """Round 4: the data gradient of a deep ConvTranspose2d (gan_code.py:25-28; few rows, K = 16 * Cout) on 128 x 128
    tiles whose K dimension is cut into slices that run on the LDS-DMA ring (VG_SPLITK_BIGK=1: conv_gemm.hip bigk_split_ok,
    gg_kernel<.., SPLITK, DMA>), against the 64 x 64 tiles it replaces and against torch in fp64 on the same bf16
    operands.  (128, 4, 1024, 512) is the Generator's stage-1 data gradient at the benchmark size."""
    dtype, k, s, p = G.BF16, 4, 2, 1
    g = torch.Generator().manual_seed(B + Cin)
    x = _q(torch.randn(B, Cin, H, H, generator=g), dtype).requires_grad_(True)
    w = torch.randn(Cin, Cout, k, k, generator=g) * 0.05
    wq = _q(w, dtype)
    y_ref = F.conv_transpose2d(x, wq, None, stride=s, padding=p)
    dy = _q(torch.randn(y_ref.shape, generator=g), dtype)
    (dx_ref,) = torch.autograd.grad(y_ref, x, dy)
    gg, pk = G.convT_dgrad(B, H, H, Cin, Cout, k, s, p, dtype)
    Wd = ops.pack_weights(pk, w.to(DEV), dtype)
    DY = _dev(to_nhwc(dy, gg.IC), dtype, ops)
    out = {}
    for mode in ("0", "1"):
        vg_switch("VG_SPLITK_BIGK", mode)
        n0 = ops.launch_count()
        DX, _, _ = ops.gather_gemm(gg, DY, Wd, dtype)
        out[mode] = (DX.clone(), ops.launch_count() - n0)
    torch.cuda.synchronize()
    expect_split = (B * H * H) % 128 == 0 and (B * H * H // 128) * (Cin // 128) >= 64
    if expect_split:
        assert out["1"][1] == 2                                               # main launch + slab reduce
    for mode in ("0", "1"):
        close(from_nhwc(out[mode][0].double().cpu(), Cin), dx_ref, dtype)
    d = (out["1"][0].float() - out["0"][0].float()).abs()
    assert float((d > 0).float().mean()) < 2e-3                               # same sums, another order: rare one-ulp flips


@pytest.mark.parametrize("kind,B,H,Cin,Cout,k,s,p", [("conv", 128, 8, 256, 512, 4, 2, 1), ("conv", 128, 6, 128, 256, 4, 2, 0),
                                                      ("conv", 32, 14, 64, 128, 4, 2, 0), ("convT", 32, 4, 1024, 512, 4, 2, 1),
                                                      ("convT", 16, 8, 512, 256, 4, 2, 1), ("conv", 37, 8, 256, 512, 4, 2, 1)])
def test_split_k_with_batchnorm_statistics_and_subpixel_phases(ops, vg_switch, kind, B, H, Cin, Cout, k, s, p):
    """Round 4: small-M forward launches (64 x 64 tiles, <= one workgroup per CU) cut along K -- also with sub-pixel phases
    (ConvTranspose2d) and with the BatchNorm partial sums, which the slab reduce forms (splitk_reduce2_kernel) -- against the
    un-split launch (VG_SPLITK_GENERAL=0) and torch in fp64 on the same bf16 operands: outputs, bias, statistics slabs
    (same slab count), zero padding channels.  (128, 8, 256, 512) is the Discriminator's last conv at B = 128."""
    dtype = G.BF16
    g = torch.Generator().manual_seed(B * 3 + Cin)
    x = _q(torch.randn(B, Cin, H, H, generator=g), dtype)
    if kind == "conv":
        w = torch.randn(Cout, Cin, k, k, generator=g) * 0.05
        b = torch.randn(Cout, generator=g)
        y_ref = F.conv2d(x, _q(w, dtype), b.double(), stride=s, padding=p)
        gg, pk = G.conv_fprop(B, H, H, Cin, Cout, k, s, p, dtype)
    else:
        w = torch.randn(Cin, Cout, k, k, generator=g) * 0.05
        b = None
        y_ref = F.conv_transpose2d(x, _q(w, dtype), None, stride=s, padding=p)
        gg, pk = G.convT_fprop(B, H, H, Cin, Cout, k, s, p, dtype)
    Wp = ops.pack_weights(pk, w.to(DEV), dtype)
    X = _dev(to_nhwc(x, gg.IC), dtype, ops)
    out = {}
    for mode in ("0", "1"):
        vg_switch("VG_SPLITK_GENERAL", mode)                 # (opt-in: off by default)
        n0 = ops.launch_count()
        Y, stats, nparts = ops.gather_gemm(gg, X, Wp, dtype, bias=None if b is None else b.to(DEV), want_stats=True)
        out[mode] = (Y.clone(), stats[: nparts * 2 * Cout].clone().view(nparts, 2, Cout), nparts, ops.launch_count() - n0)
    torch.cuda.synchronize()
    assert out["0"][3] == 1 and out["1"][3] == 2 and out["0"][2] == out["1"][2]      # split: main + slab reduce; same slab rows
    OH = y_ref.shape[-1]
    for mode in ("0", "1"):
        Yv = out[mode][0].view(B, OH, OH, -1)
        close(from_nhwc(Yv.double().cpu(), Cout), y_ref, dtype)
        assert (Yv[..., Cout:] == 0).all()
        st = out[mode][1].double().sum(0).cpu()
        torch.testing.assert_close(st[0], y_ref.sum((0, 2, 3)), rtol=1e-3, atol=3e-1)
        torch.testing.assert_close(st[1], (y_ref ** 2).sum((0, 2, 3)), rtol=3e-2, atol=1e-2)
    # slab by slab: the same rows feed the same slab in both forms
    torch.testing.assert_close(out["1"][1], out["0"][1], rtol=2e-3, atol=2e-2)
    d = (out["1"][0].float() - out["0"][0].float()).abs()
    assert float((d > 0).float().mean()) < 2e-3


@pytest.mark.parametrize("dtype", [G.F32, G.BF16])
def test_linear_fused_heads(ops, dtype):
    B, Hf, C, N = 8, 2, 256, 200
    g = torch.Generator().manual_seed(5)
    h = _q(torch.randn(B, C, Hf, Hf, generator=g), dtype).requires_grad_(True)
    w = torch.randn(N, C * Hf * Hf, generator=g) * 0.05
    b = torch.randn(N, generator=g)
    wq = _q(w, dtype).requires_grad_(True)
    y_ref = F.linear(h.view(B, -1), wq, b.double())
    gg, pk = G.linear_fprop(B, Hf, Hf, C, N, dtype)
    X = _dev(to_nhwc(h.detach(), gg.IC), dtype, ops)
    Y, _, _ = ops.gather_gemm(gg, X, ops.pack_weights(pk, w.to(DEV), dtype), dtype, bias=b.to(DEV))
    close(Y.view(B, -1)[:, :N].double().cpu(), y_ref.detach(), dtype)
    dy = _q(torch.randn(B, N, generator=g), dtype)
    dh_ref, dw_ref = torch.autograd.grad(y_ref, (h, wq), dy)
    gg, pk = G.linear_dgrad(B, Hf, Hf, C, N, dtype)
    DY = _dev(dy.view(B, 1, 1, N), dtype, ops)
    DH, _, _ = ops.gather_gemm(gg, DY, ops.pack_weights(pk, w.to(DEV), dtype), dtype)
    close(from_nhwc(DH.view(B, Hf, Hf, C).double().cpu(), C), dh_ref, dtype)
    wg = G.linear_wgrad(B, Hf, Hf, C, N, dtype)
    dW = torch.zeros(w.shape, device=DEV)
    ops.wgrad(wg, DY, X, dW, False, dtype)
    close(dW.double().cpu(), dw_ref, dtype, f32=(1e-4, 2e-5), bf16=(3e-2, 3e-2))


@pytest.mark.parametrize("dtype", [G.F32, G.BF16])
@pytest.mark.parametrize("rows,C,act,slope", [(8 * 31 * 31, 32, 2, 0.01), (4 * 16, 1024, 1, 0.0), (5 * 7, 200, 2, 0.2),
                                               (128 * 64, 64, 2, 0.2)])
def test_batchnorm_activation_forward_backward(ops, dtype, rows, C, act, slope):
    """nn.BatchNorm2d(train) + (Leaky)ReLU forward/backward vs torch autograd in fp64."""
    g = torch.Generator().manual_seed(rows + C)
    x = _q(torch.randn(rows, C, generator=g) * 1.7 + 0.3, dtype).requires_grad_(True)
    gamma = (torch.randn(C, generator=g) * 0.1 + 1).double().requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.1).double().requires_grad_(True)
    rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    xn = x.view(rows, C, 1, 1)
    z = F.batch_norm(xn, rm, rv, gamma, beta, True, 0.1, 1e-5)
    a_ref = F.leaky_relu(z, slope) if act == 2 else F.relu(z)
    dy = _q(torch.randn(rows, C, generator=g), dtype)
    dx_ref, dg_ref, db_ref = torch.autograd.grad(a_ref, (x, gamma, beta), dy.view(rows, C, 1, 1))
    X = _dev(x.detach(), dtype, ops)
    stats, nparts = ops.channel_stats(X, rows, C, dtype)
    rmd, rvd = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    co = ops.bn_finalize(stats, nparts, C, rows, gamma.detach().float().to(DEV), beta.detach().float().to(DEV),
                         rmd, rvd, 0.1, 1e-5, DEV)
    A = ops.bn_act_forward(X, co, rows, C, act, slope, dtype)
    torch.testing.assert_close(A.double().cpu(), a_ref.detach().view(rows, C), **TOL[dtype])
    torch.testing.assert_close(rmd.double().cpu(), rm, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rvd.double().cpu(), rv, rtol=1e-5, atol=1e-6)
    dgam, dbet = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    DX = ops.bn_act_backward(X, _dev(dy, dtype, ops), co, rows, C, rows, gamma.detach().float().to(DEV), act, slope,
                             dgam, dbet, False, dtype)
    t = dict(rtol=1e-3, atol=1e-4) if dtype == G.F32 else dict(rtol=5e-2, atol=5e-2)
    torch.testing.assert_close(DX.double().cpu(), dx_ref, **t)
    tg = dict(rtol=1e-4, atol=1e-3) if dtype == G.F32 else dict(rtol=5e-2, atol=5e-1)
    torch.testing.assert_close(dgam.double().cpu(), dg_ref, **tg)
    torch.testing.assert_close(dbet.double().cpu(), db_ref, **tg)


def test_bn_eval_and_activation_only(ops):
    C, rows = 64, 333
    x = torch.randn(rows, C)
    g_, b_, rm, rv = torch.rand(C) + 0.5, torch.randn(C), torch.randn(C), torch.rand(C) + 0.5
    ref = F.leaky_relu(F.batch_norm(x.view(rows, C, 1, 1), rm, rv, g_, b_, False, 0.1, 1e-5), 0.2).view(rows, C)
    co = ops.bn_eval_coeffs(g_.to(DEV), b_.to(DEV), rm.to(DEV), rv.to(DEV), 1e-5)
    out = ops.bn_act_forward(x.to(DEV), co, rows, C, 2, 0.2, G.F32)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-5)
    out = ops.bn_act_forward(x.to(DEV), None, rows, C, 2, 0.2, G.F32)
    torch.testing.assert_close(out.cpu(), F.leaky_relu(x, 0.2))
    dy = torch.randn(rows, C)
    dx = ops.act_backward(x.to(DEV), dy.to(DEV), 2, 0.2, G.F32)
    torch.testing.assert_close(dx.cpu(), torch.where(x > 0, dy, dy * 0.2))
    db = torch.zeros(C, device=DEV)
    ops.bias_grad(dy.to(DEV), rows, C, C, db, False, G.F32)
    torch.testing.assert_close(db.cpu(), dy.sum(0), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("dtype", [G.F32, G.BF16])
def test_layout_noise_tanh(ops, dtype):
    B, C, H = 3, 3, 10
    x, e = torch.randn(B, C, H, H), torch.randn(B, C, H, H)
    CP = G.padc(C, dtype)
    y = ops.nchw_to_nhwc(x.to(DEV), CP, dtype, eps=e.to(DEV), sigma=0.05)
    ref = _q(x + 0.05 * e, dtype)
    torch.testing.assert_close(from_nhwc(y.double().cpu(), C), ref, rtol=1e-6, atol=1e-6)
    assert (y[..., C:] == 0).all()
    back = ops.nhwc_to_nchw(y, C, dtype, apply_tanh=True)
    torch.testing.assert_close(back.double().cpu(), torch.tanh(from_nhwc(y.double().cpu(), C)), rtol=1e-5, atol=1e-6)
    dy = torch.randn(B, C, H, H)
    dx = ops.nchw_grad_to_nhwc(dy.to(DEV), back, CP, dtype)
    ref = dy.double() * (1 - back.double().cpu() ** 2)
    torch.testing.assert_close(from_nhwc(dx.double().cpu(), C), _q(ref, dtype), rtol=1e-2 if dtype else 1e-5,
                               atol=1e-2 if dtype else 1e-6)


@pytest.mark.parametrize("C,N,k,s,p,H,B", [(64, 3, 3, 1, 1, 64, 3),      # Generator's last layer at S=64 (gan_code.py:49)
                                            (64, 3, 4, 2, 1, 32, 3),      # image gradient below D's first conv, S=64
                                            (32, 3, 3, 1, 1, 128, 2),     # S=128 members of both
                                            (32, 3, 4, 2, 1, 64, 2),
                                            (64, 3, 3, 1, 1, 16, 1),      # whole image in one tile
                                            (64, 1, 4, 2, 1, 48, 2),      # N=1, rows that do not fill the last tile
                                            (32, 4, 4, 2, 1, 16, 1),      # K*K*N = 64: all four column tiles
                                            (16, 3, 3, 1, 1, 64, 2),      # C = 16 (the 16-channel members at S=256): half a
                                            (16, 3, 4, 2, 1, 32, 2),      # 32-channel chunk per pixel, zero weight rows beyond
                                            (16, 3, 3, 1, 1, 256, 1),     # S=256: the Generator's last layer as written (gan_code.py:49),
                                            (16, 3, 4, 2, 1, 128, 1),     # the image gradient below gan_code.py:61 -- column-block tiles
                                            (32, 2, 3, 1, 1, 192, 1)])    # a width that is no power of two
def test_edge_layer_transposed_conv_vs_torch(ops, C, N, k, s, p, H, B):
    """vg_tnconv (GEMM per input pixel + col2im, csrc/edge_conv.hip) against torch's conv_transpose2d in fp64 on the
    bf16-rounded operands: plain NHWC output, then the fused Tanh / NCHW image / instance-noise epilogue."""
    g = torch.Generator().manual_seed(C * 100 + k * 10 + H)
    x = torch.randn(B, C, H, H, generator=g)
    w = torch.randn(C, N, k, k, generator=g) * 0.05
    sp = G.tn_spec(B, H, H, C, N, k, s, p, G.BF16, s_n=k * k, s_c=N * k * k)
    assert sp is not None
    tn, pk = sp
    xh = _dev(to_nhwc(x, C), G.BF16, ops)
    wp = ops.pack_weights(pk, w.to(DEV), G.BF16)
    ref = F.conv_transpose2d(_q(x, G.BF16), _q(w, G.BF16), stride=s, padding=p)
    y, _ = ops.tnconv(tn, xh, wp)
    assert tuple(y.shape) == (B, tn.OH, tn.OW, 8) and (y[..., N:] == 0).all()
    close(from_nhwc(y.double().cpu(), N), _q(ref, G.BF16), G.BF16, bf16=(1e-2, 8e-3))
    # fused epilogue: tanh -> NCHW f32, and tanh + sigma*eps -> NHWC bf16 (vaegan_code.py:83, :92)
    eps = torch.randn(B, N, tn.OH, tn.OW, generator=g)
    yn, img = ops.tnconv(tn, xh, wp, want_nchw=True, act=3, noise=eps.to(DEV), sigma=0.05)
    torch.testing.assert_close(img.double().cpu(), torch.tanh(ref), rtol=1e-5, atol=2e-5)       # f32 accumulators: no bf16 rounding before tanh
    close(from_nhwc(yn.double().cpu(), N), _q(img.double().cpu() + 0.05 * eps.double(), G.BF16), G.BF16, bf16=(1e-2, 4e-3))
    ns = ops.NoiseStream(DEV, 77)
    ns.advance()
    yr, img2 = ops.tnconv(tn, xh, wp, want_nchw=True, act=3, noise=ns.draw(2), sigma=0.05)     # in-kernel draw ==
    ym, _ = ops.tnconv(tn, xh, wp, want_nchw=True, act=3, noise=ns.randn((B, N, tn.OH, tn.OW), 2), sigma=0.05)   # materialised
    assert torch.equal(yr, ym) and torch.equal(img2, img)
    # image only (autograd drop-in path of Generator.forward)
    none, img3 = ops.tnconv(tn, xh, wp, want_nhwc=False, want_nchw=True, act=3)
    assert none is None and torch.equal(img3, img)


def test_edge_layer_kernel_rejects_what_it_does_not_take(ops):
    assert G.tn_spec(2, 40, 40, 16, 3, 3, 1, 1, G.BF16, 9, 27) is None          # rows are not whole 16-pixel groups
    assert G.tn_spec(2, 64, 64, 64, 3, 3, 1, 1, G.F32, 9, 27) is None           # exact-f32 parity path: gather-GEMM
    assert G.tn_spec(2, 64, 64, 64, 8, 3, 1, 1, G.BF16, 9, 72) is None          # not narrow
    assert G.tn_spec(2, 31, 31, 64, 3, 3, 1, 1, G.BF16, 9, 27) is None          # rows are not whole 16-pixel groups
    tn, pk = G.tn_spec(1, 16, 16, 64, 3, 3, 1, 1, G.BF16, 9, 27)
    x = torch.zeros(1, 16, 16, 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="size mismatch"):
        ops.tnconv(tn, x, torch.zeros(5, dtype=torch.bfloat16, device=DEV))


@pytest.mark.parametrize("kind,C,k,s,p,H,B", [("conv", 64, 4, 2, 1, 64, 3),     # Discriminator's first layer, S=64 (gan_code.py:61)
                                              ("convT", 64, 3, 1, 1, 64, 2),   # Generator's last layer, S=64 (gan_code.py:49)
                                              ("conv", 32, 4, 2, 0, 64, 2),    # Encoder's first layer: 31x31 outputs, p=0
                                              ("conv", 32, 4, 2, 1, 128, 1),   # S=128 members
                                              ("convT", 32, 3, 1, 1, 128, 1),
                                              ("conv", 64, 4, 2, 1, 16, 5),    # many images per workgroup walk, tiny maps
                                              ("conv", 16, 4, 2, 1, 256, 1),   # S=256 members: 16 channels on the wide side
                                              ("convT", 16, 3, 1, 1, 256, 1),  # (gan_code.py:49 and :61 as written)
                                              ("conv", 16, 4, 2, 1, 32, 3)])
def test_edge_layer_weight_gradient_vs_torch(ops, kind, C, k, s, p, H, B):
    """vg_edge_wgrad (transposed LDS reads of both operands, narrow operand read straight from the image patch) against
    torch's convolution weight gradient in fp64 on the bf16-rounded operands; plain and accumulating."""
    g = torch.Generator().manual_seed(C + k + H)
    if kind == "conv":
        x = torch.randn(B, 3, H, H, generator=g)
        OH = G.conv_out(H, k, s, p)
        dy = torch.randn(B, C, OH, OH, generator=g)
        xq = _q(x, G.BF16).requires_grad_(False)
        w = torch.zeros(C, 3, k, k, dtype=torch.float64, requires_grad=True)
        (F.conv2d(xq, w, stride=s, padding=p) * _q(dy, G.BF16)).sum().backward()
        ew = G.conv_wgrad_edge(B, H, H, 3, C, k, s, p, G.BF16)
        wide, narrow = _dev(to_nhwc(dy, C), G.BF16, ops), _dev(to_nhwc(x, 8), G.BF16, ops)
    else:
        x = torch.randn(B, C, H, H, generator=g)
        OH = G.convT_out(H, k, s, p)
        dy = torch.randn(B, 3, OH, OH, generator=g)
        w = torch.zeros(C, 3, k, k, dtype=torch.float64, requires_grad=True)
        (F.conv_transpose2d(_q(x, G.BF16), w, stride=s, padding=p) * _q(dy, G.BF16)).sum().backward()
        ew = G.convT_wgrad_edge(B, H, H, C, 3, k, s, p, G.BF16)
        wide, narrow = _dev(to_nhwc(x, C), G.BF16, ops), _dev(to_nhwc(dy, 8), G.BF16, ops)
    assert ew is not None
    ref = w.grad
    dW = torch.full((C, 3, k, k), float("nan"), device=DEV)
    ops.edge_wgrad(ew, wide, narrow, dW, False)
    close(dW.double().cpu(), ref, G.F32, f32=(1e-4, 2e-5))            # exact bf16 products, f32 accumulation
    ops.edge_wgrad(ew, wide, narrow, dW, True)                        # accumulate: twice the gradient
    close(dW.double().cpu(), 2 * ref, G.F32, f32=(1e-4, 4e-5))
    first = dW.clone()
    ops.edge_wgrad(ew, wide, narrow, dW, False)
    ops.edge_wgrad(ew, wide, narrow, first, False)
    assert torch.equal(first, dW)                                     # fixed summation order: bitwise reproducible


def _e4m3(t):
    """Round to OCP e4m3fn the way the kernels' operands are (torch's float8_e4m3fn cast: round to nearest even)."""
    return t.float().to(torch.float8_e4m3fn).double()


@pytest.mark.parametrize("kind,cin,cout,h,B", [("conv", 64, 128, 16, 3), ("convT", 128, 64, 8, 3), ("conv", 16, 32, 24, 2),
                                               ("convT", 32, 16, 12, 5)])
def test_fp8_gather_gemm_vs_torch_on_e4m3_operands(ops, kind, cin, cout, h, B):
    """VG_FP8 launches of vg_gather_gemm (block-scaled v_mfma_scale_f32_16x16x128_f8f6f4, BASELINE configs[4]):
    exact f32 accumulation of e4m3 x e4m3 products, so against torch's convolution in fp64 on the SAME e4m3-rounded
    operands only the bf16 rounding of the output remains.  Weights are stored * 2^6 and un-scaled by the MFMA's E8M0
    operand.  k4 s2 p1 conv and transposed conv (4 sub-pixel phases), BatchNorm partial sums in the epilogue."""
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(B, cin, h, h, generator=g)
    if kind == "conv":
        w = torch.randn(cout, cin, 4, 4, generator=g) * 0.02
        gg, pk = G.conv_fprop(B, h, h, cin, cout, 4, 2, 1, G.BF16)
    else:
        w = torch.randn(cin, cout, 4, 4, generator=g) * 0.02
        gg, pk = G.convT_fprop(B, h, h, cin, cout, 4, 2, 1, G.BF16)
    assert gg.Kp % 64 == 0 and gg.IC % 16 == 0
    xb = _dev(to_nhwc(x, gg.IC), G.BF16, ops)
    wp = ops.pack_weights(pk, w.to(DEV), G.BF16)
    x8, w8 = ops.cast_fp8(xb), ops.cast_fp8(wp, 6)
    Y, stats, nparts = ops.gather_gemm(gg, x8, w8, G.FP8, want_stats=True)
    assert Y.dtype == torch.bfloat16
    xq = _e4m3(_q(x, G.BF16))
    wq = _e4m3(_q(w, G.BF16) * 64) / 64
    ref = F.conv2d(xq, wq, stride=2, padding=1) if kind == "conv" else F.conv_transpose2d(xq, wq, stride=2, padding=1)
    got = from_nhwc(Y.double().cpu(), cout)
    close(got, _q(ref, G.BF16), G.BF16, bf16=(1e-2, 4e-3))
    # BatchNorm partial sums of the epilogue: column sums of the f32 accumulators (before the bf16 rounding)
    st = stats[:nparts * 2 * gg.N].view(nparts, 2, gg.N).double().cpu().sum(0)
    torch.testing.assert_close(st[0], ref.sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(st[1], (ref * ref).sum((0, 2, 3)), rtol=1e-4, atol=1e-3)


def test_in_kernel_noise_equals_materialised_draws_and_is_standard_normal(ops):
    """vg_*_rng (the three randn_like draws of vaegan_code.py:77,91,92 generated inside the consuming kernels) against
    the same kernels fed with vg_randn's materialisation of the same (seed, iteration, draw): bitwise.  Plus the
    generator itself: moments of N(0,1), distinct draws / iterations, reproducibility per seed."""
    ns = ops.NoiseStream(DEV, 1234)
    ns.advance()
    B, C, H, Ld = 4, 3, 16, 100
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, C, H, H, generator=g).to(DEV)
    for dtype in (G.F32, G.BF16):
        CP = G.padc(C, dtype)
        e1 = ns.randn((B, C, H, H), 1)
        assert torch.equal(ops.nchw_to_nhwc(x, CP, dtype, eps=ns.draw(1), sigma=0.05),
                           ops.nchw_to_nhwc(x, CP, dtype, eps=e1, sigma=0.05))
        pre = ops.nchw_to_nhwc(x, CP, dtype)
        oa, ob = torch.empty_like(pre), torch.empty_like(pre)
        ya = ops.nhwc_tanh_to_nchw_noisy(pre, C, ns.draw(2), 0.05, oa, dtype)
        yb = ops.nhwc_tanh_to_nchw_noisy(pre, C, ns.randn((B, C, H, H), 2), 0.05, ob, dtype)
        assert torch.equal(ya, yb) and torch.equal(oa, ob)
    mulv = (torch.randn(B, 2 * Ld, generator=g) * 3).to(DEV)
    ez = ns.randn((B, Ld), 0)
    za, la = ops.reparam_forward(mulv, ns.draw(0), Ld, Ld, G.F32)
    zb, lb = ops.reparam_forward(mulv, ez, Ld, Ld, G.F32)
    assert torch.equal(za, zb) and torch.equal(la, lb)
    dz = torch.randn(B, 1, 1, Ld, generator=g).to(DEV)
    assert torch.equal(ops.reparam_kl_backward(mulv, la, ns.draw(0), dz, 0.01, Ld, G.F32),      # backward regenerates
                       ops.reparam_kl_backward(mulv, la, ez, dz, 0.01, Ld, G.F32))               # the forward's eps
    # the generator
    big = ns.randn((1 << 20,), 5).double().cpu()
    assert abs(float(big.mean())) < 4e-3 and abs(float(big.var()) - 1) < 6e-3
    assert abs(float((big ** 4).mean()) - 3) < 0.05 and abs(float((big ** 3).mean())) < 0.02   # kurtosis, skew of N(0,1)
    assert float(big.abs().max()) < 7 and bool(torch.isfinite(big).all())
    lag = float((big[:-1] * big[1:]).mean())
    assert abs(lag) < 4e-3                                                  # neighbouring counters are uncorrelated
    other = ns.randn((1 << 20,), 6).double().cpu()
    assert abs(float((big * other).mean())) < 4e-3                         # draws are independent streams
    again = ns.randn((1 << 20,), 5).double().cpu()
    assert torch.equal(big, again)                                          # same iteration: same numbers
    ns.advance()
    nxt = ns.randn((1 << 20,), 5).double().cpu()
    assert not torch.equal(big, nxt) and abs(float((big * nxt).mean())) < 4e-3      # next iteration: fresh noise
    ns2 = ops.NoiseStream(DEV, 1234)
    ns2.advance()
    assert torch.equal(ns2.randn((1 << 20,), 5).double().cpu(), big)        # reproducible per seed
    ns3 = ops.NoiseStream(DEV, 1235)
    ns3.advance()
    assert not torch.equal(ns3.randn((1 << 20,), 5).double().cpu(), big)


def test_reparam_kl_forward_backward(ops):
    """vaegan_code.py:75-77,114 incl. the clamp's gradient mask."""
    B, Ld = 6, 100
    g = torch.Generator().manual_seed(3)
    mulv = (torch.randn(B, 2 * Ld, generator=g) * 6).double().requires_grad_(True)   # some |logvar| > 10
    eps = torch.randn(B, Ld, generator=g).double()
    mu, lv = mulv[:, :Ld], torch.clamp(mulv[:, Ld:], -10, 10)
    z = mu + torch.exp(0.5 * lv) * eps
    kl = -0.5 * torch.sum(1 + lv - mu.pow(2) - lv.exp()) / B
    dz = torch.randn(B, Ld, generator=g).double()
    (z * dz).sum().add(0.07 * kl).backward()
    M = mulv.detach().float().to(DEV)
    zd, lvc = ops.reparam_forward(M, eps.float().to(DEV), Ld, Ld, G.F32)
    torch.testing.assert_close(zd.view(B, Ld).double().cpu(), z.detach(), rtol=1e-5, atol=1e-5)
    klv = ops.kl_forward(M, lvc, Ld, float(B), G.F32)
    torch.testing.assert_close(klv.double().cpu()[0], kl.detach(), rtol=1e-5, atol=1e-4)
    dm = ops.reparam_kl_backward(M, lvc, eps.float().to(DEV), dz.float().to(DEV).view(B, 1, 1, Ld), 0.07 / B, Ld, G.F32)
    torch.testing.assert_close(dm.double().cpu(), mulv.grad, rtol=1e-4, atol=1e-4)


def test_discriminator_head_and_losses(ops):
    B, C, HW = 5, 512, 16
    K = C * HW
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(B, C, 4, 4, generator=g) * 0.3).double().requires_grad_(True)
    w = (torch.randn(1, C, 4, 4, generator=g) * 0.02).double().requires_grad_(True)
    p_ref = torch.sigmoid(F.conv2d(x, w)).view(-1)
    loss_ref = F.binary_cross_entropy(p_ref, torch.full((B,), 0.9, dtype=torch.float64))
    dx_ref, dw_ref = torch.autograd.grad(loss_ref, (x, w))
    gg, pk = G.conv_fprop(B, 4, 4, C, 1, 4, 1, 0, G.F32)
    wp = ops.pack_weights(pk, w.detach().float().to(DEV), G.F32)
    X = to_nhwc(x.detach().float(), C).to(DEV)
    p = ops.dot_sigmoid_forward(X, wp, B, K, G.F32)
    torch.testing.assert_close(p.double().cpu(), p_ref.detach(), rtol=1e-5, atol=1e-6)
    loss = torch.zeros(1, device=DEV)
    dp = ops.bce_forward_backward(p, 0.9, 1.0, loss, False, True)
    torch.testing.assert_close(loss.double().cpu()[0], loss_ref.detach(), rtol=1e-5, atol=1e-6)
    dx, dlogit = ops.dot_sigmoid_backward(p, dp, wp, B, K, G.F32, True, X)
    torch.testing.assert_close(from_nhwc(dx.double().cpu(), C), dx_ref, rtol=1e-4, atol=1e-7)
    dw = torch.zeros(1, C, 4, 4, device=DEV)
    ops.dot_wgrad(X, dlogit, dw, B, K, C, HW, False, G.F32)
    torch.testing.assert_close(dw.double().cpu(), dw_ref, rtol=1e-4, atol=1e-7)
    # BCE clamp at log >= -100 (p == 0 / p == 1)
    pe = torch.tensor([0.0, 1.0, 0.5], device=DEV)
    ops.bce_forward_backward(pe, 0.9, 1.0, loss, False, False)
    ref = F.binary_cross_entropy(pe.cpu(), torch.full((3,), 0.9))
    torch.testing.assert_close(loss.cpu()[0], ref, rtol=1e-6, atol=1e-6)
    # MSE
    a, b = torch.randn(3, 3, 17, 17), torch.randn(3, 3, 17, 17)
    l = torch.zeros(1, device=DEV)
    da = ops.mse_forward_backward(a.to(DEV), b.to(DEV), 1.0, l, True)
    torch.testing.assert_close(l.cpu()[0], F.mse_loss(a, b), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(da.cpu(), 2 * (a - b) / a.numel(), rtol=1e-6, atol=1e-9)


def test_adam_against_torch_optim_golden(ops, golden_dir):
    """G5: torch.optim.Adam trajectory captured in tests/golden/adam_kat.npz."""
    g = np.load(os.path.join(golden_dir, "adam_kat.npz"))
    p = torch.from_numpy(g["p0"].copy()).to(DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    state = torch.zeros(4, device=DEV)
    for i in range(g["grads"].shape[0]):
        gr = torch.from_numpy(g["grads"][i].copy()).to(DEV)
        ops.adam_step(p, gr, m, v, 2e-4, 0.9, 0.999, 1e-8, 1.0, state)
        np.testing.assert_allclose(p.cpu().numpy(), g["traj"][i], rtol=2e-6, atol=1e-8)
    np.testing.assert_allclose(m.cpu().numpy(), g["exp_avg"], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(v.cpu().numpy(), g["exp_avg_sq"], rtol=2e-6, atol=1e-12)
    assert float(state[0]) == g["grads"].shape[0]


@pytest.mark.parametrize("dtype", [G.F32, G.BF16])
def test_tiled_multi_pack_equals_reference_pack(ops, dtype):
    """vg_pack_weights_multi (LDS-tiled, one launch for many operands) == vg_pack_weights, bit for bit, for
    every operand form the networks use (direct, 4-phase transposed, taps-in-N, linear incl. many-tap kernels)."""
    g = torch.Generator().manual_seed(11)
    cases = []
    for fn, a, wshape in [
        (G.conv_fprop, (2, 64, 64, 3, 64, 4, 2, 1), (64, 3, 4, 4)),
        (G.conv_dgrad, (2, 64, 64, 3, 64, 4, 2, 1), (64, 3, 4, 4)),
        (G.conv_fprop, (2, 14, 14, 64, 128, 4, 2, 0), (128, 64, 4, 4)),
        (G.conv_dgrad, (2, 31, 31, 32, 64, 4, 2, 0), (64, 32, 4, 4)),
        (G.convT_fprop, (2, 1, 1, 100, 256, 4, 1, 0), (100, 256, 4, 4)),
        (G.convT_dgrad, (2, 1, 1, 100, 256, 4, 1, 0), (100, 256, 4, 4)),
        (G.convT_fprop, (2, 8, 8, 128, 72, 4, 2, 1), (128, 72, 4, 4)),
        (G.convT_dgrad, (2, 8, 8, 128, 72, 4, 2, 1), (128, 72, 4, 4)),
        (G.convT_fprop, (2, 16, 16, 64, 3, 3, 1, 1), (64, 3, 3, 3)),
        (G.convT_dgrad, (2, 16, 16, 64, 3, 3, 1, 1), (64, 3, 3, 3)),
        (G.conv_fprop, (2, 4, 4, 512, 1, 4, 1, 0), (1, 512, 4, 4)),
    ]:
        _, pk = fn(*a, dtype)
        cases.append((pk, torch.randn(wshape, generator=g).to(DEV)))
    for H, C, N in ((2, 256, 200), (6, 40, 24)):               # 6x6 = 36 taps: more than one tap tile
        w = torch.randn(N, C * H * H, generator=g).to(DEV)
        cases.append((G.linear_fprop(2, H, H, C, N, dtype)[1], w))
        cases.append((G.linear_dgrad(2, H, H, C, N, dtype)[1], w))
    outs = [torch.full((pk.numel(),), 7.0, device=DEV).to(ops.TORCH_DT[dtype]) for pk, _ in cases]
    table, tiles = ops.pack_table([ops.pack_desc(pk, w, o) for (pk, w), o in zip(cases, outs)], DEV)
    ops.pack_weights_multi(table, len(cases), tiles, dtype)
    for (pk, w), o in zip(cases, outs):
        ref = ops.pack_weights(pk, w, dtype)
        assert torch.equal(o, ref), pk


# ---- patch gather-GEMM (csrc/conv_patch.hpp): the tile's input patch resident in LDS, taps read it shifted ----------
PATCH_CASES = [  # kind, B, H (input), Cin, Cout      (k4 s2 p1 everywhere; all give M % 256 == 0)
    ("conv", 8, 32, 64, 128), ("conv", 2, 64, 32, 128), ("conv", 8, 16, 128, 256), ("conv", 4, 32, 96, 160),
    ("convT", 8, 16, 128, 128), ("convT", 2, 32, 64, 128), ("convT", 8, 8, 256, 256), ("convT", 4, 16, 96, 160),
    ("conv_dgrad", 8, 32, 128, 64), ("conv_dgrad", 4, 16, 256, 96),          # transposed form, N = Cin
    ("convT_dgrad", 8, 16, 128, 128), ("convT_dgrad", 4, 8, 160, 64),        # direct stride-2 form, N = Cin
    ("convT", 8, 16, 128, 64), ("conv", 8, 32, 64, 64), ("conv_dgrad", 8, 32, 64, 128), ("convT", 2, 32, 96, 40),  # 128 x 64 tile
    ("convT", 16, 4, 256, 128), ("conv", 16, 8, 128, 256), ("convT_dgrad", 16, 4, 128, 256),   # 4 x 4 grids: 8 images per tile
    ("convT", 2, 64, 32, 128), ("conv", 2, 128, 32, 128),                                     # 64-wide grids
    ("convT", 4, 32, 64, 32), ("conv_dgrad", 4, 64, 32, 64), ("conv", 4, 64, 32, 32), ("convT", 2, 64, 64, 32),   # 128 x 32 tile (S >= 128 layers)
]


@pytest.mark.parametrize("kind,B,H,Cin,Cout", PATCH_CASES)
@pytest.mark.parametrize("variant", ["patch128", "patch256"])
def test_patch_gather_gemm_equals_reference_and_gather_path(ops, vg_switch, kind, B, H, Cin, Cout, variant):
    """Both patch forms (one phase of the 4-phase transposed form; the stride-2 4x4 conv as 4 input-parity classes),
    4-wave 128-row and 8-wave 256-row tiles, against torch fp64 and against the per-tap gather kernel."""
    dtype = G.BF16
    g = torch.Generator().manual_seed(H * 7 + Cin)
    if kind == "conv":
        x = _q(torch.randn(B, Cin, H, H, generator=g), dtype)
        w = torch.randn(Cout, Cin, 4, 4, generator=g) * 0.1
        ref = F.conv2d(x, _q(w, dtype), None, stride=2, padding=1)
        gg, pk = G.conv_fprop(B, H, H, Cin, Cout, 4, 2, 1, dtype)
        nout = Cout
    elif kind == "convT":
        x = _q(torch.randn(B, Cin, H, H, generator=g), dtype)
        w = torch.randn(Cin, Cout, 4, 4, generator=g) * 0.1
        ref = F.conv_transpose2d(x, _q(w, dtype), None, stride=2, padding=1)
        gg, pk = G.convT_fprop(B, H, H, Cin, Cout, 4, 2, 1, dtype)
        nout = Cout
    elif kind == "conv_dgrad":                       # dx of Conv2d(Cin -> Cout) on an H x H input; operand = dy
        x = _q(torch.randn(B, Cout, H // 2, H // 2, generator=g), dtype)
        w = torch.randn(Cout, Cin, 4, 4, generator=g) * 0.1
        ref = torch.nn.grad.conv2d_input((B, Cin, H, H), _q(w, dtype), x, stride=2, padding=1)
        gg, pk = G.conv_dgrad(B, H, H, Cin, Cout, 4, 2, 1, dtype)
        nout = Cin
    else:                                            # dx of ConvTranspose2d(Cin -> Cout) on an H x H input; operand = dy
        x = _q(torch.randn(B, Cout, 2 * H, 2 * H, generator=g), dtype)
        w = torch.randn(Cin, Cout, 4, 4, generator=g) * 0.1
        ref = F.conv2d(x, _q(w, dtype), None, stride=2, padding=1)          # dgrad of convT == conv with the same weights
        gg, pk = G.convT_dgrad(B, H, H, Cin, Cout, 4, 2, 1, dtype)
        nout = Cin
    Wp = ops.pack_weights(pk, w.to(DEV), dtype)
    X = _dev(to_nhwc(x, gg.IC), dtype, ops)
    vg_switch("VG_TILE_MIN_WGS", "1")       # small problems still take the 128 x 128 tile
    vg_switch("VG_GG_PHASE4", "0")          # (the <= 32-column transposed cases would go to conv_phase4.hpp: its own test below)
    outs = {}
    for mode in ("gather", variant):
        vg_switch("VG_GG_PATCH", "0" if mode == "gather" else "1")
        vg_switch("VG_PATCH256_MIN", "1" if mode == "patch256" else "2000000000")
        vg_switch("VG_PATCH256X64_MIN", "1" if mode == "patch256" else "2000000000")    # 256 x 64 for 33..64 channels
        Y, st, nparts = ops.gather_gemm(gg, X, Wp, dtype, want_stats=True)
        M = gg.B * gg.GH * gg.GW
        # 256-row tiles only where an 8-wave variant applies (N > 32 and a patch of <= 384 pixels), else 128-row tiles
        assert nparts in ((gg.nphase * (M // 128),) if mode != "patch256" else (gg.nphase * (M // 128), gg.nphase * (M // 256)))
        outs[mode] = (from_nhwc(Y.double().cpu(), nout), st[: nparts * 2 * nout].view(nparts, 2, nout).double().sum(0).cpu())
    close(outs[variant][0], ref, dtype)
    # same products, different summation order (class/chunk/tap instead of tap/chunk): bf16 output rounding apart
    close(outs[variant][0], outs["gather"][0], dtype)
    torch.testing.assert_close(outs[variant][1], outs["gather"][1], rtol=1e-5, atol=1e-3)


NARROW_CASES = [  # kind, B, H (input), Cin, Cout   (k4 s2 p1; <= 32 output channels of the 4-phase transposed form)
    ("convT", 8, 16, 32, 16), ("convT", 1, 128, 32, 16), ("conv_dgrad", 8, 32, 16, 32), ("convT", 2, 64, 32, 12),
    ("conv_dgrad", 2, 256, 16, 32),                                       # 16 columns, one chunk; 128-wide grid
    ("convT", 2, 64, 64, 32), ("conv_dgrad", 4, 64, 32, 64), ("convT", 4, 32, 64, 32), ("convT", 16, 4, 96, 24),
    ("convT", 16, 8, 32, 32), ("conv_dgrad", 1, 256, 32, 64),             # 32 columns, 1 - 3 chunks; multi-image tiles
]


@pytest.mark.parametrize("kind,B,H,Cin,Cout", NARROW_CASES)
def test_narrow_transposed_layers_all_phases_per_workgroup(ops, vg_switch, kind, B, H, Cin, Cout):
    """The narrow layers of the S >= 128 stacks (ConvTranspose2d(64 -> 32), (32 -> 16) forward, the data gradients of
    Conv2d(16 -> 32), (32 -> 64); gan_code.py:21-49, :61-84 at img_size 128 / 256) on conv_phase4.hpp -- all four sub-pixel
    phases of 256 grid pixels in one workgroup, union patch fetched once, whole output rows written: against torch fp64 and
    against the generic per-tap gather tiles (one phase per workgroup), BatchNorm partial sums included; 4- to 128-wide grids (multi-image tiles, a two-row tile of a 128-wide grid), 12 and 24 real columns."""
    dtype = G.BF16
    g = torch.Generator().manual_seed(H * 5 + Cin)
    if kind == "convT":
        x = _q(torch.randn(B, Cin, H, H, generator=g), dtype)
        w = torch.randn(Cin, Cout, 4, 4, generator=g) * 0.1
        ref = F.conv_transpose2d(x, _q(w, dtype), None, stride=2, padding=1)
        gg, pk = G.convT_fprop(B, H, H, Cin, Cout, 4, 2, 1, dtype)
        nout = Cout
    else:
        x = _q(torch.randn(B, Cout, H // 2, H // 2, generator=g), dtype)
        w = torch.randn(Cout, Cin, 4, 4, generator=g) * 0.1
        ref = torch.nn.grad.conv2d_input((B, Cin, H, H), _q(w, dtype), x, stride=2, padding=1)
        gg, pk = G.conv_dgrad(B, H, H, Cin, Cout, 4, 2, 1, dtype)
        nout = Cin
    assert gg.N <= 32 and gg.nphase == 4
    M = gg.B * gg.GH * gg.GW
    Wp = ops.pack_weights(pk, w.to(DEV), dtype)
    X = _dev(to_nhwc(x, gg.IC), dtype, ops)
    bias = torch.randn(nout, generator=g).to(DEV)
    outs = {}
    vg_switch("VG_GG_PHASE4_MIN", "1")       # (by default only launches of >= 512 tiles come here)
    for name, ph4 in (("generic", "0"), ("phase4", "1")):
        vg_switch("VG_GG_PHASE4", ph4)
        Y, st, nparts = ops.gather_gemm(gg, X, Wp, dtype, bias=bias, want_stats=True)
        if name != "generic":
            assert nparts == gg.nphase * (M // 256)
        assert (Y[..., nout:] == 0).all()
        outs[name] = (from_nhwc(Y.double().cpu(), nout), st[: nparts * 2 * nout].view(nparts, 2, nout).double().sum(0).cpu())
    refb = ref + bias.double().cpu().view(1, -1, 1, 1)
    for name in outs:
        close(outs[name][0], refb, dtype)
        close(outs[name][0], outs["generic"][0], dtype)
        torch.testing.assert_close(outs[name][1], outs["generic"][1], rtol=1e-5, atol=2e-3)
    # the two fused epilogue forms of BatchNorm-less neighbours: LeakyReLU on the way out (forward), and the activation
    # backward of the layer below as a mask on a data gradient (the Discriminator's first conv, gan_code.py:61-62)
    mx = torch.randn(gg.B, gg.OH, gg.OW, gg.OC, generator=g).to(DEV).to(torch.bfloat16)
    fused = {}
    for name, ph4 in (("generic", "0"), ("phase4", "1")):
        vg_switch("VG_GG_PHASE4", ph4)
        Ya, _, _ = ops.gather_gemm(gg, X, Wp, dtype, bias=bias, act=(2, 0.2))
        Ym, _, _ = ops.gather_gemm(gg, X, Wp, dtype, mask=(mx, 2, 0.2))
        fused[name] = (Ya.float().cpu(), Ym.float().cpu())
    ref_act = torch.nn.functional.leaky_relu(refb, 0.2)
    close(from_nhwc(fused["phase4"][0].double(), nout), ref_act, dtype)
    ref_mask = ref * torch.where(from_nhwc(mx.double().cpu(), nout) > 0, 1.0, 0.2)
    close(from_nhwc(fused["phase4"][1].double(), nout), ref_mask, dtype)
    for a_, b_ in zip(fused["phase4"], fused["generic"]):
        torch.testing.assert_close(a_, b_, rtol=3e-2, atol=3e-2 * max(1.0, float(b_.abs().max())))


@pytest.mark.parametrize("kind,B,H,Cin,Cout", [("conv", 8, 16, 64, 128), ("convT", 8, 8, 128, 64), ("convT", 8, 1, 100, 1024),
                                               ("conv", 3, 16, 32, 64)])
def test_workgroup_order_switches_do_not_change_results(ops, vg_switch, kind, B, H, Cin, Cout):
    """The XCD-aware workgroup orders (wgrad: VG_WG_XCD, gather-GEMM: VG_GG_NMAJOR) are pure placement: forced on for
    every launch (=2) and off (=0) must give bit-identical outputs, also where the grid is not a multiple of 8."""
    dtype = G.BF16
    g = torch.Generator().manual_seed(B + H + Cin)
    k, s, p = (4, 1, 0) if H == 1 else (4, 2, 1)
    a = (B, H, H, Cin, Cout, k, s, p, dtype)
    if kind == "conv":
        gg, pk = G.conv_fprop(*a)
        ggd, pkd = G.conv_dgrad(*a)
        wg = G.conv_wgrad(*a)
        w = torch.randn(Cout, Cin, k, k, generator=g).to(DEV) * 0.1
    else:
        gg, pk = G.convT_fprop(*a)
        ggd, pkd = G.convT_dgrad(*a)
        wg = G.convT_wgrad(*a)
        w = torch.randn(Cin, Cout, k, k, generator=g).to(DEV) * 0.1
    X = torch.randn(gg.B, gg.IH, gg.IW, gg.IC, generator=g).to(DEV).to(torch.bfloat16)
    DY = torch.randn(ggd.B, ggd.IH, ggd.IW, ggd.IC, generator=g).to(DEV).to(torch.bfloat16)
    Wp, Wd = ops.pack_weights(pk, w, dtype), ops.pack_weights(pkd, w, dtype)
    res = {}
    for mode in ("0", "2"):
        vg_switch("VG_WG_XCD", mode)
        vg_switch("VG_GG_NMAJOR", mode)
        Y, _, _ = ops.gather_gemm(gg, X, Wp, dtype)
        DX, _, _ = ops.gather_gemm(ggd, DY, Wd, dtype)
        dW = torch.zeros(w.shape, device=DEV)
        P, Q = (X, DY) if kind == "convT" else (DY, X)
        ops.wgrad(wg, P, Q, dW, False, dtype)
        res[mode] = (Y.clone(), DX.clone(), dW)
    for a_, b_ in zip(res["0"], res["2"]):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("rpg,C,groups,act,slope", [(2048, 512, 2, 2, 0.2), (4096, 256, 1, 1, 0.0), (512, 1024, 1, 1, 0.0), (4608, 128, 1, 2, 0.01)])
def test_bn_finalize_and_forward_in_one_launch(ops, vg_switch, rpg, C, groups, act, slope):
    """bf16 train-mode BatchNorm with a small statistics slab: vg_bn_finalize_act_forward (every workgroup re-derives its
    64 channels' coefficients) against vg_bn_finalize_grouped + vg_bn_act_forward -- coefficients, running statistics
    (updated group after group) and the activated output."""
    dt = G.BF16
    rows = rpg * groups
    g = torch.Generator().manual_seed(rpg + C)
    x = (torch.randn(rows, C, generator=g) * 1.5 + 0.4).to(DEV).to(torch.bfloat16)
    gamma = (torch.randn(C, generator=g) * 0.1 + 1).to(DEV)
    beta = (torch.randn(C, generator=g) * 0.1).to(DEV)
    stats, nparts = [], 0
    for k in range(groups):                                   # one slab block per group, as the grouped forward emits them
        st, n = ops.channel_stats(x[k * rpg:(k + 1) * rpg], rpg, C, dt)
        stats.append(st[: n * 2 * C].clone())
        nparts += n
    stats = torch.cat(stats)
    assert nparts // groups <= 200
    rm1, rv1 = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    co1 = ops.bn_finalize(stats, nparts, C, rows, gamma, beta, rm1, rv1, 0.1, 1e-5, DEV, groups=groups)
    y1 = ops.bn_act_forward(x, co1, rows, C, act, slope, dt)
    rm2, rv2 = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    res = ops.bn_finalize_act_forward(x, stats, nparts, C, rows, gamma, beta, rm2, rv2, 0.1, 1e-5, act, slope, dt, groups=groups)
    assert res is not None
    co2, y2 = res
    torch.cuda.synchronize()
    torch.testing.assert_close(co2, co1, rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(rm2, rm1, rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(rv2, rv1, rtol=2e-6, atol=1e-7)
    d = (y2.float() - y1.float()).abs()
    assert float(d.max()) <= 2.0 ** -7 * float(y1.float().abs().max()) and float((d > 0).float().mean()) < 1e-3
    # reference semantics: torch batch_norm in fp64 on the same bf16 input, per group
    for k in range(groups):
        xs = x[k * rpg:(k + 1) * rpg].double().cpu().view(rpg, C, 1, 1)
        z = F.batch_norm(xs, None, None, gamma.double().cpu(), beta.double().cpu(), True, 0.1, 1e-5)
        ref = (F.leaky_relu(z, slope) if act == 2 else F.relu(z)).view(rpg, C)
        torch.testing.assert_close(y2[k * rpg:(k + 1) * rpg].double().cpu(), ref, **TOL[dt])
    vg_switch("VG_BN_FUSED_FWD", "0")
    assert ops.bn_finalize_act_forward(x, stats, nparts, C, rows, gamma, beta, rm2, rv2, 0.1, 1e-5, act, slope, dt, groups=groups) is None
    assert ops.bn_finalize_act_forward(x.float(), stats, nparts, C, rows, gamma, beta, rm2, rv2, 0.1, 1e-5, act, slope, G.F32, groups=groups) is None


@pytest.mark.parametrize("rpg,C,groups,act,slope", [(2048, 512, 2, 2, 0.2), (4096, 256, 1, 1, 0.0), (4608, 128, 1, 2, 0.01)])
def test_bn_backward_finalize_and_apply_in_one_launch(ops, vg_switch, rpg, C, groups, act, slope):
    """bf16 BatchNorm backward of a small layer: vg_bn_backward_finalize_apply (through ops.bn_act_backward) against the
    grouped finalize + apply it replaces (VG_BN_FUSED_FWD=0): dx, and dgamma / dbeta accumulated onto old values in
    group order."""
    dt = G.BF16
    vg_switch("VG_BN_ONEPASS", "0")          # (the one-launch form of round 4 would take these shapes: its own test below)
    rows = rpg * groups
    g = torch.Generator().manual_seed(rpg * 3 + C)
    x = (torch.randn(rows, C, generator=g) * 1.3 + 0.2).to(DEV).to(torch.bfloat16)
    dy = torch.randn(rows, C, generator=g).to(DEV).to(torch.bfloat16)
    gamma = (torch.randn(C, generator=g) * 0.1 + 1).to(DEV)
    beta = (torch.randn(C, generator=g) * 0.1).to(DEV)
    xs = x.float().view(groups, rpg, C)
    mean, var = xs.mean(1), xs.var(1, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale = gamma * invstd
    co = torch.stack([mean, invstd, scale, beta - mean * scale], 1).contiguous()
    seed_g, seed_b = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
    out = {}
    for mode in ("0", "1"):
        vg_switch("VG_BN_FUSED_FWD", mode)
        dg, db = seed_g.clone(), seed_b.clone()
        dx = ops.bn_act_backward(x, dy, co, rows, C, rows, gamma, act, slope, dg, db, True, dt)
        out[mode] = (dx, dg, db)
    torch.cuda.synchronize()
    torch.testing.assert_close(out["1"][1], out["0"][1], rtol=2e-5, atol=2e-4)
    torch.testing.assert_close(out["1"][2], out["0"][2], rtol=2e-5, atol=2e-4)
    d = (out["1"][0].float() - out["0"][0].float()).abs()
    assert float(d.max()) <= 2.0 ** -7 * float(out["0"][0].float().abs().max()) and float((d > 0).float().mean()) < 1e-3
    # against torch autograd in fp64 (first group)
    xr = x[:rpg].double().cpu().view(rpg, C, 1, 1).requires_grad_(True)
    z = F.batch_norm(xr, None, None, gamma.double().cpu(), beta.double().cpu(), True, 0.1, 1e-5)
    a = F.leaky_relu(z, slope) if act == 2 else F.relu(z)
    (dx_ref,) = torch.autograd.grad(a, xr, dy[:rpg].double().cpu().view(rpg, C, 1, 1))
    torch.testing.assert_close(out["1"][0][:rpg].double().cpu(), dx_ref.view(rpg, C), rtol=5e-2, atol=5e-2)


@pytest.mark.parametrize("rpg,C,groups,act,slope,accumulate",
                         [(2048, 512, 2, 2, 0.2, True), (4096, 256, 1, 1, 0.0, False), (4608, 128, 1, 2, 0.01, True),
                          (8 * 31 * 31, 32, 1, 2, 0.01, False), (35, 256, 1, 2, 0.2, True), (32, 1024, 1, 1, 0.0, False),
                          (128 * 16 * 16, 256, 1, 1, 0.0, True), (128 * 16 * 16, 128, 2, 2, 0.2, False),
                          (128 * 31 * 31, 32, 1, 2, 0.01, True), (3, 8, 1, 0, 0.0, False)])
def test_bn_backward_in_one_launch_with_a_grid_wide_exchange(ops, vg_switch, rpg, C, groups, act, slope, accumulate):
    """csrc/bn_onepass.hip (round 4): column sums, coefficients and dx in ONE launch, the x / dy rows of a workgroup held in
    registers across a grid-wide exchange of the partial sums -- against the column-reduce -> finalize -> apply launches it
    replaces (VG_BN_ONEPASS=0): dx equal up to rare one-ulp bf16 flips (the double sums associate differently), dgamma /
    dbeta to fp32 rounding; called three times in a row (the grid counters must return to zero), from full blocks (K = 8,
    256 workgroups) down to a ragged 3-row tensor; the bounded wait never gives up."""
    dt = G.BF16
    rows = rpg * groups
    g = torch.Generator().manual_seed(rpg * 5 + C)
    x = (torch.randn(rows, C, generator=g) * 1.3 + 0.2).to(DEV).to(torch.bfloat16)
    dy = torch.randn(rows, C, generator=g).to(DEV).to(torch.bfloat16)
    gamma = (torch.randn(C, generator=g) * 0.1 + 1).to(DEV)
    beta = (torch.randn(C, generator=g) * 0.1).to(DEV)
    xs = x.float().view(groups, rpg, C)
    mean, var = xs.mean(1), xs.var(1, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale = gamma * invstd
    co = torch.stack([mean, invstd, scale, beta - mean * scale], 1).contiguous()
    seed_g, seed_b = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
    vg_switch("VG_BN_ONEPASS", "1")
    assert importlib.import_module(PKG + "._lib").load().vg_bn_backward_onepass_supported(rows, C, groups, dt) == 1
    out = {}
    for mode in ("0", "1", "1", "1"):
        vg_switch("VG_BN_ONEPASS", mode)
        dg, db = seed_g.clone(), seed_b.clone()
        n0 = ops.launch_count()
        dx = ops.bn_act_backward(x, dy, co, rows, C, rows, gamma, act, slope, dg, db, accumulate, dt)
        assert (ops.launch_count() - n0 == 1) == (mode == "1")
        torch.cuda.synchronize()
        if mode == "1" and "1" in out:
            assert torch.equal(dx, out["1"][0]) and torch.equal(dg, out["1"][1]) and torch.equal(db, out["1"][2])   # deterministic
        out[mode] = (dx, dg, db)
    assert not ops.grid_sync_error(x.device)
    torch.testing.assert_close(out["1"][1], out["0"][1], rtol=2e-5, atol=2e-4)
    torch.testing.assert_close(out["1"][2], out["0"][2], rtol=2e-5, atol=2e-4)
    d = (out["1"][0].float() - out["0"][0].float()).abs()
    assert float(d.max()) <= 2.0 ** -7 * float(out["0"][0].float().abs().max()) and float((d > 0).float().mean()) < 1e-3
    xr = x[:rpg].double().cpu().view(rpg, C, 1, 1).requires_grad_(True)
    z = F.batch_norm(xr, None, None, gamma.double().cpu(), beta.double().cpu(), True, 0.1, 1e-5)
    a = F.leaky_relu(z, slope) if act == 2 else (F.relu(z) if act == 1 else z)
    (dx_ref,) = torch.autograd.grad(a, xr, dy[:rpg].double().cpu().view(rpg, C, 1, 1))
    torch.testing.assert_close(out["1"][0][:rpg].double().cpu(), dx_ref.view(rpg, C), rtol=5e-2, atol=5e-2)


def test_bn_backward_in_one_launch_replays_from_a_hipgraph(ops, vg_switch):
    """The grid counters of bn_onepass.hip go back to zero inside every launch (no memset node): a captured launch replays
    any number of times with the results of the eager call."""
    vg_switch("VG_BN_ONEPASS", "1")
    dt, rows, C = G.BF16, 128 * 8 * 8, 512
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(rows, C, generator=g) + 0.1).to(DEV).to(torch.bfloat16)
    dy = torch.randn(rows, C, generator=g).to(DEV).to(torch.bfloat16)
    gamma, beta = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    mean, var = x.float().mean(0), x.float().var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    co = torch.stack([mean, invstd, gamma * invstd, beta - mean * gamma * invstd], 0).unsqueeze(0).contiguous()
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    ref = ops.bn_act_backward(x, dy, co, rows, C, rows, gamma, 1, 0.0, dg, db, False, dt).clone()
    ref_g = dg.clone()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        graph.capture_begin()
        out = ops.bn_act_backward(x, dy, co, rows, C, rows, gamma, 1, 0.0, dg, db, False, dt)
        graph.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(5):
        out.zero_(), dg.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref) and torch.equal(dg, ref_g)
    assert not ops.grid_sync_error(x.device)


@pytest.mark.parametrize("B", [1, 37, 128, 300])
def test_pair_bce_is_two_bce_launches_bit_for_bit(ops, B):
    """vg_bce_pair_forward_backward (the grouped Discriminator pass) against two vg_bce_forward_backward launches, the
    second accumulating: same loss bits, same gradient bits."""
    g = torch.Generator().manual_seed(B)
    p = torch.rand(2 * B, generator=g).clamp(1e-4, 1 - 1e-4).to(DEV)
    p[0] = 0.0                                        # log clamp
    l1, l2 = torch.zeros(1, device=DEV), torch.full((1,), 7.0, device=DEV)
    d1, d2 = torch.empty_like(p), torch.empty_like(p)
    ops.bce_forward_backward(p[:B], 1.0, 0.5, l1, False, True, out=d1[:B])
    ops.bce_forward_backward(p[B:], 0.0, 0.5, l1, True, True, out=d1[B:])
    ops.bce_pair_forward_backward(p, 1.0, 0.0, 0.5, l2, d2)
    torch.cuda.synchronize()
    assert torch.equal(l1, l2) and torch.equal(d1, d2)

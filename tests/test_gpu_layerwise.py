"""GPU: the bf16 engine's training iteration checked LAYER BY LAYER against exact arithmetic on the engine's OWN stored
tensors (round-3 review weak #1 / round-4 item 2).

Why not end to end: two faithful executions of the same bf16-storage arithmetic do not stay together.  A one-ulp rounding
flip of a stored activation (the engine accumulates in f32, the emulation oracle/vaegan_ref_bf16.py in f64: 1.6e-5 of the
first layer's outputs round the other way) changes thousands of downstream sums, which flip more roundings: measured
(tools/bf16_layer_diff.py, profiles/r04_bf16_layer_diff.txt) 1.6e-5 -> 2.5e-4 -> 4.6e-3 -> 6 % of the elements per Encoder
layer and 33 % ... 72 % through the Generator, i.e. deep activations of engine and emulation differ by the bf16 rounding
noise itself (relative Frobenius 6e-3), and their gradients by 2e-3 ... 1.7e-1 -- as far from each other as each is from
the fp64 oracle (profiles/r04_bf16_engine_vs_emulation.txt).  An end-to-end bound tighter than the bf16 noise floor
cannot hold for ANY correct implementation.  What CAN be held tight is every kernel group inside the real iteration: given
the tensors the engine itself stored as a stage's inputs, its stored outputs must equal exact arithmetic + one bf16
rounding, up to rare one-ulp flips.  An indexing or scheduling defect that corrupts even 1 % of one stage's output in the
benchmarked dtype, batch and tile shapes fails these bounds by orders of magnitude.

vaegan_code.py:74-135 (one iteration, S=64, B=128, bf16, eager launches with injected noise)."""
import importlib
import os

import pytest
import torch
import torch.nn.functional as F

from _inputs import make_inputs

import vaegan_amd as V
from test_gpu_parity import DEV, build

pytestmark = pytest.mark.gpu
torch.set_num_threads(min(32, os.cpu_count() or 1))

FLIP_TOL = 3e-4         # relative Frobenius distance of a bf16 tensor from bf16(exact): one-ulp flips of <= ~0.5 % of the elements
F32_TOL = 2e-4          # f32 results (weight gradients, the reconstruction): f32 accumulation against f64


def q(t):
    return t.to(torch.bfloat16).to(torch.float64)


def frob(a, r):
    return float((a.double() - r.double()).norm() / r.double().norm().clamp_min(1e-30))


def nchw(t, C):
    """[B,H,W,CP] engine tensor -> [B,C,H,W] float64 on the CPU."""
    return t[..., :C].permute(0, 3, 1, 2).double().cpu().contiguous()


def stage_fn(st, w, b, x):
    if st.kind == "conv":
        return F.conv2d(x, w, b, stride=st.s, padding=st.p)
    if st.kind == "convT":
        return F.conv_transpose2d(x, w, b, stride=st.s, padding=st.p)
    raise AssertionError(st.kind)


def act(z, st):
    code = st.act
    return F.leaky_relu(z, st.slope) if code == 2 else (F.relu(z) if code == 1 else z)


def stage_weights(st):
    if st.kind == "linear2":
        w = torch.cat([st.conv.weight.detach(), st.conv2.weight.detach()], 0)
        b = torch.cat([st.conv.bias.detach(), st.conv2.bias.detach()], 0)
        return q(w.cpu()), b.double().cpu()
    b = st.conv.bias.detach().double().cpu() if getattr(st.conv, "bias", None) is not None else None
    return q(st.conv.weight.detach().cpu()), b


def check_engine(name, eng, report):
    from_fwd = {}
    prev_gw = {}
    for rec in eng.trace:
        i, what = rec["stage"], rec["what"]
        st = eng.stages[i]
        tag = f"{name}.{i} {what}"
        if what == "fwd":
            w, b = stage_weights(st)
            if st.kind == "linear2":
                x = nchw(rec["x"], st.cin).flatten(1)
                yr = F.linear(x, w, b)
                ye = rec["Y"].double().cpu().reshape(x.shape[0], -1)[:, :st.cout]
            else:
                x = nchw(rec["x"], st.cin)
                yr = stage_fn(st, w, b, x)
                ye = nchw(rec["Y"], st.cout)
            if rec["fused_act"]:
                yr = act(yr, st)
            report(tag + " Y", frob(ye, q(yr)), FLIP_TOL)
            if rec["coeffs"] is not None and st.bn is not None:
                co = rec["coeffs"].double().cpu()                       # [groups][4][C]
                G_ = co.shape[0]
                yg = yr.reshape(G_, -1, *yr.shape[1:])                  # statistics: from the UNROUNDED conv output, per group
                mean = yg.mean(dim=(1, 3, 4)) if yg.dim() == 5 else yg.mean(dim=1)
                var = yg.var(dim=(1, 3, 4), unbiased=False) if yg.dim() == 5 else yg.var(dim=1, unbiased=False)
                if eng.spec(i, rec["B"], "fprop")[1].tap_in_n:          # (1x1-input layer: statistics of the STORED tensor)
                    yq = ye.reshape(G_, -1, *ye.shape[1:])
                    mean, var = yq.mean(dim=(1, 3, 4)), yq.var(dim=(1, 3, 4), unbiased=False)
                report(tag + " mean", float((co[:, 0] - mean).abs().max() / mean.abs().max().clamp_min(1e-6)), 1e-4)
                report(tag + " invstd", frob(co[:, 1], torch.rsqrt(var + 1e-5)), 1e-4)
                # normalise + activation, teacher-forced on the engine's stored Y and published coefficients
                sc = co[:, 2].reshape(G_, 1, -1, 1, 1)
                sh = co[:, 3].reshape(G_, 1, -1, 1, 1)
                ar = act(sc * ye.reshape(G_, -1, *ye.shape[1:]) + sh, st).reshape(ye.shape)
                report(tag + " A", frob(nchw(rec["A"], st.cout), q(ar)), FLIP_TOL)
        elif what == "fwd_tn":
            w, _ = stage_weights(st)
            x = nchw(rec["x"], st.cin)
            yr = stage_fn(st, w, None, x)
            if rec["A"].dim() == 4 and rec["A"].shape[1] == st.cout and rec["A"].dtype == torch.float32:
                report(tag + " tanh image", frob(rec["A"].double().cpu(), torch.tanh(yr)), F32_TOL)
        elif what == "bn_bwd" and st.bn is not None:
            co = rec["coeffs"].double().cpu()
            G_ = co.shape[0]
            Y = nchw(rec["Y"], st.cout)
            dA = nchw(rec["dA"], st.cout)
            Yg, dAg = Y.reshape(G_, -1, *Y.shape[1:]), dA.reshape(G_, -1, *Y.shape[1:])
            mean, invstd = co[:, 0].reshape(G_, 1, -1, 1, 1), co[:, 1].reshape(G_, 1, -1, 1, 1)
            z = co[:, 2].reshape(G_, 1, -1, 1, 1) * Yg + co[:, 3].reshape(G_, 1, -1, 1, 1)
            slope = st.slope if st.act == 2 else (0.0 if st.act == 1 else 1.0)
            dz = torch.where(z > 0, dAg, dAg * slope)
            xh = (Yg - mean) * invstd
            n = Yg.shape[1] * Yg.shape[3] * Yg.shape[4]
            s1 = dz.sum(dim=(1, 3, 4), keepdim=True)
            s2 = (dz * xh).sum(dim=(1, 3, 4), keepdim=True)
            a = st.bn.weight.detach().double().cpu().reshape(1, 1, -1, 1, 1) * invstd
            dyr = (a * (dz - s1 / n - xh * s2 / n)).reshape(Y.shape)
            report(tag + " dY", frob(nchw(rec["dY"], st.cout), q(dyr)), FLIP_TOL)
            from_fwd[(i, "dgamma")] = s2.sum(0).flatten()
            from_fwd[(i, "dbeta")] = s1.sum(0).flatten()
        elif what in ("wgrad", "dgrad"):
            w, _ = stage_weights(st)
            if what == "wgrad":
                if st.kind == "linear2":
                    x = nchw(rec["x"], st.cin).flatten(1)
                    dy = rec["dY"].double().cpu().reshape(x.shape[0], -1)[:, :st.cout]
                    gr = dy.t() @ x
                    ge = torch.cat([rec["gw"], rec["gw2"]], 0).double().cpu()
                else:
                    x = nchw(rec["x"], st.cin).requires_grad_(False)
                    wv = w.clone().requires_grad_(True)
                    out = stage_fn(st, wv, None, x)
                    out.backward(nchw(rec["dY"], st.cout))
                    gr, ge = wv.grad, rec["gw"].double().cpu()
                if rec["acc"]:
                    gr = gr + prev_gw[i]
                prev_gw[i] = ge
                report(tag + " dW", frob(ge, gr), F32_TOL if not rec["acc"] else 5e-4)
                if (i, "dgamma") in from_fwd and st.bn is not None and not rec["acc"] and name != "D":
                    report(tag + " dgamma", frob(st.bn.weight.grad.double().cpu(), from_fwd[(i, "dgamma")]), F32_TOL)
                    report(tag + " dbeta", frob(st.bn.bias.grad.double().cpu(), from_fwd[(i, "dbeta")]), F32_TOL)
            else:
                if st.kind == "linear2":
                    dy = rec["dY"].double().cpu().reshape(rec["dY"].shape[0], -1)[:, :st.cout]
                    dxr = (dy @ w).reshape(dy.shape[0], st.cin, st.hin, st.hin)
                else:
                    xz = torch.zeros(rec["dX"].shape[0], st.cin, st.hin, st.hin, dtype=torch.float64, requires_grad=True)
                    out = stage_fn(st, w, None, xz)
                    out.backward(nchw(rec["dY"], st.cout))
                    dxr = xz.grad
                if rec["mask"] is not None:
                    # the activation backward of the BatchNorm-less stage below, fused into this launch's epilogue: it
                    # multiplies the tile AFTER its bf16 rounding (conv_gemm.hip mask_segment) -- two roundings where the slope applies
                    ym, mact, mslope = rec["mask"]
                    ymf = nchw(ym, st.cin)
                    dxr = q(dxr)
                    dxr = torch.where(ymf > 0, dxr, dxr * (mslope if mact == 2 else 0.0))
                report(tag + " dX", frob(nchw(rec["dX"], st.cin), q(dxr)), FLIP_TOL)


def test_every_stage_of_the_bf16_iteration_equals_exact_arithmetic_on_its_own_stored_inputs():
    S, B = 64, 128
    e, g, d, tr = build(S, dtype="bf16", lr=0.0)       # lr = 0: the weights every pass used are the ones read back below
    for m in (e, g, d):
        m._engine.trace = []
    dev_in = [t.to(DEV) for t in make_inputs(B, S, 1234)]
    tr.train_step(dev_in[0], 60, *dev_in[1:])
    torch.cuda.synchronize()
    lines, worst = [], []

    def report(tag, err, tol):
        lines.append(f"{tag:34s} {err:.2e} (bound {tol:.0e})")
        if not err <= tol:
            worst.append(lines[-1])

    try:
        for name, m in (("E", e), ("G", g), ("D", d)):
            check_engine(name, m._engine, report)
    finally:
        for m in (e, g, d):
            m._engine.trace = None
    print("\n".join(lines))
    assert len(lines) >= 120, len(lines)
    assert not worst, "stages outside their bound:\n" + "\n".join(worst)

"""GPU: the configurations BASELINE.json names, at the sizes it quotes them (round-2 VERDICT item 1).

  C2  S=64,  B=128, bf16  -- the driver-timed configuration: EVERY parameter gradient of one iteration, executed by the
                             replayed graph, against the fp64 oracle (the first end-to-end check of the Encoder's and the
                             Generator's backward passes in the benchmarked dtype and at the benchmarked tile shapes);
  C0  S=256 reference geometry, bf16 -- the state after iteration 1 against the REFERENCE's own checksums;
  C5  S=256, B=32, bf16 and fp8 forward -- first-iteration losses at the quoted size;
  C4  S=64,  B=128 denoise evaluation, fp32 and bf16.

bf16 tolerances are stated per quantity next to the assertion that uses them; each was measured on MI355X first
(tools/calibrate_r03.py prints the same numbers) and is set at about twice the worst value seen.
"""
import os

import numpy as np
import pytest
import torch

import vaegan_ref as R
import vaegan_ref_bf16 as RB
from _inputs import make_inputs

import vaegan_amd as V
from test_gpu_parity import (DEV, build, check_state_after_first_iteration, rel, sync_from_oracle)

pytestmark = pytest.mark.gpu
torch.set_num_threads(min(16, os.cpu_count() or 1))


def frob(a, ref):
    return float((a - ref).norm() / ref.norm().clamp_min(1e-300))


def oracle_grads(S, B, seed, double):
    o = R.RefVAEGAN(img_size=S, seed=42, lr=0.0)
    if double:
        o.double_()
    o.train_step(*make_inputs(B, S, seed), 60)
    return {f"{n}.{k}": st[k].grad.double() for n, st in (("E", o.E), ("G", o.G), ("D", o.D)) for k in R.trainable_keys(st)}


# bf16 storage keeps 8 significant bits.  Round 4 replaced the argued explanation of round 3 by a measured one
# (oracle/vaegan_ref_bf16.py: exact arithmetic + bf16 rounding at the engine's storage points; tools/bf16_ablation.py,
# profiles/r04_bf16_ablation.txt): the 0.4 % ... 20 % distance of the bf16 gradients from the fp64 oracle at the seed-42
# initial point is reproduced by that emulation tensor by tensor (G.main.0.weight 1.99e-1 against the engine's 2.00e-1),
# it comes from the FORWARD storage roundings together (forward exact: 7e-3; weights / images / raw outputs / activations
# exact one class at a time: 1.9e-1 ... 2.0e-1) and NOT from any stored gradient (backward exact: unchanged 2.1e-1; any
# single dX / dY exact: unchanged) -- at initialisation the Discriminator's output is nearly the same for every sample, so
# the generator-side gradient is a small difference of large terms that every forward perturbation moves.  Bounds: per
# tensor the engine may be at most 1.5 x as far from fp64 as the emulation of the arithmetic it was asked to do (measured
# <= 1.24 x, tools/calibrate_r04.py) -- or 4 x the CPU-fp32 oracle's own distance where that is larger --, with the absolute
# backstop 3e-1 and the direction (cosine) within 0.95 of the true gradient's.  The kernel-level correctness bound of
# the same iteration is tests/test_gpu_layerwise.py (every stage against exact arithmetic on its own stored inputs, <= 3e-4).
BF16_GRAD_TOL, BF16_GRAD_COS, BF16_VS_EMULATION = 3e-1, 0.95, 1.5


def test_bf16_gradients_of_the_benchmarked_configuration_replayed_graph_vs_fp64_oracle():
    """C2 (S=64, B=128, bf16, hipGraph replay), lr = 0: the gradients the third call leaves in the optimizers' flat
    buffers were produced by the REPLAYED graph (call 1 eager, call 2 capture + replay, call 3 replay) -- 256x128 /
    256x64 / 128x64 patch tiles, the wave-specialised weight-gradient kernel, grouped 2B-row Discriminator passes.
    Bound per tensor: max(BF16_GRAD_TOL, 4 x the CPU-fp32 oracle's own distance from fp64), as the fp32 test
    (test_all_parameter_gradients_vs_fp64_oracle) calibrates it.  vaegan_code.py:95-135."""
    S, B, seed = 64, 128, 1234
    e, g, d, tr = build(S, dtype="bf16", lr=0.0)
    dev_in = [t.to(DEV) for t in make_inputs(B, S, seed)]
    for _ in range(3):
        tr.train_step_graphed(dev_in[0], 60, *dev_in[1:])
    assert tr._graph is not None and len(tr._graph[1]) == 1
    torch.cuda.synchronize()
    hip = {f"{n}.{k}": p.grad.double().cpu() for n, m in (("E", e), ("G", g), ("D", d)) for k, p in m.named_parameters()}
    g32, g64 = oracle_grads(S, B, seed, False), oracle_grads(S, B, seed, True)
    assert sorted(hip) == sorted(g64)
    em = RB.RefVAEGANbf16(img_size=S, seed=42, lr=0.0)
    em.train_step(*make_inputs(B, S, seed), 60)
    gem = {f"{n}.{k}": st[k].grad.double() for n, st in (("E", em.E), ("G", em.G), ("D", em.D)) for k in R.trainable_keys(st)}
    worst, report = 0.0, []
    for k, r in g64.items():
        if float(r.abs().max()) < 1e-6:
            # conv bias in front of BatchNorm: exactly zero in exact arithmetic, rounding noise in every implementation
            assert float(hip[k].abs().max()) < 1e-3, k
            continue
        err, cal, emu = frob(hip[k], r), frob(g32[k], r), frob(gem[k], r)
        cos = float((hip[k] * r).sum() / (hip[k].norm() * r.norm()))
        report.append((err, cal, k))
        assert err <= max(BF16_VS_EMULATION * emu, 4 * cal, 1e-3), \
            f"{k}: bf16 engine {err:.2e} from fp64, the bf16-storage emulation of the same arithmetic only {emu:.2e}"
        worst = max(worst, err)
        assert cos >= BF16_GRAD_COS, f"{k}: bf16 gradient points {cos:.3f} (cosine) from the fp64 oracle's"
    report.sort(reverse=True)
    print("bf16 S=64 B=128 replayed-graph gradients vs fp64 oracle, relative Frobenius error (worst five):",
          [(k, f"{err:.1e}", f"cpu-fp32 {cal:.1e}") for err, cal, k in report[:5]])
    for err, cal, k in report:
        assert err <= max(BF16_GRAD_TOL, 4 * cal), f"{k}: bf16 engine {err:.2e} from fp64 (cpu-fp32: {cal:.1e})"


# State after ONE bf16 iteration against the reference's fp32 checksums (tests/golden/steps_S256_B4_e0.npz: B = 4, where
# BatchNorm backward over four images is close to its most ill-conditioned, on top of the amplification described above).
# Measured on MI355X (tools/calibrate_r03.py): Adam first moments (= 0.1 x gradient) up to 1.9e-1 and second moments
# (quadratic in the gradient) up to 6.2e-1 in abs-sum / sum of squares per tensor -- both worst on G.main.13.bias, five
# BatchNorm backward passes deep --, BatchNorm running statistics 9.5e-4, 9.4 % of one tensor's weights taking their +-lr
# Adam(t=1) step the other way (gradients whose sign is decided below bf16 resolution).  Stated bf16 bounds: 3.5e-1 for
# the first moments, 2.5 x that for the second, 3e-3 for the running statistics, 15 % flipped steps.
BF16_STATE_TOLS = (3.5e-1, 3e-3, 0.15)


def test_bf16_state_after_first_iteration_vs_reference_checksums_S256(golden_dir):
    gold = np.load(os.path.join(golden_dir, "steps_S256_B4_e0.npz"))
    names = [str(n) for n in gold["loss_names"]]
    e, g, d, tr = build(256, dtype="bf16")
    real, ez, er, ec = make_inputs(4, 256, 5000 + 10 * 4 + 0)
    got = tr.loss_dict(tr.train_step(real.to(DEV), 0, ez.to(DEV), er.to(DEV), ec.to(DEV)), 0)
    for j, n in enumerate(names):
        ref = float(gold["losses"][0][j])
        assert rel(got[n], ref) <= 3e-2, f"bf16 S=256 B=4 {n}: hip {got[n]} reference {ref}"
    o64 = R.RefVAEGAN(img_size=256, seed=42).double_()
    o64.train_step(real, ez, er, ec, 0)
    check_state_after_first_iteration(gold, e, g, d, tr, o64, tols=BF16_STATE_TOLS)


@pytest.mark.parametrize("dtype", ["bf16", "fp8"])
def test_first_iteration_at_the_quoted_fp8_configuration_S256_B32(dtype):
    """BASELINE configs[4] per GPU (S=256, B=32).  bf16 against the live oracle (stated bf16 bound 3e-2); fp8 forward
    against the bf16 engine (no reference semantics for fp8: statistical bound 10 %).  Executed by the replayed graph."""
    S, B = 256, 32
    real, ez, er, ec = make_inputs(B, S, 1234)
    dev_in = [t.to(DEV) for t in (real, ez, er, ec)]

    def first_iteration(dt):
        e, g, d, tr = build(S, dtype=dt)
        start = tr.state_dict()
        start = {k: _deep_clone(v) for k, v in start.items()}
        tr.train_step_graphed(dev_in[0], 60, *dev_in[1:])
        tr.train_step_graphed(dev_in[0], 60, *dev_in[1:])
        tr.load_state_dict(start)
        out = tr.loss_dict(tr.train_step_graphed(dev_in[0], 60, *dev_in[1:]).clone(), 60)
        for p in list(e.parameters()) + list(g.parameters()) + list(d.parameters()):
            assert bool(torch.isfinite(p).all())
        return out

    got = first_iteration(dtype)
    if dtype == "bf16":
        ref = R.RefVAEGAN(img_size=S, seed=42).train_step(real, ez, er, ec, 60)
        tol = 3e-2
    else:
        ref = first_iteration("bf16")
        tol = 0.10
    print(f"S=256 B=32 {dtype}:", {n: f"{rel(got[n], ref[n]):.1e}" for n in V.LOSS_NAMES})
    for n in V.LOSS_NAMES:
        assert rel(got[n], ref[n]) <= tol, f"S=256 B=32 {dtype} {n}: {got[n]} vs {ref[n]} (tol {tol})"


def _deep_clone(x):
    if isinstance(x, torch.Tensor):
        return x.detach().clone()
    if isinstance(x, dict):
        return {k: _deep_clone(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_deep_clone(v) for v in x)
    return x


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_denoise_eval_at_the_quoted_configuration_S64_B128(dtype):
    """BASELINE configs[3]: S=64, B=128, sigma=0.2, eval-mode E -> reparam -> G, MSE + KL(sum), PSNR / SSIM
    against the oracle (vaegan_code.py:147-171).  fp32: the bounds of the small-batch test.  bf16 (stated): the
    reconstruction within 3e-2 absolute of the fp32 oracle's (tanh output, |x| <= 1), losses within 3e-2, PSNR within
    0.2 dB, SSIM within 2e-2."""
    S, B, sigma = 64, 128, 0.2
    e, g, d, tr = build(S, dtype=dtype)
    o = R.RefVAEGAN(img_size=S, seed=42)
    real, ez, er, ec = make_inputs(B, S, 7000 + S)
    o.train_step(real, ez, er, ec, 60)                        # one training step so BN running stats are non-trivial
    tr.train_step(real.to(DEV), 60, ez.to(DEV), er.to(DEV), ec.to(DEV))
    sync_from_oracle(o, e, g, d, tr)
    e.eval(), g.eval()
    img, ez2, noise, _ = make_inputs(B, S, 31337 + S)
    noisy_ref, recon_ref, rl_ref, kl_ref = o.denoise(img, sigma * noise, ez2)
    out = V.denoise_eval(e, g, img.to(DEV), sigma=sigma, eps=noise.to(DEV), eps_z=ez2.to(DEV))
    a01, b01 = (recon_ref + 1) / 2, (img + 1) / 2
    err = {"recon": float((out["recon"].cpu() - recon_ref).abs().max()), "recon_loss": rel(out["recon_loss"], rl_ref),
           "kl_loss": rel(out["kl_loss"], kl_ref), "psnr": abs(out["psnr"] - R.psnr(a01, b01)),
           "ssim": abs(out["ssim"] - R.ssim(a01, b01))}
    print(f"denoise S=64 B=128 {dtype}:", {k: f"{v:.1e}" for k, v in err.items()})
    torch.testing.assert_close(out["noisy"].cpu(), noisy_ref, rtol=0, atol=1e-6)
    tol = ({"recon": 1e-3, "recon_loss": 1e-4, "kl_loss": 1e-4, "psnr": 1e-3, "ssim": 1e-4} if dtype == "fp32" else
           {"recon": 3e-2, "recon_loss": 3e-2, "kl_loss": 3e-2, "psnr": 0.2, "ssim": 2e-2})
    for k, v in err.items():
        assert v <= tol[k], f"denoise {dtype} {k}: {v:.2e} > {tol[k]:.0e}"

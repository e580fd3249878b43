"""GPU: the HIP training path against (a) golden vectors captured from the reference's own classes
(S=256, tests/golden/) and (b) the CPU oracle run live on the same seeded inputs (S=64/128).

Tolerances.  First-step losses: 1e-4 relative (BASELINE north_star).  Later steps and gradients: the
algorithm is chaotic -- the reference's OWN fp32 CPU run differs from an fp64 run of the same code
by 5e-2..3e-1 in the step-2/3 losses and by up to 1e-1 (relative to the tensor max) in some generator
gradients (sum(dz) cancellation behind BatchNorm).  Gradients are therefore judged against the fp64
oracle with a bound calibrated by the CPU-fp32 error of the same quantity; later-step losses get a
2e-2 bound against the CPU-fp32 oracle."""
import os

import numpy as np
import pytest
import torch

import vaegan_ref as R
from _inputs import make_inputs, tstats

import vaegan_amd as V

pytestmark = pytest.mark.gpu
DEV = "cuda"
torch.set_num_threads(min(16, os.cpu_count() or 1))


def build(S, dtype="fp32", lr=2e-4):
    V.configure_seed(42)
    e = V.Encoder([3, S, S], 100, dtype=dtype)
    g = V.Generator(nz=100, img_size=S, dtype=dtype)
    d = V.Discriminator(img_size=S, dtype=dtype)
    g.apply(V.weights_init)
    d.apply(V.weights_init)
    e.to(DEV), g.to(DEV), d.to(DEV)
    oE, oG, oD = (V.Adam(m.parameters(), lr=lr) for m in (e, g, d))
    tr = V.VAEGANTrainer(e, g, d, oE, oG, oD)
    tr.train()
    return e, g, d, tr


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-12)


# First-iteration bounds.  recon / kl / d_loss_1 are pure forward passes of the initial weights: 1e-4 (measured
# ~1e-6).  d_loss_2, g_loss_adv (and total) are evaluated AFTER one / two Adam(t=1) updates of the Discriminator
# inside the same iteration (vaegan_code.py:95-110): those updates are sign(g)*lr per weight, so any two fp32
# implementations differ by +-lr on noise-level gradients; at B=2, S=256 this shows as ~1.5e-4 in g_loss_adv.
FIRST_STEP_TOL = {"recon_loss": 1e-4, "kl_loss": 1e-4, "d_loss_1": 1e-4, "d_loss_2": 5e-4, "g_loss_adv": 5e-4,
                  "total": 5e-4}


def later_step_tol(ref32, ref64):
    """Free-running steps >= 2 are only a sanity check: Adam's first updates are sign(g)*lr, so every weight
    whose gradient is at rounding-noise level moves by +-lr differently in ANY two fp32 implementations, and the
    GAN dynamics amplify that (the reference's own fp32-vs-fp64 spread is 5e-2..4e-1 here).  The rigorous
    later-step checks are the teacher-forced tests below.  (Floor 1.0: at S=256, B=2 merely changing the summation
    order of the Discriminator head's dot product moved step-3 g_loss_adv from 5.6 to 8.1 against a reference 5.3.)"""
    return max(1.0, 4 * rel(ref32, ref64))


def oracle_twin_fp64(o):
    """An fp64 copy of the oracle in its CURRENT state (parameters, BatchNorm buffers, Adam moments and step counts):
    running the same step on both measures how far the reference's fp32 arithmetic itself is from exact."""
    t = R.RefVAEGAN(img_size=o.img_size, latent_dim=o.latent_dim, lr=o.opt_E.lr, seed=None).double_()
    for src, dst in ((o.E, t.E), (o.G, t.G), (o.D, t.D)):
        for k, v in src.items():
            with torch.no_grad():
                dst[k].copy_(v.detach())
    for so, do in ((o.opt_E, t.opt_E), (o.opt_G, t.opt_G), (o.opt_D, t.opt_D)):
        do.t = so.t
        for a, b in zip(do.exp_avg, so.exp_avg):
            a.copy_(b)
        for a, b in zip(do.exp_avg_sq, so.exp_avg_sq):
            a.copy_(b)
    return t


def sync_from_oracle(o, e, g, d, tr):
    """Teacher forcing: copy the oracle's parameters, BN buffers and Adam state into the HIP model."""
    for m, st in ((e, o.E), (g, o.G), (d, o.D)):
        m.load_state_dict({k: v.detach().float() for k, v in st.items()})
    for opt, ro, st in ((tr.opt_E, o.opt_E, o.E), (tr.opt_G, o.opt_G, o.G), (tr.opt_D, o.opt_D, o.D)):
        sd = {"param_groups": [dict(lr=ro.lr, betas=ro.betas, eps=ro.eps)],
              "state": {i: dict(step=torch.tensor(float(ro.t)), exp_avg=ro.exp_avg[i].float(),
                                exp_avg_sq=ro.exp_avg_sq[i].float()) for i in range(len(ro.params))}}
        opt.load_state_dict(sd)


# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("S", [64, 128, 256])
def test_encoder_forward_kat_vs_reference_golden(golden_dir, S):
    gold = np.load(os.path.join(golden_dir, "forward_kat.npz"))
    V.configure_seed(42)
    e = V.Encoder([3, S, S], 100).to(DEV)
    real = make_inputs(2, S, 1000 + S)[0].to(DEV)
    e.train()
    with torch.no_grad():
        mu, lv = e(real)
    np.testing.assert_allclose(mu.cpu().numpy(), gold[f"E{S}.train.mu"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(lv.cpu().numpy(), gold[f"E{S}.train.logvar"], rtol=1e-4, atol=2e-5)
    e.eval()
    with torch.no_grad():
        mu, lv = e(real)
    np.testing.assert_allclose(mu.cpu().numpy(), gold[f"E{S}.eval.mu"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(lv.cpu().numpy(), gold[f"E{S}.eval.logvar"], rtol=1e-4, atol=2e-5)
    assert int(e.state_dict()["cnn.0.bn.num_batches_tracked"]) == 2     # 1 (construction) + 1 train forward


def test_generator_discriminator_forward_kat_vs_reference_golden(golden_dir):
    gold = np.load(os.path.join(golden_dir, "forward_kat.npz"))
    V.configure_seed(42)
    V.Encoder([3, 256, 256], 100)                      # consume the RNG exactly as vaegan_code.py:29 does
    g, d = V.Generator(nz=100), V.Discriminator()
    g.apply(V.weights_init), d.apply(V.weights_init)
    g.to(DEV), d.to(DEV)
    z = torch.randn(2, 100, 1, 1, generator=torch.Generator().manual_seed(77)).to(DEV)
    g.train(), d.train()
    with torch.no_grad():
        img = g(z)
        p = d(img)
    s, samp = tstats(img)
    np.testing.assert_allclose(s, gold["G256.train.stats"], rtol=2e-5)
    np.testing.assert_allclose(samp, gold["G256.train.samp"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(p.cpu().numpy(), gold["D256.train.out"], rtol=1e-4, atol=1e-6)
    g.eval(), d.eval()
    with torch.no_grad():
        img2 = g(z)
        p2 = d(img)
    s, samp = tstats(img2)
    np.testing.assert_allclose(s, gold["G256.eval.stats"], rtol=1e-4)
    np.testing.assert_allclose(samp, gold["G256.eval.samp"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(p2.cpu().numpy(), gold["D256.eval.out"], rtol=1e-4, atol=1e-6)



# State after the FIRST iteration against the reference-generated checksums (steps_S256_*.npz "after1.*": sum, abs-sum,
# sum of squares of every parameter / BatchNorm buffer / Adam moment after vaegan_code.py:65-135 ran once from the
# seed-42 state).  exp_avg after one Adam step is (1-beta1)*grad, exp_avg_sq is (1-beta2)*grad^2, so the moment
# checksums pin the whole backward pass of iteration 1 (D: its two updates) against the reference.  Tolerances:
#   * Adam moments, per tensor: abs-sum and sum of squares within MOMENT_TOL relative (2.5x that for exp_avg_sq, which
#     is quadratic in the gradient); the plain sum (which cancels) within MOMENT_TOL of the abs-sum.  Measured worst
#     case: 1.6e-3 at B=4; 3.6e-2 (exp_avg_sq of the Encoder's BatchNorm gammas) at B=2, where BatchNorm backward over
#     two images is at its most ill-conditioned (test_all_parameter_gradients_vs_fp64_oracle: the reference's own fp32
#     run is that far from fp64 there).  Tensors whose reference gradient is rounding noise (a conv bias in front of
#     BatchNorm: exactly zero in exact arithmetic) are skipped -- any two fp32 implementations disagree there.
#   * BatchNorm running statistics: same form, BUFFER_TOL.
#   * parameters: one Adam(t=1) step moves every weight by lr*sign(g) (eps aside), so two implementations differ by
#     2*lr on every weight whose gradient sign differs; the checksum of the parameters must agree within
#     2*lr*n*FLIP_FRAC, i.e. at most FLIP_FRAC of a tensor's weights may have moved the other way (D takes two steps).
MOMENT_TOL, BUFFER_TOL, FLIP_FRAC = 2e-2, 2e-3, 0.05


def check_state_after_first_iteration(gold, e, g, d, tr, o64=None, lr=2e-4, tols=None):
    """tols = (moment, buffer, flip fraction) overrides the fp32 bounds above (the bf16 engine states its own)."""
    MOMENT_TOL, BUFFER_TOL, FLIP_FRAC = tols if tols is not None else (globals()["MOMENT_TOL"], globals()["BUFFER_TOL"], globals()["FLIP_FRAC"])
    worst = {"moment": 0.0, "buffer": 0.0, "param_flip": 0.0}
    ref64 = {} if o64 is None else {"E": o64.opt_E, "G": o64.opt_G, "D": o64.opt_D}
    for name, m, opt in (("E", e, tr.opt_E), ("G", g, tr.opt_G), ("D", d, tr.opt_D)):
        steps = 2 if name == "D" else 1
        sd = m.state_dict()
        for k, v in sd.items():
            ref = gold[f"after1.{name}.{k}#stats"]
            if k.endswith("num_batches_tracked"):
                assert int(v) == int(gold[f"after1.{name}.{k}#samp"][0]), k
                continue
            got, _ = tstats(v.float())
            if k.endswith("running_mean") or k.endswith("running_var"):
                err = max(abs(got[1] - ref[1]) / ref[1], abs(got[2] - ref[2]) / ref[2], abs(got[0] - ref[0]) / ref[1])
                worst["buffer"] = max(worst["buffer"], err)
                assert err <= BUFFER_TOL, f"after iteration 1: {name}.{k} checksum differs by {err:.2e}"
            elif not (name == "E" and k.endswith("conv.bias")):
                flips = abs(got[0] - ref[0]) / (2 * lr * steps * v.numel())
                worst["param_flip"] = max(worst["param_flip"], flips)
                allowed = max(FLIP_FRAC, 2.0 / v.numel())          # a 16-element BatchNorm gamma: one flipped sign is 6 %
                assert flips <= allowed, f"after iteration 1: {name}.{k} parameter checksum off by {flips:.3f} x 2*lr*n"
        hsd = opt.state_dict()["state"]
        pnames = [k for k, _ in m.named_parameters()]

        def dist(a, ref):
            return max(abs(a[1] - ref[1]) / ref[1], abs(a[2] - ref[2]) / ref[2], abs(a[0] - ref[0]) / ref[1])

        def skip(i, mom, ref, n):
            if name == "E" and pnames[i].endswith("conv.bias"):
                return True     # conv bias in front of BatchNorm: exactly-zero true gradient, rounding noise in ANY fp32 run
            return ref[1] / n < (1e-9 if mom == "exp_avg" else 1e-18)

        # calibration: how far the REFERENCE's own fp32 arithmetic (the fixture) is from an fp64 run of the same
        # iteration -- the worst tensor of this network (the error enters through the chaotic small-batch BatchNorm
        # backward chain and reaches every gradient behind it); the HIP engine may be 4x that, as for the gradients
        cal = {"exp_avg": 0.0, "exp_avg_sq": 0.0}
        if name in ref64:
            for i in range(len(hsd)):
                for mom in cal:
                    ref = gold[f"after1.adam.{name}.{i}.{mom}#stats"]
                    if not skip(i, mom, ref, hsd[i][mom].numel()):
                        r64, _ = tstats(getattr(ref64[name], mom)[i])
                        cal[mom] = max(cal[mom], dist(r64, ref))
        for i in range(len(hsd)):
            for mom in ("exp_avg", "exp_avg_sq"):
                ref = gold[f"after1.adam.{name}.{i}.{mom}#stats"]
                got, _ = tstats(hsd[i][mom])
                if skip(i, mom, ref, hsd[i][mom].numel()):
                    continue
                err = dist(got, ref)
                worst["moment"] = max(worst["moment"], err)
                worst[mom] = max(worst.get(mom, 0.0), err)
                if err == worst[mom]:
                    worst[mom + "_at"] = f"{name}.{pnames[i]}"
                tol = MOMENT_TOL if mom == "exp_avg" else 2.5 * MOMENT_TOL      # second moment: quadratic in the gradient
                tol = max(tol, 4 * cal[mom])
                assert err <= tol, (f"after iteration 1: Adam {name} param {i} ({pnames[i]}) {mom} checksum differs by "
                                    f"{err:.2e} (tol {tol:.1e}; reference fp32 vs fp64: {cal[mom]:.1e})")
    print("after-iteration-1 state vs reference checksums, worst:", {k: (v if isinstance(v, str) else f"{v:.2e}") for k, v in worst.items()})
    return worst

# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,epoch", [(2, 25), (2, 60), (4, 0)])
def test_three_training_steps_vs_reference_golden_S256(golden_dir, B, epoch):
    """vaegan_code.py:65-135 on the reference-native geometry; losses captured from the reference classes."""
    gold = np.load(os.path.join(golden_dir, f"steps_S256_B{B}_e{epoch}.npz"))
    names = [str(n) for n in gold["loss_names"]]
    e, g, d, tr = build(256)
    o64 = R.RefVAEGAN(img_size=256, seed=42).double_()      # only calibrates the later-step bound
    for step in range(3):
        real, ez, er, ec = make_inputs(B, 256, 5000 + 10 * B + step)
        r64 = o64.train_step(real, ez, er, ec, epoch)
        got = tr.loss_dict(tr.train_step(real.to(DEV), epoch, ez.to(DEV), er.to(DEV), ec.to(DEV)), epoch)
        for j, n in enumerate(names):
            ref = float(gold["losses"][step][j])
            tol = FIRST_STEP_TOL[n] if step == 0 else later_step_tol(ref, r64[n])
            assert rel(got[n], ref) <= tol, f"step {step} {n}: hip {got[n]} reference {ref} (tol {tol:.1e})"
        if step == 0:
            check_state_after_first_iteration(gold, e, g, d, tr, o64)
    assert float(tr.opt_D.state_dev[0]) == 6 and float(tr.opt_E.state_dev[0]) == 3
    for name, m in (("E", e), ("G", g), ("D", d)):
        sd = m.state_dict()
        k = [k for k in sd if k.endswith("num_batches_tracked")][0]
        assert int(sd[k]) == int(gold[f"final.{name}.{k}#samp"][0])          # E,G: +3; D: +15 (5 forwards/step)


@pytest.mark.parametrize("S,B", [(64, 4), (128, 2), (64, 16)])
def test_three_training_steps_vs_live_oracle(S, B):
    e, g, d, tr = build(S)
    o = R.RefVAEGAN(img_size=S, seed=42)
    o64 = R.RefVAEGAN(img_size=S, seed=42).double_()
    for step in range(3):
        real, ez, er, ec = make_inputs(B, S, 7000 + S + step)
        ref = o.train_step(real, ez, er, ec, 60)
        r64 = o64.train_step(real, ez, er, ec, 60)
        got = tr.loss_dict(tr.train_step(real.to(DEV), 60, ez.to(DEV), er.to(DEV), ec.to(DEV)), 60)
        for n in V.LOSS_NAMES + ("total",):
            tol = FIRST_STEP_TOL[n] if step == 0 else later_step_tol(ref[n], r64[n])
            assert rel(got[n], ref[n]) <= tol, f"S={S} step {step} {n}: hip {got[n]} oracle {ref[n]} (tol {tol:.1e})"
    for m, st in ((e, o.E), (g, o.G), (d, o.D)):
        for k, v in m.state_dict().items():
            if k.endswith("num_batches_tracked"):
                assert int(v) == int(st[k]), k


@pytest.mark.parametrize("S,B,dtype", [(64, 1, "fp32"), (64, 3, "fp32"), (64, 37, "fp32"), (128, 5, "fp32"),
                                       (64, 1, "bf16"), (64, 37, "bf16")])
def test_ragged_batch_sizes_first_iteration(S, B, dtype):
    """The reference's loader keeps the ragged last batch (drop_last=False, vaegan_code.py:67): any batch size must
    work, including 1 (BatchNorm over the spatial positions only) and sizes that fill no tile."""
    e, g, d, tr = build(S, dtype=dtype)
    o = R.RefVAEGAN(img_size=S, seed=42)
    real, ez, er, ec = make_inputs(B, S, 7300 + B)
    ref = o.train_step(real, ez, er, ec, 60)
    got = tr.loss_dict(tr.train_step(real.to(DEV), 60, ez.to(DEV), er.to(DEV), ec.to(DEV)), 60)
    for n in V.LOSS_NAMES + ("total",):
        # B = 1: every BatchNorm normalises over 4..1024 spatial values of ONE image; d_loss_2 / g_loss_adv sit behind
        # the Adam(t=1) sign-updates of D computed from that single sample -> 2e-3 there
        post = n in ("d_loss_2", "g_loss_adv", "total")            # evaluated after in-iteration Adam(t=1) updates of D
        if dtype == "bf16":
            tol = 6e-2 if (B == 1 and post) else 3e-2               # single-sample BatchNorm + bf16 rounding: measured 4e-2
        else:
            tol = 2e-3 if (B == 1 and post) else FIRST_STEP_TOL[n]
        assert rel(got[n], ref[n]) <= tol, f"S={S} B={B} {dtype} {n}: hip {got[n]} oracle {ref[n]}"


@pytest.mark.parametrize("S,B", [(64, 8), (256, 2)])
def test_teacher_forced_steps_adam_and_batchnorm_state(S, B):
    """Steps 1..3 each started from the ORACLE's state (parameters, BN buffers, Adam moments/step), so Adam with
    t > 1, non-trivial running statistics and momentum history are checked without chaotic compounding."""
    e, g, d, tr = build(S)
    o = R.RefVAEGAN(img_size=S, seed=42)
    worst_moment = 0.0
    for step in range(3):
        sync_from_oracle(o, e, g, d, tr)
        before_all = {id(st): {k: v.detach().clone() for k, v in st.items()} for st in (o.E, o.G, o.D)}
        real, ez, er, ec = make_inputs(B, S, 9000 + S + step)
        o64 = oracle_twin_fp64(o)
        o64.train_step(real, ez, er, ec, 25)
        ref = o.train_step(real, ez, er, ec, 25)
        got = tr.loss_dict(tr.train_step(real.to(DEV), 25, ez.to(DEV), er.to(DEV), ec.to(DEV)), 25)
        for n in V.LOSS_NAMES:
            assert rel(got[n], ref[n]) <= 2 * FIRST_STEP_TOL[n], f"S={S} forced step {step} {n}: hip {got[n]} oracle {ref[n]}"
        for m, st, opt, ro in ((e, o.E, tr.opt_E, o.opt_E), (g, o.G, tr.opt_G, o.opt_G), (d, o.D, tr.opt_D, o.opt_D)):
            assert float(opt.state_dev[0]) == ro.t
            before = before_all[id(st)]
            # Adam moments, PER TENSOR, in the calibration of check_state_after_first_iteration: relative Frobenius
            # distance from the CPU-fp32 oracle <= max(MOMENT_TOL (2.5x for the quadratic second moment), 4 x the
            # distance of that oracle from an fp64 run of the SAME step from the SAME state) -- the fp64 twin is a copy
            # of the oracle taken before the step (the worst generator gradients carry up to 1e-1 of their size as fp32
            # error in the reference too, test_all_parameter_gradients_vs_fp64_oracle).  Round 2 held the concatenation
            # of all tensors to a flat 15 % of its max.
            ro64 = {id(o.opt_E): o64.opt_E, id(o.opt_G): o64.opt_G, id(o.opt_D): o64.opt_D}[id(ro)]
            hsd = opt.state_dict()["state"]
            pk = R.trainable_keys(st)
            for i in range(len(ro.params)):
                if pk[i].endswith("conv.bias") and m is e:
                    continue                    # exactly-zero true gradient: rounding noise in every implementation
                for mom, base in (("exp_avg", MOMENT_TOL), ("exp_avg_sq", 2.5 * MOMENT_TOL)):
                    r32, r64 = getattr(ro, mom)[i].double(), getattr(ro64, mom)[i].double()
                    if float(r32.norm()) == 0.0:
                        continue
                    cal = float((r32 - r64).norm() / r32.norm())
                    err = float((hsd[i][mom].double().cpu() - r32).norm() / r32.norm())
                    worst_moment = max(worst_moment, err)
                    assert err <= max(base, 4 * cal), (f"forced step {step} {pk[i]} {mom}: {err:.2e} from the fp32 oracle "
                                                       f"(its own distance from fp64: {cal:.1e})")
            for k, v in m.state_dict().items():
                r = st[k].detach()
                if k.endswith("num_batches_tracked"):
                    assert int(v) == int(r), k
                elif k.endswith("running_mean") or k.endswith("running_var"):
                    err = float((v.cpu() - r).abs().max() / r.abs().max())
                    assert err <= 5e-3, f"step {step} {k}: running stat differs by {err:.2e} of its max"   # D: 3 of its 5 forwards follow Adam(t=1) updates
                elif k.endswith("conv.bias"):
                    # A conv bias in front of BatchNorm has an exactly-zero true gradient: what Adam normalises
                    # there is pure rounding noise (|g| ~ 1e-8..1e-7 in the reference too) -> +-lr steps of random
                    # sign in every implementation; only boundedness can be checked.
                    assert float((v.cpu() - before[k]).abs().max()) <= 1.05 * 2e-4, k
                else:
                    # One Adam step: |delta| <= ~lr.  Isolated elements whose gradient is rounding noise flip sign
                    # (error up to 2*lr in any fp32 implementation); the bulk must agree.
                    err = ((v.cpu() - before[k]).double() - (r - before[k]).double()).abs() / 2e-4
                    assert float(err.median()) <= 0.02 and float(err.mean()) <= 0.25, \
                        f"step {step} {k}: median/mean update error {float(err.median()):.3f}/{float(err.mean()):.3f} lr"
    print(f"teacher-forced S={S} B={B}: worst per-tensor Adam-moment distance from the fp32 oracle {worst_moment:.2e}")


def _grads_step(S, B, dtype="fp32"):
    e, g, d, tr = build(S, dtype=dtype, lr=0.0)
    real, ez, er, ec = make_inputs(B, S, 7000 + S)
    tr.train_step(real.to(DEV), 60, ez.to(DEV), er.to(DEV), ec.to(DEV))
    return {f"{n}.{k}": p.grad.double().cpu() for n, m in (("E", e), ("G", g), ("D", d)) for k, p in m.named_parameters()}


def _oracle_grads(S, B, double):
    o = R.RefVAEGAN(img_size=S, seed=42, lr=0.0)
    if double:
        o.double_()
    real, ez, er, ec = make_inputs(B, S, 7000 + S)
    o.train_step(real, ez, er, ec, 60)
    return {f"{n}.{k}": st[k].grad.double() for n, st in (("E", o.E), ("G", o.G), ("D", o.D)) for k in R.trainable_keys(st)}


def test_all_parameter_gradients_vs_fp64_oracle():
    """Every gradient of one full iteration (E, G, D incl. the accumulated generator-loss pass), lr=0.
    Bound per tensor: max-error relative to the tensor's max <= max(1e-5, 4 x the CPU-fp32 oracle's own error)."""
    S, B = 64, 4
    hip = _grads_step(S, B)
    g32, g64 = _oracle_grads(S, B, False), _oracle_grads(S, B, True)
    assert sorted(hip) == sorted(g64)
    for k, r in g64.items():
        scale = float(r.abs().max())
        if scale < 1e-6:
            # conv bias in front of BatchNorm: exactly zero gradient in exact arithmetic, rounding noise otherwise
            assert float(hip[k].abs().max()) < 1e-5, k
            continue
        err_hip = float((hip[k] - r).abs().max()) / scale
        err_cpu = float((g32[k] - r).abs().max()) / scale
        assert err_hip <= max(1e-5, 4 * err_cpu), f"{k}: hip err {err_hip:.2e}, cpu-fp32 err {err_cpu:.2e}"


def test_step_is_bitwise_deterministic():
    outs = []
    for _ in range(2):
        e, g, d, tr = build(64)
        for step in range(2):
            real, ez, er, ec = (t.to(DEV) for t in make_inputs(4, 64, 7064 + step))
            l = tr.train_step(real, 60, ez, er, ec)
        outs.append((l[:5].cpu().clone(), tr.opt_G.flat_p.cpu().clone(), tr.opt_D.flat_p.cpu().clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


def test_bf16_engine_tracks_fp32_oracle():
    """bf16 storage / f32 accumulate / fp32 master weights (the throughput path, BASELINE configs[1]) against the
    fp32 CPU oracle: first-iteration losses within 3e-2 (bf16 has 8 significant bits; stated bf16 tolerance)."""
    S, B = 64, 16
    e, g, d, tr = build(S, dtype="bf16")
    o = R.RefVAEGAN(img_size=S, seed=42)
    real, ez, er, ec = make_inputs(B, S, 7000 + S)
    ref = o.train_step(real, ez, er, ec, 60)
    got = tr.loss_dict(tr.train_step(real.to(DEV), 60, ez.to(DEV), er.to(DEV), ec.to(DEV)), 60)
    for n in V.LOSS_NAMES:
        assert rel(got[n], ref[n]) <= 3e-2, f"bf16 {n}: hip {got[n]} oracle {ref[n]}"
    for p in list(e.parameters()) + list(g.parameters()) + list(d.parameters()):
        assert p.dtype == torch.float32 and bool(torch.isfinite(p).all())


@pytest.mark.parametrize("S,B,dtype", [(64, 128, "fp32"), (64, 128, "bf16"), (128, 64, "bf16")])
def test_benchmarked_configurations_replayed_graph_vs_oracle(S, B, dtype):
    """The configurations bench.py / BASELINE.json time -- C2 (S=64, B=128, bf16, hipGraph replay) with its fp32
    parity twin, and the per-GPU shard of C3 (S=128, B=64, bf16) -- end to end against the live CPU oracle: the first
    iteration from the seed-42 state, executed by the REPLAYED graph (the tile picker chooses other kernels at these
    batch sizes than at the small parity batches: 256x128 / 128x128 patch tiles, grouped 2B-row D passes)."""
    e, g, d, tr = build(S, dtype=dtype)
    o = R.RefVAEGAN(img_size=S, seed=42)
    real, ez, er, ec = make_inputs(B, S, 1234)
    dev_in = [t.to(DEV) for t in (real, ez, er, ec)]
    start = tr.state_dict()
    start = {k: (v if not isinstance(v, dict) else _deep_clone(v)) for k, v in start.items()}
    eager = tr.loss_dict(tr.train_step_graphed(dev_in[0], 60, *dev_in[1:]).clone(), 60)     # eager warm-up call
    tr.train_step_graphed(dev_in[0], 60, *dev_in[1:])                                        # capture + first replay
    assert tr._graph is not None and len(tr._graph[1]) == 1
    tr.load_state_dict(start)                                                                # back to the seed-42 state
    graph = tr._graph
    got = tr.loss_dict(tr.train_step_graphed(dev_in[0], 60, *dev_in[1:]).clone(), 60)        # pure replay
    assert tr._graph is graph, "the restored state must replay the captured graph, not re-capture"
    ref = o.train_step(real, ez, er, ec, 60)
    for n in V.LOSS_NAMES + ("total",):
        # fp32 at the benchmarked batch: EVERY loss inside north_star's 1e-4, also the two evaluated after Adam(t=1)
        # updates of D (measured <= 2.1e-5; the looser FIRST_STEP_TOL entries are for the B = 2..4 fixtures)
        tol = 1e-4 if dtype == "fp32" else 3e-2                       # bf16: stated tolerance of the bf16 path
        assert rel(got[n], ref[n]) <= tol, f"S={S} B={B} {dtype} replay {n}: hip {got[n]} oracle {ref[n]} (tol {tol:.0e})"
        assert got[n] == eager[n], f"replayed graph differs from the eager iteration in {n}: {got[n]} vs {eager[n]}"
    print(f"S={S} B={B} {dtype}:", {n: f"{rel(got[n], ref[n]):.1e}" for n in V.LOSS_NAMES})


def _deep_clone(x):
    if isinstance(x, torch.Tensor):
        return x.detach().clone()
    if isinstance(x, dict):
        return {k: _deep_clone(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_deep_clone(v) for v in x)
    return x


def test_fp8_forward_engine_tracks_the_bf16_engine():
    """dtype='fp8' (BASELINE configs[4]: e4m3 operands for the forward GEMMs of the wide conv layers, f32 accumulate;
    backward, storage and master weights as the bf16 engine).  The reference has NO fp8 semantics -- this is a roofline
    configuration without a parity claim -- so the check is statistical: first-iteration losses within 10 % of the
    bf16 engine's on the same inputs (e4m3 carries 3 mantissa bits), everything finite, and the forward GEMMs of the
    eligible layers did run on the fp8 kernel."""
    S, B = 64, 16
    real, ez, er, ec = (t.to(DEV) for t in make_inputs(B, S, 7000 + S))
    out = {}
    for dtype in ("bf16", "fp8"):
        e, g, d, tr = build(S, dtype=dtype)
        out[dtype] = tr.loss_dict(tr.train_step(real, 60, ez, er, ec), 60)
        if dtype == "fp8":
            assert [g._engine.fp8_ok(i) for i in range(6)] == [False, True, True, True, True, False]   # G0 is a plain GEMM, G5 an edge layer
            assert [d._engine.fp8_ok(i) for i in range(4)] == [False, True, True, True]                  # D0 reads the 3-channel image
            assert e._engine.fp8_ok(0) is False and e._engine.fp8_ok(1) is True
            for p in list(e.parameters()) + list(g.parameters()) + list(d.parameters()):
                assert p.dtype == torch.float32 and bool(torch.isfinite(p).all())
    for n in V.LOSS_NAMES:
        assert rel(out["fp8"][n], out["bf16"][n]) <= 0.10, f"fp8 {n}: {out['fp8'][n]} vs bf16 {out['bf16'][n]}"
    print("fp8 vs bf16 first-iteration losses:", {n: f"{rel(out['fp8'][n], out['bf16'][n]):.1e}" for n in V.LOSS_NAMES})


def test_grouped_discriminator_pass_equals_separate_passes():
    """One grouped 2B-row pass per D iteration (per-group BatchNorm statistics) vs the reference's two calls."""
    outs = []
    for grouped in (True, False):
        e, g, d, tr = build(64)
        tr.group_d_passes = grouped
        real, ez, er, ec = (t.to(DEV) for t in make_inputs(8, 64, 4321))
        l = tr.train_step(real, 60, ez, er, ec)
        assert d._engine.can_group(8, 2, real) is True
        outs.append((l[:5].cpu().clone(), {k: v.cpu().clone() for k, v in d.state_dict().items()},
                     tr.opt_D.exp_avg.cpu().clone(), tr.opt_G.exp_avg.cpu().clone()))
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=2e-4, atol=1e-6)   # adv loss sits behind 2 Adam steps of D
    for k, v in outs[0][1].items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(outs[1][1][k]) == 5
        elif "running" in k:
            torch.testing.assert_close(v, outs[1][1][k], rtol=1e-4, atol=1e-6, msg=k)
    for a, b in ((outs[0][2], outs[1][2]), (outs[0][3], outs[1][3])):          # Adam first moments = 0.1 * grads
        assert float((a - b).abs().max() / b.abs().max()) < 2e-3


@pytest.mark.parametrize("dtype,B", [("fp32", 8), ("bf16", 128)])
def test_fused_head_backward_is_bit_identical_to_the_three_launches(dtype, B):
    """vg_head_backward (BCE of the Discriminator's output + its gradient + sigmoid backward + the head's data and weight
    gradients in ONE launch, vaegan_code.py:99-104 / :115,133) against the path it replaces (vg_bce[_pair]_forward_backward,
    vg_dot_sigmoid_backward, vg_dot_wgrad): two whole iterations, every loss, parameter and Adam moment bit for bit."""
    outs = []
    for fused in (True, False):
        e, g, d, tr = build(64, dtype=dtype)
        tr.fuse_head_backward = fused
        ls = []
        for step in range(2):
            real, ez, er, ec = (t.to(DEV) for t in make_inputs(B, 64, 4400 + step))
            ls.append(tr.train_step(real, 60, ez, er, ec)[:5].clone())
        torch.cuda.synchronize()
        outs.append((torch.stack(ls).cpu(), [o.flat_p.cpu().clone() for o in (tr.opt_E, tr.opt_G, tr.opt_D)],
                     [o.exp_avg.cpu().clone() for o in (tr.opt_E, tr.opt_G, tr.opt_D)], tr.opt_D.flat_g.cpu().clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1] + outs[0][2] + [outs[0][3]], outs[1][1] + outs[1][2] + [outs[1][3]]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype,B,graphed", [("fp32", 8, False), ("bf16", 128, False), ("bf16", 128, True)])
def test_step_prologue_is_bit_identical_to_per_optimizer_prepares(dtype, B, graphed):
    """The noise counter and the three optimizers' step counters / bias corrections prepared by ONE launch at the top of
    the iteration (ops.step_prologue) against one prepare launch per optimizer.step() (vaegan_code.py:105, :134-135): three
    (four) iterations, eager and replayed -- every loss, parameter, Adam moment and device-side counter bit for bit."""
    outs = []
    for merged in (True, False):
        e, g, d, tr = build(64, dtype=dtype)
        tr.fuse_step_prologue = merged
        fn = tr.train_step_graphed if graphed else tr.train_step
        ls = []
        for step in range(4 if graphed else 3):
            real, ez, er, ec = (t.to(DEV) for t in make_inputs(B, 64, 4500 + step))
            ls.append(fn(real, 60, ez, er, ec)[:5].clone())
        torch.cuda.synchronize()
        outs.append((torch.stack(ls).cpu(), [o.flat_p.cpu().clone() for o in (tr.opt_E, tr.opt_G, tr.opt_D)],
                     [o.exp_avg.cpu().clone() for o in (tr.opt_E, tr.opt_G, tr.opt_D)],
                     [o.exp_avg_sq.cpu().clone() for o in (tr.opt_E, tr.opt_G, tr.opt_D)],
                     [(o.steps, float(o.state_dev[0]), float(o.state_dev[1]), float(o.state_dev[2]))
                      for o in (tr.opt_E, tr.opt_G, tr.opt_D)]))
    for it in range(outs[0][0].shape[0]):
        assert torch.equal(outs[0][0][it], outs[1][0][it]), f"losses of iteration {it + 1} differ: {outs[0][0][it] - outs[1][0][it]}"
    for k in (1, 2, 3):
        for a, b in zip(outs[0][k], outs[1][k]):
            assert torch.equal(a, b)
    assert outs[0][4] == outs[1][4]


@pytest.mark.parametrize("dtype,B,graphed,inject", [("fp32", 8, False, True), ("bf16", 128, False, True), ("bf16", 128, True, True),
                                                    ("bf16", 128, True, False)])
def test_merged_small_launches_are_bit_identical_to_the_separate_ones(dtype, B, graphed, inject):
    """Round 4: four pairs of small launches per iteration merged -- Encoder-input conversion + noisy real batch (one pass
    over the images), the MSE's final sum inside the KL launch, the Encoder's + the Generator's Adam step, the loss slots'
    memset inside the step prologue (vaegan_code.py:74/:91, :113-114, :134-135) -- against VAEGANTrainer with
    merge_small_launches = False: losses, parameters, Adam moments and counters of three (four) iterations bit for bit,
    eager and replayed, with injected and with in-kernel noise; and the launch count drops by four (bf16)."""
    import importlib
    ops = importlib.import_module(V.Encoder.__module__.rsplit(".", 1)[0] + ".ops")
    outs, counts = [], []
    for merged in (True, False):
        e, g, d, tr = build(64, dtype=dtype)
        torch.cuda.manual_seed(77)
        tr.merge_small_launches = merged
        fn = tr.train_step_graphed if graphed else tr.train_step
        ls = []
        for step in range(4 if graphed else 3):
            real, ez, er, ec = (t.to(DEV) for t in make_inputs(B, 64, 4600 + step))
            n0 = ops.launch_count()
            out = fn(real, 60, ez, er, ec) if inject else fn(real, 60)
            ls.append(out[:5].clone())
            if step == 0:
                counts.append(ops.launch_count() - n0)       # (the first call is eager in both modes)
        torch.cuda.synchronize()
        outs.append((torch.stack(ls).cpu(), [o.flat_p.cpu().clone() for o in (tr.opt_E, tr.opt_G, tr.opt_D)],
                     [o.exp_avg.cpu().clone() for o in (tr.opt_E, tr.opt_G, tr.opt_D)],
                     [o.exp_avg_sq.cpu().clone() for o in (tr.opt_E, tr.opt_G, tr.opt_D)],
                     [(o.steps, float(o.state_dev[0]), float(o.state_dev[1]), float(o.state_dev[2]))
                      for o in (tr.opt_E, tr.opt_G, tr.opt_D)]))
    for it in range(outs[0][0].shape[0]):
        assert torch.equal(outs[0][0][it], outs[1][0][it]), f"losses of iteration {it + 1} differ: {outs[0][0][it] - outs[1][0][it]}"
    for k in (1, 2, 3):
        for a, b in zip(outs[0][k], outs[1][k]):
            assert torch.equal(a, b)
    assert outs[0][4] == outs[1][4]
    assert counts[1] - counts[0] == (3 if dtype == "bf16" else 2), counts     # (+ the memset, which is not a library launch)


class _LocalReducer:
    """world_size-1 stand-in with the whole-buffer GradReducer surface: lets the single GPU exercise the SEGMENTED
    graph path (collectives between hipGraph segments) and records how the trainer drives it."""

    def __init__(self):
        self.calls = []

    def reduce(self, opt):
        self.calls.append(("reduce", id(opt)))
        opt.flat_g.mul_(1.0)                       # an eager op on the gradient buffer between two segments

    def reduce_async(self, opt):
        self.calls.append(("async", id(opt)))

    def wait(self, opt):
        self.calls.append(("wait", id(opt)))


class _LocalBucketReducer(_LocalReducer):
    """Same, with the BUCKETED surface (plan / launch_ready / drain / outstanding) of ddp.GradReducer: buckets are
    planned by the real ddp.plan_buckets, 'launching' one is an eager op on its slice of the gradient buffer."""

    def __init__(self, bucket_bytes):
        super().__init__()
        self.bucket_bytes, self.plans, self.pending = bucket_bytes, {}, 0

    def plan(self, opt, ready):
        import importlib
        ddp = importlib.import_module("vae-gan-based-model-for-image-generation-and-denoising_amd.ddp")
        self.plans[id(opt)] = ddp.plan_buckets(opt.offsets, [p.numel() for p in opt.params], ready,
                                               opt.flat_g.numel(), self.bucket_bytes)
        return self.plans[id(opt)]

    def launch_ready(self, opt, event):
        n = 0
        for lo, hi, ev in self.plans[id(opt)]:
            if ev == event:
                opt.flat_g[lo:hi].mul_(1.0)
                self.calls.append(("bucket", id(opt), lo, hi))
                self.pending += 1
                n += 1
        return n

    def wait(self, opt):
        super().wait(opt)
        self.pending = 0

    def drain(self):
        self.pending = 0

    def outstanding(self):
        return self.pending


def test_segmented_graph_replay_with_reducer_is_bitwise_identical_to_eager():
    res, calls = [], []
    for graphed in (False, True):
        e, g, d, tr = build(64)
        tr.reducer = _LocalReducer()
        fn = tr.train_step_graphed if graphed else tr.train_step
        for step in range(4):
            real, ez, er, ec = (t.to(DEV) for t in make_inputs(4, 64, 7064 + step))
            l = fn(real, 60, ez, er, ec)
        res.append((l[:5].cpu().clone(), tr.opt_E.flat_p.cpu().clone(), tr.opt_G.flat_p.cpu().clone(),
                    tr.opt_D.flat_p.cpu().clone(), float(tr.opt_D.state_dev[0])))
        calls.append([c[0] for c in tr.reducer.calls])
        if graphed:
            assert len(tr._graph[1]) == 5 and len(tr._graph[2]) == 4        # 5 segments, 4 hand-offs
    for a, b in zip(res[0][:4], res[1][:4]):
        assert torch.equal(a, b)
    assert res[0][4] == res[1][4] == 8.0
    per_step = ["reduce", "wait", "reduce", "wait", "async", "reduce", "wait", "wait"]   # D, D, G (async), E + wait(E), wait(G)
    assert calls[0] == per_step * 4
    assert calls[1] == per_step * 4                  # eager warm-up + 3 replays (capture itself runs no collective)


def test_bucketed_reduction_cuts_the_graph_at_every_bucket_and_equals_eager():
    """ddp buckets (reverse layer order) under graph replay: one hipGraph segment boundary per bucket launch, every
    bucket of every optimizer launched exactly once per use and before that optimizer's step, results bit-identical
    to the eager iteration.  Small bucket size so that each network has several buckets."""
    res, calls, plans = [], [], []
    for graphed in (False, True):
        e, g, d, tr = build(64)
        tr.reducer = _LocalBucketReducer(bucket_bytes=1 << 20)
        fn = tr.train_step_graphed if graphed else tr.train_step
        for step in range(4):
            real, ez, er, ec = (t.to(DEV) for t in make_inputs(4, 64, 7064 + step))
            l = fn(real, 60, ez, er, ec)
        res.append((l[:5].cpu().clone(), tr.opt_E.flat_p.cpu().clone(), tr.opt_G.flat_p.cpu().clone(),
                    tr.opt_D.flat_p.cpu().clone()))
        calls.append(list(tr.reducer.calls))
        plans.append({k: list(v) for k, v in tr.reducer.plans.items()})
        if graphed:
            nb = {o: len(plans[-1][id(o)]) for o in (tr.opt_E, tr.opt_G, tr.opt_D)}
            cuts_in_backward = lambda o: len({ev for _, _, ev in plans[-1][id(o)] if ev > 0})
            expect_cuts = 2 * (cuts_in_backward(tr.opt_D) + 1) + (cuts_in_backward(tr.opt_G) + 1) + (cuts_in_backward(tr.opt_E) + 1)
            assert len(tr._graph[2]) == expect_cuts and len(tr._graph[1]) == expect_cuts + 1
            assert nb[tr.opt_G] >= 3 and nb[tr.opt_D] >= 2
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    assert [c[0] for c in calls[0]] == [c[0] for c in calls[1]]              # same hand-off sequence eager / replayed
    # per iteration: every bucket of D twice (two D updates), of G and E once
    per_iter = len(calls[1]) // 4
    one = [c for c in calls[1][:per_iter] if c[0] == "bucket"]
    for o, uses in ((tr.opt_D, 2), (tr.opt_G, 1), (tr.opt_E, 1)):
        got = sorted((c[2], c[3]) for c in one if c[1] == id(o))
        want = sorted([(lo, hi) for lo, hi, _ in plans[1][id(o)]] * uses)
        assert got == want


def test_graph_capture_is_refused_while_collectives_are_outstanding():
    """Structural guard for the abort recorded in round 1 (c10d's watchdog polling an unfinished collective with
    hipEventQuery next to a capturing stream): before capture_begin the trainer drains the reducer and refuses to
    capture if it still reports outstanding work -- instead of relying on it having happened to be idle."""
    e, g, d, tr = build(64)

    class _Stuck(_LocalBucketReducer):
        def drain(self):                            # a reducer whose work cannot be retired
            pass

        def wait(self, opt):
            _LocalReducer.wait(self, opt)

    tr.reducer = _Stuck(bucket_bytes=1 << 20)
    real, ez, er, ec = (t.to(DEV) for t in make_inputs(4, 64, 7064))
    tr.train_step_graphed(real, 60, ez, er, ec)                  # eager warm-up: leaves `pending` > 0 (never retired)
    assert tr.reducer.outstanding() > 0
    before = tr.opt_G.flat_p.clone()
    with pytest.raises(RuntimeError, match="capture refused"):
        tr.train_step_graphed(real, 60, ez, er, ec)
    assert torch.equal(before, tr.opt_G.flat_p) and tr._graph is None       # nothing ran, nothing half-captured
    tr.reducer.pending = 0                                        # once idle, the same call captures and replays
    tr.train_step_graphed(real, 60, ez, er, ec)
    assert tr._graph is not None


def test_replayed_graph_survives_workspace_growth_by_later_eager_work():
    """A captured graph holds raw pointers into the shared scratch buffers (BatchNorm slabs, split-K / wgrad slabs).
    A later, LARGER eager call grows those buffers; the superseded ones must stay alive (ops._Workspace retires them)
    or the graph would replay into freed memory.  Capture at B=8, run another trainer eagerly at B=32 (and let the
    allocator reuse whatever was freed), replay: must equal an undisturbed run bit for bit."""
    import gc
    ins = [tuple(t.to(DEV) for t in make_inputs(8, 64, 7064 + i)) for i in range(4)]

    import importlib
    ops_mod = importlib.import_module("vae-gan-based-model-for-image-generation-and-denoising_amd.ops")

    def run(disturb):
        ops_mod.WS.bufs.clear()                     # scratch sized by THIS run's B=8 capture, whatever ran before
        e, g, d, tr = build(64)
        out = []
        for i in range(4):
            if disturb and i == 2:
                # a later, larger request for EVERY scratch buffer the captured graph points into (what a second
                # trainer at a larger size, or a bigger validation batch, does through the same ops.WS) ...
                before = {k: v.data_ptr() for k, v in ops_mod.WS.bufs.items()}
                assert before
                for (tag, dev), buf in list(ops_mod.WS.bufs.items()):
                    ops_mod.WS.get(tag, buf.numel() * 4 * 2, buf.device)
                assert all(ops_mod.WS.bufs[k].data_ptr() != p for k, p in before.items()), "the workspaces did not grow"
                # ... followed by real work of another trainer and allocations that would recycle anything freed
                e2, g2, d2, tr2 = build(64)
                big = tuple(t.to(DEV) for t in make_inputs(16, 64, 99))
                tr2.train_step(big[0], 60, *big[1:])
                del e2, g2, d2, tr2, big
                gc.collect()
                junk = [torch.full((1 << 20,), float("nan"), device=DEV) for _ in range(64)]   # poison recycled blocks
                del junk
            out.append(tr.train_step_graphed(ins[i][0], 60, *ins[i][1:])[:5].clone())
        return torch.stack(out).cpu(), tr.opt_G.flat_p.cpu().clone(), tr.opt_D.flat_p.cpu().clone()

    ref, dis = run(False), run(True)
    for a, b in zip(ref, dis):
        assert torch.equal(a, b)


def test_graph_replay_is_bitwise_identical_to_eager():
    """train_step_graphed (eager warm-up, capture, replays) == train_step, bit for bit, incl. BN counters."""
    res = []
    for graphed in (False, True):
        e, g, d, tr = build(64)
        fn = tr.train_step_graphed if graphed else tr.train_step
        for step in range(4):
            real, ez, er, ec = (t.to(DEV) for t in make_inputs(4, 64, 7064 + step))
            l = fn(real, 60, ez, er, ec)
        res.append((l[:5].cpu().clone(), tr.opt_E.flat_p.cpu().clone(), tr.opt_G.flat_p.cpu().clone(),
                    tr.opt_D.flat_p.cpu().clone(), {k: v.cpu().clone() for k, v in d.state_dict().items()},
                    (tr.opt_D.steps, float(tr.opt_D.state_dev[0]))))
    for a, b in zip(res[0][:4], res[1][:4]):
        assert torch.equal(a, b)
    for k in res[0][4]:
        assert torch.equal(res[0][4][k], res[1][4][k]), k
    assert res[0][5] == res[1][5] == (8, 8.0)
    assert len(tr._graph[1]) == 1                    # no reducer: the whole iteration is ONE graph


# --------------------------------------------------------------------------------------------------
def test_dropin_autograd_path_matches_direct_trainer_and_oracle():
    """The reference trainer's own code shape (vaegan_code.py:65-135: module calls, torch ops, .backward(),
    optimizer.step()) running on the engine's nn.Modules + Adam -- the drop-in path of INTEGRATION.md."""
    S, B, epoch = 64, 4, 60
    V.configure_seed(42)
    encoder = V.Encoder([3, S, S], 100)
    decoder = V.Generator(nz=100, img_size=S)
    discriminator = V.Discriminator(img_size=S)
    decoder.apply(V.weights_init)          # on the host generator, like the CPU oracle (the reference applies it
    discriminator.apply(V.weights_init)    # after .to(device), i.e. with the device generator when on a GPU)
    encoder.to(DEV), decoder.to(DEV), discriminator.to(DEV)
    opt_E = V.Adam(encoder.parameters(), lr=2e-4)
    opt_Dec = V.Adam(decoder.parameters(), lr=2e-4)
    opt_Dis = V.Adam(discriminator.parameters(), lr=2e-4)
    bce, mse = torch.nn.BCELoss(), torch.nn.MSELoss(reduction='mean')
    encoder.train(), decoder.train(), discriminator.train()
    o = R.RefVAEGAN(img_size=S, seed=42)
    for step in range(2):
        real, ez, er, ec = make_inputs(B, S, 7000 + S + step)
        ref = o.train_step(real, ez, er, ec, epoch)
        real_images, ez, er, ec = (t.to(DEV) for t in (real, ez, er, ec))
        mu, logvar = encoder(real_images)
        logvar = torch.clamp(logvar, min=-10, max=10)
        std = torch.exp(0.5 * logvar)
        z = (mu + std * ez).unsqueeze(-1).unsqueeze(-1)
        recon_images = decoder(z)
        real_labels = torch.full((B,), 0.9, device=DEV)
        fake_labels = torch.full((B,), 0.1, device=DEV)
        real_images_noisy = real_images + 0.05 * er
        recon_images_noisy = recon_images + 0.05 * ec
        dl = []
        for _ in range(2):
            real_output = discriminator(real_images_noisy)
            fake_output = discriminator(recon_images_noisy.detach())
            d_loss = bce(real_output, real_labels) + bce(fake_output, fake_labels)
            opt_Dis.zero_grad()
            d_loss.backward()
            opt_Dis.step()
            dl.append(d_loss.item())
        fake_output = discriminator(recon_images_noisy)
        recon_loss = mse(recon_images, real_images)
        kl_loss = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp()) / B
        g_loss_adv = bce(fake_output, real_labels)
        total = recon_loss + 0.1 * min(1.0, epoch / 50) * kl_loss + 0.1 * g_loss_adv
        opt_E.zero_grad()
        opt_Dec.zero_grad()
        total.backward()
        opt_E.step()
        opt_Dec.step()
        got = dict(recon_loss=recon_loss.item(), kl_loss=kl_loss.item(), g_loss_adv=g_loss_adv.item(),
                   d_loss_1=dl[0], d_loss_2=dl[1], total=total.item())
        for n, v in got.items():
            tol = FIRST_STEP_TOL[n] if step == 0 else 0.5
            assert rel(v, ref[n]) <= tol, f"step {step} {n}: drop-in {v} oracle {ref[n]}"


def test_autograd_returns_real_gradients_where_adam_has_not_homed_them():
    """writing parameter gradients in place and handing autograd None is only legal for parameters whose
    .grad is vaegan_amd.Adam's slot.  (a) modules WITHOUT the engine's Adam: loss.backward() populates .grad through
    AccumulateGrad (hooks fire) and torch.autograd.grad returns tensors; (b) with Adam homed and DIRECT_PARAM_GRADS=False
    the returned gradients equal what the in-place path writes, bit for bit; (c) a frozen parameter gets no .grad."""
    import importlib
    nets_mod = importlib.import_module(V.Discriminator.__module__)
    S, B = 64, 4
    real = make_inputs(B, S, 8800)[0].to(DEV)

    def build():
        V.configure_seed(42)
        d = V.Discriminator(img_size=S)
        d.apply(V.weights_init)
        return d.to(DEV).train()

    # (a) no vaegan_amd.Adam at all: gradients must come back through autograd
    d = build()
    fired = []
    w0 = next(d.parameters())
    w0.register_hook(lambda g: fired.append(g.shape))
    frozen = list(d.parameters())[-1]
    frozen.requires_grad_(False)
    d(real).sum().backward()
    assert fired and frozen.grad is None
    ga = {n: p.grad.clone() for n, p in d.named_parameters() if p.requires_grad}
    assert all(g is not None and torch.isfinite(g).all() for g in ga.values())
    params = [p for p in d.parameters() if p.requires_grad]
    gl = torch.autograd.grad(d(real).sum(), params)
    assert all(g is not None for g in gl)
    # (b) homed by Adam: in place (default) == returned (DIRECT_PARAM_GRADS = False)
    res = []
    for direct in (True, False):
        d = build()
        opt = V.Adam(d.parameters(), lr=2e-4)
        nets_mod.DIRECT_PARAM_GRADS = direct
        try:
            opt.zero_grad()
            d(real).sum().backward()
        finally:
            nets_mod.DIRECT_PARAM_GRADS = True
        res.append(opt.flat_g.clone())
    assert torch.equal(res[0], res[1])
    live = [n for n in ga if not n.endswith("conv.bias")]
    for n, p in build().named_parameters():
        if n in live:
            o = dict(zip([k for k, _ in d.named_parameters()], opt.offsets))[n]
            torch.testing.assert_close(res[0][o:o + p.numel()].view(p.shape), ga[n], rtol=1e-5, atol=1e-7)


def _reference_shaped_step(encoder, decoder, discriminator, opt_E, opt_Dec, opt_Dis, bce, mse):
    """vaegan_code.py:65-135 as a function of (real_images, eps_z, eps_real, eps_recon; epoch) returning device tensors."""
    def step(real_images, ez, er, ec, epoch=60):
        B = real_images.size(0)
        mu, logvar = encoder(real_images)
        logvar = torch.clamp(logvar, min=-10, max=10)
        std = torch.exp(0.5 * logvar)
        z = (mu + std * ez).unsqueeze(-1).unsqueeze(-1)
        recon_images = decoder(z)
        real_labels = torch.full((B,), 0.9, device=DEV)
        fake_labels = torch.full((B,), 0.1, device=DEV)
        real_images_noisy = real_images + 0.05 * er
        recon_images_noisy = recon_images + 0.05 * ec
        dl = []
        for _ in range(2):
            d_loss = bce(discriminator(real_images_noisy), real_labels) + bce(discriminator(recon_images_noisy.detach()), fake_labels)
            opt_Dis.zero_grad()
            d_loss.backward()
            opt_Dis.step()
            dl.append(d_loss.detach())
        fake_output = discriminator(recon_images_noisy)
        recon_loss = mse(recon_images, real_images)
        kl_loss = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp()) / B
        g_loss_adv = bce(fake_output, real_labels)
        total = recon_loss + 0.1 * min(1.0, epoch / 50) * kl_loss + 0.1 * g_loss_adv
        opt_E.zero_grad()
        opt_Dec.zero_grad()
        total.backward()
        opt_E.step()
        opt_Dec.step()
        return torch.stack([recon_loss.detach(), kl_loss.detach(), g_loss_adv.detach(), dl[0], dl[1]])
    return step


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_graphed_reference_shaped_step_equals_the_eager_one_and_the_oracle(dtype):
    """vaegan_amd.graphed(step): the reference's loop body (module calls, torch ops, nn.BCELoss / nn.MSELoss, .backward(),
    optimizer.step()) captured into ONE hipGraph -- autograd's backward included -- and replayed: bit-identical to the
    same function called eagerly (losses of 5 iterations, parameters, BatchNorm buffers, Adam state, host-side step /
    num_batches_tracked mirrors), first-iteration losses against the oracle (fp32: FIRST_STEP_TOL; bf16: 3e-2), a new
    keyword scalar (epoch -> KL weight) re-captures."""
    S, B = 64, 8
    res = []
    for use_graph in (False, True):
        V.configure_seed(42)
        nets = (V.Encoder([3, S, S], 100, dtype=dtype), V.Generator(nz=100, img_size=S, dtype=dtype),
                V.Discriminator(img_size=S, dtype=dtype))
        nets[1].apply(V.weights_init), nets[2].apply(V.weights_init)
        for m in nets:
            m.to(DEV), m.train()
        opts = tuple(V.Adam(m.parameters(), lr=2e-4) for m in nets)
        step = _reference_shaped_step(*nets, *opts, torch.nn.BCELoss(), torch.nn.MSELoss(reduction="mean"))
        if use_graph:
            step = V.graphed(step, modules=nets, optimizers=opts)
        losses = []
        for i in range(5):
            ins = [t.to(DEV) for t in make_inputs(B, S, 7300 + i)]
            losses.append(step(*ins, epoch=60).clone())
        losses.append(step(*[t.to(DEV) for t in make_inputs(B, S, 7306)], epoch=25).clone())     # other KL weight
        if use_graph:
            assert len(step._graphs) == 1 and sum(step._seen.values()) == 2 + 1     # epoch=25: still in its eager warm-up
        torch.cuda.synchronize()
        res.append((torch.stack(losses).cpu(), [{k: v.cpu() for k, v in m.state_dict().items()} for m in nets],
                    [(o.exp_avg.cpu(), o.exp_avg_sq.cpu(), o.steps, float(o.state_dev[0])) for o in opts]))
    assert torch.equal(res[0][0], res[1][0])
    for sa, sb in zip(res[0][1], res[1][1]):
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k
    for a, b in zip(res[0][2], res[1][2]):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2:] == b[2:]
    assert res[1][2][2][2] == 12 and res[1][2][0][2] == 6          # D stepped twice per iteration, E once
    ref = R.RefVAEGAN(img_size=S, seed=42).train_step(*make_inputs(B, S, 7300), 60)
    for j, n in enumerate(V.LOSS_NAMES):
        tol = FIRST_STEP_TOL[n] if dtype == "fp32" else 3e-2
        assert rel(float(res[1][0][0, j]), ref[n]) <= tol, f"graphed drop-in {dtype} {n}: {float(res[1][0][0, j])} oracle {ref[n]}"


def test_graphed_step_keeps_a_bounded_number_of_graphs_and_shares_equal_kl_weights():
    """step(x, epoch=epoch) must not retain one graph (and one private memory pool) per epoch.  Ten distinct
    epochs keep at most max_graphs entries; with scalar_key = the KL warm-up weight every epoch >= 50 replays ONE graph, and
    that replay is bit-identical to the eager function."""
    S, B = 64, 4
    res = []
    for mode in ("eager", "lru", "keyed"):
        V.configure_seed(42)
        nets = (V.Encoder([3, S, S], 100, dtype="bf16"), V.Generator(nz=100, img_size=S, dtype="bf16"),
                V.Discriminator(img_size=S, dtype="bf16"))
        nets[1].apply(V.weights_init), nets[2].apply(V.weights_init)
        for m in nets:
            m.to(DEV), m.train()
        opts = tuple(V.Adam(m.parameters(), lr=2e-4) for m in nets)
        step = _reference_shaped_step(*nets, *opts, torch.nn.BCELoss(), torch.nn.MSELoss(reduction="mean"))
        if mode == "lru":
            step = V.graphed(step, modules=nets, optimizers=opts, warmup=1)
        elif mode == "keyed":
            step = V.graphed(step, modules=nets, optimizers=opts, warmup=1, scalar_key=lambda epoch: min(1.0, epoch / 50))
        losses = []
        for epoch in range(50, 60):
            for rep in range(2):
                ins = [t.to(DEV) for t in make_inputs(B, S, 9100 + 2 * epoch + rep)]
                losses.append(step(*ins, epoch=epoch).clone())
        if mode == "lru":
            assert len(step._graphs) <= 2 and len(step._seen) <= 16
        if mode == "keyed":
            assert len(step._graphs) == 1 and len(step._seen) == 1
        torch.cuda.synchronize()
        res.append(torch.stack(losses).cpu())
    assert torch.equal(res[0], res[1]) and torch.equal(res[0], res[2])


def test_modules_losses_mirror_torch():
    p = torch.rand(7, device=DEV) * 0.98 + 0.01
    t = torch.full((7,), 0.9, device=DEV)
    p1 = p.clone().requires_grad_(True)
    p2 = p.clone().requires_grad_(True)
    l1 = V.BCELoss()(p1, t)
    l2 = torch.nn.BCELoss()(p2, t)
    l1.backward(), l2.backward()
    torch.testing.assert_close(l1, l2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(p1.grad, p2.grad, rtol=1e-5, atol=1e-6)
    a = torch.randn(2, 3, 8, 8, device=DEV, requires_grad=True)
    b = torch.randn(2, 3, 8, 8, device=DEV)
    l = V.MSELoss()(a, b)
    l.backward()
    torch.testing.assert_close(l, torch.nn.functional.mse_loss(a.detach(), b), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a.grad, 2 * (a.detach() - b) / a.numel(), rtol=1e-5, atol=1e-8)


def test_adam_state_dict_roundtrip_and_checkpoint_keys():
    e, g, d, tr = build(64)
    real, ez, er, ec = (t.to(DEV) for t in make_inputs(4, 64, 1))
    tr.train_step(real, 60, ez, er, ec)
    sd = tr.opt_G.state_dict()
    assert set(sd) == {"state", "param_groups"} and set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    ref = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in g.parameters()], lr=2e-4)
    ref.load_state_dict(sd)                       # torch.optim.Adam accepts our state_dict
    assert float(ref.state_dict()["state"][0]["step"]) == 1.0
    # decoder checkpoint (vaegan_code.py:193) loads into a fresh module with identical keys
    g2 = V.Generator(nz=100, img_size=64)
    g2.load_state_dict(g.state_dict())


def test_failed_graph_capture_leaves_the_trainer_where_it_was(monkeypatch):
    """A capture that dies half-way (here: the Generator's Adam step raises while the stream is capturing) has
    executed nothing: the trainer must come back un-captured, with its host counters restored, and the same four
    iterations run afterwards -- eagerly, then captured again -- must give bit for bit what an undisturbed run gives."""
    from importlib import import_module
    ops = import_module("vae-gan-based-model-for-image-generation-and-denoising_amd.ops")
    res = []
    for disturb in (False, True):
        e, g, d, tr = build(64)
        batches = [tuple(t.to(DEV) for t in make_inputs(4, 64, 7064 + step)) for step in range(4)]
        l = tr.train_step_graphed(batches[0][0], 60, *batches[0][1:])          # eager warm-up call
        if disturb:
            real_adam = ops.adam_apply2       # (round 4: the Encoder's and the Generator's step are one launch)

            def failing_adam(*a, **k):
                if torch.cuda.is_current_stream_capturing():           # after D's two steps of the capture
                    raise RuntimeError("injected failure during capture")
                return real_adam(*a, **k)

            monkeypatch.setattr(ops, "adam_apply2", failing_adam)
            steps_before = (tr.opt_E.steps, tr.opt_G.steps, tr.opt_D.steps)
            with pytest.raises(RuntimeError, match="injected failure"):
                tr.train_step_graphed(batches[1][0], 60, *batches[1][1:])
            monkeypatch.setattr(ops, "adam_apply2", real_adam)
            assert tr._graph is None and not torch.cuda.is_current_stream_capturing()
            assert (tr.opt_E.steps, tr.opt_G.steps, tr.opt_D.steps) == steps_before
            torch.cuda.synchronize()                                           # the device is usable
            l = tr.train_step(batches[1][0], 60, *batches[1][1:])              # the iteration that failed, eagerly
            for b in batches[2:]:
                l = tr.train_step_graphed(b[0], 60, *b[1:])                    # warm-up again, then capture + replay
            assert tr._graph is not None
        else:
            for b in batches[1:]:
                l = tr.train_step_graphed(b[0], 60, *b[1:])
        res.append((l[:5].cpu().clone(), tr.opt_E.flat_p.cpu().clone(), tr.opt_G.flat_p.cpu().clone(),
                    tr.opt_D.flat_p.cpu().clone(), (tr.opt_D.steps, float(tr.opt_D.state_dev[0]))))
    for a, b in zip(res[0][:4], res[1][:4]):
        assert torch.equal(a, b)
    assert res[0][4] == res[1][4] == (8, 8.0)


@pytest.mark.parametrize("segmented", [False, True])
def test_graph_replay_draws_fresh_device_noise(segmented):
    """Without injected noise the three randn draws of the iteration (vaegan_code.py:77,91,92) are made on the
    device inside the captured graph; every replay must see NEW noise (graph-safe Philox offsets), and the same
    seed must give the same sequence again.  lr = 0 keeps the weights fixed, so only the noise moves the losses."""
    real = make_inputs(8, 64, 77)[0].to(DEV)
    runs = []
    for _ in range(2):
        e, g, d, tr = build(64, lr=0.0)
        if segmented:                                   # the N>1 shape of the iteration: 5 hipGraph segments
            tr.reducer = _LocalReducer()
        torch.manual_seed(123)
        torch.cuda.manual_seed(123)
        runs.append(torch.stack([tr.train_step_graphed(real, 60)[:5].clone() for _ in range(5)]).cpu())
    a = runs[0]
    for i in range(5):
        for j in range(i + 1, 5):
            assert not torch.equal(a[i], a[j]), (i, j)              # eager, capture+replay, replays: all different
    assert float((a[:, 0].max() - a[:, 0].min()) / a[:, 0].mean()) < 0.2      # ...but the same distribution
    assert torch.equal(runs[0][2:], runs[1][2:])                     # replays reproduce under the same seed


@pytest.mark.parametrize("graphed", [False, True])
def test_checkpoint_resume_is_bitwise_identical(tmp_path, graphed):
    """save after 2 iterations -> a FRESH trainer loads the file (weights_only) -> the next 2 iterations equal the
    uninterrupted run bit for bit (parameters, BatchNorm buffers, Adam moments and step counts, losses)."""
    S, B = 64, 4
    ins = [tuple(t.to(DEV) for t in make_inputs(B, S, 700 + i)) for i in range(4)]

    def run(tr, idx):
        fn = tr.train_step_graphed if graphed else tr.train_step
        return [fn(ins[i][0], 60, *ins[i][1:])[:5].clone() for i in idx]

    e, g, d, tr = build(S)
    run(tr, [0, 1])
    path = str(tmp_path / "ck.pth")
    tr.save_checkpoint(path, epoch=60)
    la = run(tr, [2, 3])
    e2, g2, d2, tr2 = build(S)
    with torch.no_grad():
        for m in (e2, g2, d2):
            for p_ in m.parameters():
                p_.add_(1.0)                       # make sure the load is what restores the state
    assert tr2.load_checkpoint(path) == {"epoch": 60}
    lb = run(tr2, [2, 3])
    torch.cuda.synchronize()
    for a, b in zip(la, lb):
        assert torch.equal(a, b)
    for ma, mb in ((e, e2), (g, g2), (d, d2)):
        sa, sb = ma.state_dict(), mb.state_dict()
        assert list(sa) == list(sb)
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k
    for oa, ob in ((tr.opt_E, tr2.opt_E), (tr.opt_G, tr2.opt_G), (tr.opt_D, tr2.opt_D)):
        assert torch.equal(oa.exp_avg, ob.exp_avg) and torch.equal(oa.exp_avg_sq, ob.exp_avg_sq)
        assert oa.steps == ob.steps
    assert tr.opt_D.steps == 8 and tr.opt_E.steps == 4


@pytest.mark.parametrize("graphed", [False, True])
def test_checkpoint_resume_continues_the_device_noise_stream(tmp_path, graphed):
    """round-2 ADVICE: resume with the in-kernel randn draws (eps=None).  The checkpoint carries the noise stream's
    {seed, iteration}; a resuming process that is seeded DIFFERENTLY (or not at all) must continue that stream, not
    restart one from its own seed at iteration 0 -- only an explicit re-seed after the restore starts a new stream."""
    S, B = 64, 4
    real = [make_inputs(B, S, 710 + i)[0].to(DEV) for i in range(4)]

    def run(tr, idx):
        fn = tr.train_step_graphed if graphed else tr.train_step
        return [fn(real[i], 60)[:5].clone() for i in idx]

    e, g, d, tr = build(S)
    torch.cuda.manual_seed(555)
    run(tr, [0, 1])
    path = str(tmp_path / "ck_noise.pth")
    tr.save_checkpoint(path)
    la = run(tr, [2, 3])
    e2, g2, d2, tr2 = build(S)
    torch.cuda.manual_seed(99)                          # the resuming process has another device seed
    tr2.load_checkpoint(path)
    lb = run(tr2, [2, 3])
    for a, b in zip(la, lb):
        assert torch.equal(a, b)
    torch.cuda.manual_seed(1234)                        # an explicit re-seed AFTER the restore does start a new stream
    lc = run(tr2, [2])
    assert tr2.noise.seed == 1234 and not torch.equal(lc[0], la[0])


@pytest.mark.parametrize("S,sigma", [(64, 0.2), (64, 0.05), (128, 0.2)])
def test_denoise_eval_path_vs_oracle(S, sigma):
    """BASELINE config 4 / vaegan_code.py:147-171: eval-mode E -> reparam -> G on clamp(img + sigma*eps, -1, 1),
    MSE + KL(sum), PSNR and SSIM (oracle restatement of the torchmetrics SSIM recipe; parity unpinned)."""
    B = 8
    e, g, d, tr = build(S)
    o = R.RefVAEGAN(img_size=S, seed=42)
    real, ez, er, ec = make_inputs(B, S, 7000 + S)
    o.train_step(real, ez, er, ec, 60)                        # one training step so BN running stats are non-trivial
    tr.train_step(real.to(DEV), 60, ez.to(DEV), er.to(DEV), ec.to(DEV))
    sync_from_oracle(o, e, g, d, tr)                          # identical weights / buffers for the eval comparison
    e.eval(), g.eval()
    img, ez2, noise, _ = make_inputs(B, S, 31337 + S)
    noisy_ref, recon_ref, rl_ref, kl_ref = o.denoise(img, sigma * noise, ez2)
    out = V.denoise_eval(e, g, img.to(DEV), sigma=sigma, eps=noise.to(DEV), eps_z=ez2.to(DEV))
    torch.testing.assert_close(out["noisy"].cpu(), noisy_ref, rtol=0, atol=1e-6)
    torch.testing.assert_close(out["recon"].cpu(), recon_ref, rtol=1e-3, atol=2e-5)
    assert rel(out["recon_loss"], rl_ref) < 1e-4 and rel(out["kl_loss"], kl_ref) < 1e-4
    a01, b01 = (recon_ref + 1) / 2, (img + 1) / 2
    assert abs(out["psnr"] - R.psnr(a01, b01)) < 1e-3
    assert abs(out["ssim"] - R.ssim(a01, b01)) < 1e-4


@pytest.mark.gpu
def test_garbage_collection_cannot_run_inside_a_graph_capture():
    """A cyclic collection in the middle of a capture can destroy an older trainer's CUDAGraph (its private pool frees
    device memory: not permitted while the thread captures -> the process aborts inside a destructor; seen once as
    'Fatal Python error: Aborted ... Garbage-collecting' under train_step_graphed).  The capture collects first and keeps
    the collector off until capture_end: with an older captured trainer left behind in a reference cycle and the
    collector set to fire on every allocation, a second trainer still captures, replays and equals its eager twin."""
    import gc
    x = torch.randn(4, 3, 64, 64, generator=torch.Generator().manual_seed(5)).clamp(-1, 1).to(DEV)
    V.configure_seed(42)
    _, _, _, old = build(64, dtype="bf16")
    for _ in range(3):
        old.train_step_graphed(x, 60)
    torch.cuda.synchronize()
    old._cycle = old                       # only the cyclic collector can free it (and its graph) from now on
    del old
    thr = gc.get_threshold()
    gc.set_threshold(1, 1, 1)
    try:
        V.configure_seed(42)
        _, _, _, tr = build(64, dtype="bf16")
        got = torch.stack([tr.train_step_graphed(x, 60)[:5].clone() for _ in range(3)]).cpu()
        assert tr._graph is not None and gc.isenabled()
    finally:
        gc.set_threshold(*thr)
    V.configure_seed(42)
    _, _, _, ref = build(64, dtype="bf16")
    want = torch.stack([ref.train_step(x, 60)[:5].clone() for _ in range(3)]).cpu()
    assert torch.equal(got, want)

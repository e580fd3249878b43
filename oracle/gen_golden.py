"""Generate tests/golden/*.npz from the reference's OWN classes (build container only).

TEST INFRASTRUCTURE.  Run:  python oracle/gen_golden.py
Needs /root/reference (read-only); imports its hot-path classes through the stub recipe in
``oracle/load_reference.py`` and restates the training iteration of vaegan_code.py:65-135
around them with stock ``torch.optim.Adam`` / ``nn.BCELoss`` / ``nn.MSELoss`` exactly as
vaegan_code.py:42-47 builds them.  Only inputs/outputs (small arrays, checksums) are stored --
never reference source.  While generating, every vector is also recomputed with the CPU
oracle (``oracle/vaegan_ref.py``) and must match bit-for-bit at S=256 (same ATen kernels).
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import vaegan_ref as R                      # noqa: E402
from load_reference import load_reference   # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


# ---- deterministic inputs shared with the tests (tests/_inputs.py restates these) -------------
def make_inputs(B, S, seed, latent=100):
    g = torch.Generator().manual_seed(seed)
    real = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    eps_z = torch.randn(B, latent, generator=g)
    eps_real = torch.randn(B, 3, S, S, generator=g)
    eps_recon = torch.randn(B, 3, S, S, generator=g)
    return real, eps_z, eps_real, eps_recon


def tstats(t):
    t = t.detach().double().flatten()
    idx = torch.linspace(0, t.numel() - 1, min(16, t.numel())).long()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()]), t[idx].numpy()


def state_stats(sd):
    out = {}
    for k, v in sd.items():
        s, samp = tstats(v.float() if v.dtype != torch.float32 else v)
        out[k + "#stats"] = s
        out[k + "#samp"] = samp
    return out


def ref_step(ref, E, G, D, oE, oG, oD, real, eps_z, eps_real, eps_recon, epoch):
    """vaegan_code.py:65-135 around the reference modules; randn_like draws injected."""
    bce, mse = nn.BCELoss(), nn.MSELoss(reduction="mean")
    B = real.size(0)
    mu, logvar = E(real)
    logvar = torch.clamp(logvar, min=-10, max=10)
    std = torch.exp(0.5 * logvar)
    z = (mu + std * eps_z).unsqueeze(-1).unsqueeze(-1)
    recon = G(z)
    real_labels = torch.full((B,), 0.9)
    fake_labels = torch.full((B,), 0.1)
    real_noisy = real + 0.05 * eps_real
    recon_noisy = recon + 0.05 * eps_recon
    dl = []
    for _ in range(2):
        d_loss = bce(D(real_noisy), real_labels) + bce(D(recon_noisy.detach()), fake_labels)
        oD.zero_grad()
        d_loss.backward()
        oD.step()
        dl.append(float(d_loss))
    fake_out = D(recon_noisy)
    recon_loss = mse(recon, real)
    kl_loss = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp()) / B
    adv = bce(fake_out, real_labels)
    total = recon_loss + 0.1 * min(1.0, epoch / 50) * kl_loss + 0.1 * adv
    oE.zero_grad()
    oG.zero_grad()
    total.backward()
    oE.step()
    oG.step()
    return dict(recon_loss=float(recon_loss), kl_loss=float(kl_loss), g_loss_adv=float(adv),
                d_loss_1=dl[0], d_loss_2=dl[1], total=float(total)), recon.detach()


def build_ref(ref, S=256):
    ref.configure_seed(42)
    E = ref.Encoder([3, S, S], 100)
    G = ref.Generator(nz=100)
    D = ref.Discriminator()
    G.apply(ref.weights_init)
    D.apply(ref.weights_init)
    return E, G, D


def assert_same_state(sd_ref, st_oracle, what):
    for k, v in sd_ref.items():
        o = st_oracle[k]
        if not torch.equal(v.detach(), o.detach()):
            raise AssertionError(f"oracle != reference for {what}:{k} maxdiff "
                                 f"{(v.detach().double() - o.detach().double()).abs().max().item():g}")


def gen_init(ref):
    """G4: seed-42 construction-order parameter statistics + Encoder dummy-forward BN buffers."""
    E, G, D = build_ref(ref)
    out = {}
    for name, m in (("E", E), ("G", G), ("D", D)):
        for k, v in state_stats(m.state_dict()).items():
            out[f"{name}.{k}"] = v
    fz = torch.randn(64, 100, 1, 1)                      # vaegan_code.py:40 (consumes RNG after init)
    out["fixed_noise#samp"] = fz.flatten()[:8].numpy()
    # Encoder BN buffers after construction at the three sizes -- stored in full (small)
    for S in (64, 128, 256):
        ref.configure_seed(42)
        e = ref.Encoder([3, S, S], 100)
        out[f"E{S}.flatten_size"] = np.array([e.flatten_size])
        for i in range(4):
            out[f"E{S}.cnn.{i}.bn.running_mean"] = e.cnn[i].bn.running_mean.numpy().copy()
            out[f"E{S}.cnn.{i}.bn.running_var"] = e.cnn[i].bn.running_var.numpy().copy()
            out[f"E{S}.cnn.{i}.conv.bias"] = e.cnn[i].conv.bias.detach().numpy().copy()
            out[f"E{S}.cnn.{i}.bn.num_batches_tracked"] = np.array([int(e.cnn[i].bn.num_batches_tracked)])
    # oracle check
    o = R.RefVAEGAN(img_size=256, seed=42)
    assert_same_state(E.state_dict(), o.E, "E")
    assert_same_state(G.state_dict(), o.G, "G")
    assert_same_state(D.state_dict(), o.D, "D")
    np.savez_compressed(os.path.join(OUT, "init_seed42.npz"), **out)
    print("init_seed42.npz", len(out), "arrays")


def gen_forward(ref):
    """G1: forward KATs, train then eval mode, for each net (E at S=64/128/256; G, D at 256)."""
    out = {}
    for S in (64, 128, 256):
        ref.configure_seed(42)
        e = ref.Encoder([3, S, S], 100)
        real, _, _, _ = make_inputs(2, S, 1000 + S)
        e.train()
        mu, lv = e(real)
        out[f"E{S}.train.mu"], out[f"E{S}.train.logvar"] = mu.detach().numpy(), lv.detach().numpy()
        e.eval()
        with torch.no_grad():
            mu2, lv2 = e(real)
        out[f"E{S}.eval.mu"], out[f"E{S}.eval.logvar"] = mu2.numpy(), lv2.numpy()
        R.configure_seed(42)
        st = R.make_encoder_state([3, S, S], 100)
        m, l = R.encoder_forward(st, real, True)
        assert torch.equal(m, mu.detach()) and torch.equal(l, lv.detach()), f"oracle E{S} train"
        m, l = R.encoder_forward(st, real, False)
        assert torch.equal(m, mu2) and torch.equal(l, lv2), f"oracle E{S} eval"
    E, G, D = build_ref(ref)
    o = R.RefVAEGAN(img_size=256, seed=42)
    g = torch.Generator().manual_seed(77)
    z = torch.randn(2, 100, 1, 1, generator=g)
    G.train()
    img = G(z).detach()
    D.train()
    p = D(img).detach()
    out["G256.train.stats"], out["G256.train.samp"] = tstats(img)
    out["D256.train.out"] = p.numpy()
    assert torch.equal(R.generator_forward(o.G, o.g_spec, z, True).detach(), img)
    assert torch.equal(R.discriminator_forward(o.D, o.d_spec, img, True).detach(), p)
    G.eval(), D.eval()
    with torch.no_grad():
        img2 = G(z)
        p2 = D(img)
    out["G256.eval.stats"], out["G256.eval.samp"] = tstats(img2)
    out["D256.eval.out"] = p2.numpy()
    with torch.no_grad():
        assert torch.equal(R.generator_forward(o.G, o.g_spec, z, False), img2)
        assert torch.equal(R.discriminator_forward(o.D, o.d_spec, img, False), p2)
    np.savez_compressed(os.path.join(OUT, "forward_kat.npz"), **out)
    print("forward_kat.npz", len(out), "arrays")


def gen_steps(ref):
    """G3: three consecutive training steps, S=256, for (B, epoch) in {(2,25), (2,60), (4,0)}."""
    for B, epoch in ((2, 25), (2, 60), (4, 0)):
        E, G, D = build_ref(ref)
        oE = optim.Adam(E.parameters(), lr=2e-4)
        oG = optim.Adam(G.parameters(), lr=2e-4)
        oD = optim.Adam(D.parameters(), lr=2e-4)
        E.train(), G.train(), D.train()
        o = R.RefVAEGAN(img_size=256, seed=42)
        out = {"B": np.array([B]), "epoch": np.array([epoch])}
        names = ["recon_loss", "kl_loss", "g_loss_adv", "d_loss_1", "d_loss_2", "total"]
        losses = []
        for step in range(3):
            real, ez, er, ec = make_inputs(B, 256, 5000 + 10 * B + step)
            lr_, recon = ref_step(ref, E, G, D, oE, oG, oD, real, ez, er, ec, epoch)
            lo = o.train_step(real, ez, er, ec, epoch)
            for n in names:
                assert lr_[n] == lo[n], f"oracle step {step} {n}: {lr_[n]} vs {lo[n]}"
            losses.append([lr_[n] for n in names])
            out[f"step{step}.recon#stats"], out[f"step{step}.recon#samp"] = tstats(recon)
            out[f"step{step}.real#stats"], _ = tstats(real)
            if step == 0:
                # state after the FIRST iteration (not yet chaotic: one Adam step of E and G, two of D, all from the
                # seed-42 initial state): parameters, BatchNorm buffers and Adam moments.  exp_avg after one step is
                # (1-beta1)*grad, so these checksums pin the whole backward pass of iteration 1 against the reference.
                for name, m in (("E", E), ("G", G), ("D", D)):
                    for k, v in state_stats(m.state_dict()).items():
                        out[f"after1.{name}.{k}"] = v
                for name, opt in (("E", oE), ("G", oG), ("D", oD)):
                    ss = opt.state_dict()["state"]
                    for i in sorted(ss):
                        out[f"after1.adam.{name}.{i}.exp_avg#stats"], _ = tstats(ss[i]["exp_avg"])
                        out[f"after1.adam.{name}.{i}.exp_avg_sq#stats"], _ = tstats(ss[i]["exp_avg_sq"])
        assert_same_state(E.state_dict(), o.E, "E after steps")
        assert_same_state(G.state_dict(), o.G, "G after steps")
        assert_same_state(D.state_dict(), o.D, "D after steps")
        out["loss_names"] = np.array(names)
        out["losses"] = np.array(losses, dtype=np.float64)
        for name, m in (("E", E), ("G", G), ("D", D)):
            for k, v in state_stats(m.state_dict()).items():
                out[f"final.{name}.{k}"] = v
        for name, opt in (("E", oE), ("G", oG), ("D", oD)):
            ss = opt.state_dict()["state"]
            out[f"final.adam.{name}.step"] = np.array([float(ss[0]["step"])])
            s, _ = tstats(torch.cat([ss[i]["exp_avg"].flatten() for i in sorted(ss)]))
            out[f"final.adam.{name}.exp_avg#stats"] = s
            s, _ = tstats(torch.cat([ss[i]["exp_avg_sq"].flatten() for i in sorted(ss)]))
            out[f"final.adam.{name}.exp_avg_sq#stats"] = s
        fn = f"steps_S256_B{B}_e{epoch}.npz"
        np.savez_compressed(os.path.join(OUT, fn), **out)
        print(fn, out["losses"])


def gen_adam():
    """G5: single-tensor Adam KAT over 5 steps vs torch.optim.Adam (third-party, torch 2.10)."""
    g = torch.Generator().manual_seed(9)
    p0 = torch.randn(257, generator=g)
    grads = [torch.randn(257, generator=g) * (10.0 ** (i - 2)) for i in range(5)]
    p = p0.clone().requires_grad_(True)
    opt = optim.Adam([p], lr=2e-4)
    q = p0.clone().requires_grad_(True)
    ro = R.RefAdam([q], lr=2e-4)
    traj = []
    for gr in grads:
        p.grad = gr.clone()
        q.grad = gr.clone()
        opt.step()
        ro.step()
        assert torch.equal(p.detach(), q.detach())
        traj.append(p.detach().numpy().copy())
    st = opt.state_dict()["state"][0]
    np.savez_compressed(os.path.join(OUT, "adam_kat.npz"), p0=p0.numpy(), grads=np.stack([g_.numpy() for g_ in grads]),
                        traj=np.stack(traj), exp_avg=st["exp_avg"].numpy(), exp_avg_sq=st["exp_avg_sq"].numpy())
    print("adam_kat.npz")


def gen_oracle_only():
    """Regression vectors for the A0-derived sizes (S=64/128): produced by the ORACLE, not by the
    reference (no runnable reference geometry exists) -- marked 'oracle-generated'."""
    names = ["recon_loss", "kl_loss", "g_loss_adv", "d_loss_1", "d_loss_2", "total"]
    for S, B in ((64, 4), (128, 2)):
        o = R.RefVAEGAN(img_size=S, seed=42)
        losses = []
        for step in range(3):
            real, ez, er, ec = make_inputs(B, S, 7000 + S + step)
            lo = o.train_step(real, ez, er, ec, 60)
            losses.append([lo[n] for n in names])
        np.savez_compressed(os.path.join(OUT, f"oracle_steps_S{S}_B{B}_e60.npz"),
                            provenance=np.array(["oracle-generated (no reference geometry at this size)"]),
                            loss_names=np.array(names), losses=np.array(losses, dtype=np.float64))
        print(f"oracle_steps_S{S}", np.array(losses))


if __name__ == "__main__":
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    ref = load_reference()
    gen_init(ref)
    gen_forward(ref)
    gen_adam()
    gen_steps(ref)
    gen_oracle_only()
    print("all golden vectors written; oracle == reference bit-for-bit on every S=256 vector")

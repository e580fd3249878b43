"""Generate tests/golden/sibling_*.npz from the reference's OWN classes (build container only).

TEST INFRASTRUCTURE.  Run:  python oracle/gen_golden_siblings.py
The reference's ``train_vae`` / ``train_gan`` / ``train_wgan`` cannot run as written (hard-coded Windows dataset
path, torchvision / torchmetrics absent), so -- as gen_golden.py does for vaegan_code.py -- their loop bodies
(main_vae.py:103-127, gan_code.py:194-219, gan_code.py:296-331) are restated here around the reference's imported
``Encoder`` / ``Generator`` / ``Discriminator`` / ``weights_init`` with stock ``torch.optim.Adam`` /
``nn.BCELoss`` / ``nn.MSELoss`` exactly as those functions build them, with the random draws injected.
Only inputs/outputs (losses, checksums) are stored.  Every vector is recomputed with ``oracle/siblings_ref.py``
and must match bit for bit (same ATen CPU kernels at S=256).
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import siblings_ref as SR                        # noqa: E402
from gen_golden import OUT, assert_same_state, state_stats   # noqa: E402
from load_reference import load_reference        # noqa: E402

S, B, STEPS = 256, 2, 2


def sib_inputs(step, critic_iters=5, nz=100):
    g = torch.Generator().manual_seed(8100 + step)
    real = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    eps_img = torch.randn(B, 3, S, S, generator=g)
    eps_z = torch.randn(B, nz, generator=g)
    noises = [torch.randn(B, nz, 1, 1, generator=g) for _ in range(critic_iters + 1)]
    return real, eps_img, eps_z, noises


def adam_stats(out, name, opt):
    ss = opt.state_dict()["state"]
    out[f"final.adam.{name}.step"] = np.array([float(ss[0]["step"])])
    for key in ("exp_avg", "exp_avg_sq"):
        t = torch.cat([ss[i][key].flatten() for i in sorted(ss)]).double()
        out[f"final.adam.{name}.{key}#stats"] = np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def gen_vae(ref):
    ref.configure_seed(42)
    E, G = ref.Encoder([3, S, S], 100), ref.Generator(nz=100)        # main_vae.py:82-83 (no weights_init)
    opt = optim.Adam(list(E.parameters()) + list(G.parameters()), lr=1e-3)
    mse = nn.MSELoss(reduction="mean")
    E.train(), G.train()
    o = SR.RefVAE(img_size=S, seed=42)
    names = ["recon_loss", "kl_loss", "total"]
    losses = []
    for epoch, step in ((25, 0), (60, 1)):
        img, eps_img, eps_z, _ = sib_inputs(step)
        noisy = torch.clamp(img + eps_img * 0.5, -1.0, 1.0)
        mu, logvar = E(noisy)
        logvar = torch.clamp(logvar, min=-10, max=10)
        std = torch.exp(0.5 * logvar)
        z = (mu + std * eps_z).unsqueeze(-1).unsqueeze(-1)
        recon = G(z)
        recon_loss = mse(recon, img)
        kl_loss = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp())
        total = recon_loss + kl_loss * min(epoch / 50, 1.0) * 1e-5
        opt.zero_grad()
        total.backward()
        opt.step()
        got = {"recon_loss": recon_loss.item(), "kl_loss": kl_loss.item(), "total": total.item()}
        lo = o.train_step(img, eps_img, eps_z, epoch)
        for n in names:
            assert got[n] == lo[n], f"VAE step {step} {n}: {got[n]} vs {lo[n]}"
        losses.append([got[n] for n in names])
    assert_same_state(E.state_dict(), o.E, "VAE E")
    assert_same_state(G.state_dict(), o.G, "VAE G")
    out = {"loss_names": np.array(names), "losses": np.array(losses, dtype=np.float64), "epochs": np.array([25, 60])}
    for name, m in (("E", E), ("G", G)):
        for k, v in state_stats(m.state_dict()).items():
            out[f"final.{name}.{k}"] = v
    adam_stats(out, "EG", opt)
    np.savez_compressed(os.path.join(OUT, "sibling_vae_S256_B2.npz"), **out)
    print("sibling_vae", out["losses"])


def build_gan(ref):
    ref.configure_seed(42)
    G, D = ref.Generator(nz=100), ref.Discriminator()
    G.apply(ref.weights_init), D.apply(ref.weights_init)
    oD = optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    oG = optim.Adam(G.parameters(), lr=2e-4, betas=(0.5, 0.999))
    return G, D, oG, oD


def finish_gan(fn, names, losses, G, D, oG, oD, o):
    assert_same_state(G.state_dict(), o.G, fn + " G")
    assert_same_state(D.state_dict(), o.D, fn + " D")
    out = {"loss_names": np.array(names), "losses": np.array(losses, dtype=np.float64)}
    for name, m in (("G", G), ("D", D)):
        for k, v in state_stats(m.state_dict()).items():
            out[f"final.{name}.{k}"] = v
    adam_stats(out, "G", oG), adam_stats(out, "D", oD)
    np.savez_compressed(os.path.join(OUT, fn), **out)
    print(fn, out["losses"])


def gen_dcgan(ref):
    G, D, oG, oD = build_gan(ref)
    crit = nn.BCELoss()
    o = SR.RefDCGAN(img_size=S, seed=42)
    names = ["errD_real", "errD_fake", "errG"]
    losses = []
    for step in range(STEPS):
        real, _, _, noises = sib_inputs(step)
        D.zero_grad()
        label = torch.full((B,), 1.)
        errD_real = crit(D(real), label)
        errD_real.backward()
        fake = G(noises[0])
        label.fill_(0.)
        errD_fake = crit(D(fake.detach()), label)
        errD_fake.backward()
        oD.step()
        G.zero_grad()
        label.fill_(1.)
        errG = crit(D(fake), label)
        errG.backward()
        oG.step()
        got = {"errD_real": errD_real.item(), "errD_fake": errD_fake.item(), "errG": errG.item()}
        lo = o.train_step(real, noises[0])
        for n in names:
            assert got[n] == lo[n], f"DCGAN step {step} {n}: {got[n]} vs {lo[n]}"
        losses.append([got[n] for n in names])
    finish_gan("sibling_dcgan_S256_B2.npz", names, losses, G, D, oG, oD, o)


def gen_wgan(ref):
    G, D, oG, oD = build_gan(ref)
    o = SR.RefWGAN(img_size=S, seed=42)
    names = ["d_loss", "g_loss"]
    losses = []
    for step in range(STEPS):
        real, _, _, noises = sib_inputs(step)
        for it in range(5):
            D.zero_grad()
            d_loss_real = -D(real).mean()
            fake = G(noises[it]).detach()
            d_loss_fake = D(fake).mean()
            d_loss = d_loss_real + d_loss_fake
            d_loss.backward()
            oD.step()
            for p in D.parameters():
                p.data.clamp_(-0.01, 0.01)
        G.zero_grad()
        g_loss = -D(G(noises[5])).mean()
        g_loss.backward()
        oG.step()
        got = {"d_loss": d_loss.item(), "g_loss": g_loss.item()}
        lo = o.train_step(real, noises[:5], noises[5])
        for n in names:
            assert got[n] == lo[n], f"WGAN step {step} {n}: {got[n]} vs {lo[n]}"
        losses.append([got[n] for n in names])
    finish_gan("sibling_wgan_S256_B2.npz", names, losses, G, D, oG, oD, o)


if __name__ == "__main__":
    torch.set_num_threads(8)
    ref = load_reference()
    gen_vae(ref)
    gen_dcgan(ref)
    gen_wgan(ref)

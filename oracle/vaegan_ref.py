"""CPU oracle: restatement of the reference VAE-GAN training path in plain PyTorch (CPU, fp32).

TEST INFRASTRUCTURE.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this file -- and only as the checker / reported baseline.
The product package never imports it and has no CPU fallback.

Parity pin: every function below is checked against the reference's *own* classes
(imported in the build container with ``oracle/load_reference.py``) by
``oracle/gen_golden.py``; the resulting vectors live in ``tests/golden/*.npz`` and are
re-checked on CPU by ``tests/test_oracle_golden.py``.  At S=256 the oracle runs the same
ATen CPU kernels as the reference, so those checks are bit-exact.  The S=64 / S=128
members of the family (size rule A0) have no runnable reference
geometry (gan_code.py hard-wires 256x256) -- for those, only the Encoder is pinned by
reference code; G_S / D_S are pinned through the shared layer code exercised at S=256.

State is kept in plain ``OrderedDict[str, Tensor]`` objects whose keys are the reference's
``state_dict`` keys; forward passes are functional.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

BN_EPS = 1e-5        # nn.BatchNorm2d default (main_vae.py:24, gan_code.py:22)
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------------------
# Architecture specs.  A spec is the list of entries of the nn.Sequential in the reference;
# the list index is the Sequential index and therefore the state_dict key prefix.
# --------------------------------------------------------------------------------------

def encoder_channels(in_ch: int) -> List[int]:
    return [in_ch, 32, 64, 128, 256]                       # main_vae.py:37


def encoder_feature_hw(size: int) -> List[int]:
    """Spatial sizes after each ConvBlock (k=4, s=2, p=0): main_vae.py:21-23."""
    out = []
    for _ in range(4):
        size = (size - 4) // 2 + 1
        out.append(size)
    return out


def generator_spec(nz: int = 128, ngf: int = 64, nc: int = 3, img_size: int = 256):
    """gan_code.py:19-51 for img_size=256; rule A0 drops the last n=log2(256/S) stride-2 stages."""
    n_drop = int(round(math.log2(256 // img_size)))
    assert 256 // (2 ** n_drop) == img_size and 0 <= n_drop <= 4
    chans = [ngf * 16, ngf * 8, ngf * 4, ngf * 2, ngf, ngf // 2, ngf // 4]   # outputs of the 7 convT stages
    chans = chans[: len(chans) - n_drop]
    spec = [("convT", nz, chans[0], 4, 1, 0), ("bn", chans[0]), ("relu",)]
    for i in range(1, len(chans)):
        spec += [("convT", chans[i - 1], chans[i], 4, 2, 1), ("bn", chans[i]), ("relu",)]
    spec += [("convT", chans[-1], nc, 3, 1, 1), ("tanh",)]
    return spec


def discriminator_spec(ndf: int = 64, nc: int = 3, img_size: int = 256):
    """gan_code.py:59-86 for img_size=256; rule A0 drops the first n stages (first conv keeps no BN)."""
    n_drop = int(round(math.log2(256 // img_size)))
    assert 256 // (2 ** n_drop) == img_size and 0 <= n_drop <= 4
    chans = [ndf // 4, ndf // 2, ndf, ndf * 2, ndf * 4, ndf * 8]           # outputs of the 6 stride-2 convs
    chans = chans[n_drop:]
    spec = [("conv", nc, chans[0], 4, 2, 1), ("lrelu", 0.2)]
    for i in range(1, len(chans)):
        spec += [("conv", chans[i - 1], chans[i], 4, 2, 1), ("bn", chans[i]), ("lrelu", 0.2)]
    spec += [("conv", chans[-1], 1, 4, 1, 0), ("sigmoid",)]
    return spec


# --------------------------------------------------------------------------------------
# Construction (consumes the global torch RNG in the reference's order)
# --------------------------------------------------------------------------------------

def _bn_state(prefix: str, c: int, st: "OrderedDict[str, torch.Tensor]") -> None:
    bn = nn.BatchNorm2d(c)
    for k, v in bn.state_dict().items():
        st[f"{prefix}.{k}"] = v.clone()


def make_sequential_state(spec) -> "OrderedDict[str, torch.Tensor]":
    """Default-initialised state of an nn.Sequential described by ``spec`` (bias=False convs)."""
    st: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for i, ent in enumerate(spec):
        if ent[0] == "conv":
            _, cin, cout, k, s, p = ent
            st[f"main.{i}.weight"] = nn.Conv2d(cin, cout, k, s, p, bias=False).weight.detach().clone()
        elif ent[0] == "convT":
            _, cin, cout, k, s, p = ent
            st[f"main.{i}.weight"] = nn.ConvTranspose2d(cin, cout, k, s, p, bias=False).weight.detach().clone()
        elif ent[0] == "bn":
            _bn_state(f"main.{i}", ent[1], st)
    return st


def weights_init_state(st: "OrderedDict[str, torch.Tensor]", spec) -> None:
    """gan_code.py:91-97 applied via ``.apply`` (children in Sequential order)."""
    for i, ent in enumerate(spec):
        if ent[0] in ("conv", "convT"):
            nn.init.normal_(st[f"main.{i}.weight"], 0.0, 0.02)
        elif ent[0] == "bn":
            nn.init.normal_(st[f"main.{i}.weight"], 1.0, 0.02)
            nn.init.constant_(st[f"main.{i}.bias"], 0)


def make_encoder_state(img_size: Sequence[int], latent_dim: int):
    """main_vae.py:35-48 including the train-mode dummy forward of :43-44 (BN buffer side effect)."""
    ch = encoder_channels(img_size[0])
    st: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for i in range(4):
        conv = nn.Conv2d(ch[i], ch[i + 1], 4, 2)
        st[f"cnn.{i}.conv.weight"] = conv.weight.detach().clone()
        st[f"cnn.{i}.conv.bias"] = conv.bias.detach().clone()
        _bn_state(f"cnn.{i}.bn", ch[i + 1], st)
    with torch.no_grad():
        feat = _encoder_cnn(st, torch.zeros(1, img_size[0], img_size[1], img_size[2]), train=True)
    flatten = feat.reshape(1, -1).size(1)
    for name in ("fc_mu", "fc_logvar"):
        fc = nn.Linear(flatten, latent_dim)
        st[f"{name}.weight"] = fc.weight.detach().clone()
        st[f"{name}.bias"] = fc.bias.detach().clone()
    return st


def trainable_keys(st) -> List[str]:
    return [k for k in st if not (k.endswith("running_mean") or k.endswith("running_var")
                                  or k.endswith("num_batches_tracked"))]


def require_grads(st) -> List[torch.Tensor]:
    ps = []
    for k in trainable_keys(st):
        st[k].requires_grad_(True)
        ps.append(st[k])
    return ps


# --------------------------------------------------------------------------------------
# Forward passes
# --------------------------------------------------------------------------------------

def _bn(st, prefix: str, x: torch.Tensor, train: bool) -> torch.Tensor:
    if train:
        st[f"{prefix}.num_batches_tracked"] += 1
    return F.batch_norm(x, st[f"{prefix}.running_mean"], st[f"{prefix}.running_var"],
                        st[f"{prefix}.weight"], st[f"{prefix}.bias"], train, BN_MOMENTUM, BN_EPS)


def _encoder_cnn(st, x, train: bool):
    for i in range(4):                                              # ConvBlock.forward, main_vae.py:27-31
        x = F.conv2d(x, st[f"cnn.{i}.conv.weight"], st[f"cnn.{i}.conv.bias"], stride=2)
        x = _bn(st, f"cnn.{i}.bn", x, train)
        x = F.leaky_relu(x, 0.01)
    return x


def encoder_forward(st, x, train: bool = True):
    """Encoder.forward, main_vae.py:50-58."""
    h = _encoder_cnn(st, x, train)
    h = h.reshape(h.size(0), -1)
    mu = F.linear(h, st["fc_mu.weight"], st["fc_mu.bias"])
    logvar = F.linear(h, st["fc_logvar.weight"], st["fc_logvar.bias"])
    return mu, logvar


def sequential_forward(st, spec, x, train: bool = True):
    """nn.Sequential forward of Generator (gan_code.py:53-54) / Discriminator (:88-89, without .view)."""
    for i, ent in enumerate(spec):
        kind = ent[0]
        if kind == "conv":
            x = F.conv2d(x, st[f"main.{i}.weight"], None, stride=ent[4], padding=ent[5])
        elif kind == "convT":
            x = F.conv_transpose2d(x, st[f"main.{i}.weight"], None, stride=ent[4], padding=ent[5])
        elif kind == "bn":
            x = _bn(st, f"main.{i}", x, train)
        elif kind == "relu":
            x = F.relu(x)
        elif kind == "lrelu":
            x = F.leaky_relu(x, ent[1])
        elif kind == "tanh":
            x = torch.tanh(x)
        elif kind == "sigmoid":
            x = torch.sigmoid(x)
        else:
            raise ValueError(kind)
    return x


def generator_forward(st, spec, z, train: bool = True):
    return sequential_forward(st, spec, z, train)


def discriminator_forward(st, spec, x, train: bool = True):
    return sequential_forward(st, spec, x, train).view(-1)          # gan_code.py:89


# --------------------------------------------------------------------------------------
# Losses
# --------------------------------------------------------------------------------------

def bce_loss(p: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """nn.BCELoss() (vaegan_code.py:46): mean of -[t*max(log p,-100) + (1-t)*max(log(1-p),-100)]."""
    return F.binary_cross_entropy(p, t)


def mse_loss(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return F.mse_loss(a, b, reduction="mean")                       # vaegan_code.py:47


def kl_sum(mu: torch.Tensor, logvar: torch.Tensor) -> torch.Tensor:
    return -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp())  # vaegan_code.py:114 numerator


# --------------------------------------------------------------------------------------
# Adam (torch 2.10 ``_single_tensor_adam`` op sequence)
# --------------------------------------------------------------------------------------

class RefAdam:
    def __init__(self, params: Sequence[torch.Tensor], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        self.params = list(params)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.t = 0
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]

    def zero_grad(self) -> None:
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self) -> None:
        self.t += 1
        b1, b2 = self.betas
        bc1 = 1 - b1 ** self.t
        bc2 = 1 - b2 ** self.t
        step_size = self.lr / bc1
        bc2_sqrt = bc2 ** 0.5
        for p, m, v in zip(self.params, self.exp_avg, self.exp_avg_sq):
            if p.grad is None:
                continue
            g = p.grad
            m.lerp_(g, 1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / bc2_sqrt).add_(self.eps)
            p.addcdiv_(m, denom, value=-step_size)


# --------------------------------------------------------------------------------------
# The training step (vaegan_code.py:65-135) with host-injected noise
# --------------------------------------------------------------------------------------

class RefVAEGAN:
    """Holds E/G/D state + three Adam optimisers exactly as vaegan_code.py:29-44 builds them."""

    def __init__(self, img_size: int = 256, latent_dim: int = 100, in_ch: int = 3, lr: float = 2e-4,
                 seed: Optional[int] = 42):
        if seed is not None:
            configure_seed(seed)
        self.img_size, self.latent_dim = img_size, latent_dim
        self.g_spec = generator_spec(nz=latent_dim, img_size=img_size)
        self.d_spec = discriminator_spec(img_size=img_size)
        self.E = make_encoder_state([in_ch, img_size, img_size], latent_dim)     # vaegan_code.py:29
        self.G = make_sequential_state(self.g_spec)                             # :30
        self.D = make_sequential_state(self.d_spec)                             # :31
        weights_init_state(self.G, self.g_spec)                                 # :37
        weights_init_state(self.D, self.d_spec)                                 # :38
        self.opt_E = RefAdam(require_grads(self.E), lr=lr)                      # :42
        self.opt_G = RefAdam(require_grads(self.G), lr=lr)                      # :43
        self.opt_D = RefAdam(require_grads(self.D), lr=lr)                      # :44

    def double_(self) -> "RefVAEGAN":
        """Switch the whole model + optimiser state to float64 (call before the first step).  Used by the
        parity tests to measure how ill-conditioned a quantity is: |cpu-fp32 - fp64| calibrates the tolerance
        a faithful fp32 implementation can be held to (the trajectory is chaotic)."""
        assert self.opt_E.t == 0
        for st in (self.E, self.G, self.D):
            for k in list(st.keys()):
                if st[k].is_floating_point():
                    st[k] = st[k].detach().double()
        lr = self.opt_E.lr
        self.opt_E = RefAdam(require_grads(self.E), lr=lr)
        self.opt_G = RefAdam(require_grads(self.G), lr=lr)
        self.opt_D = RefAdam(require_grads(self.D), lr=lr)
        self.dtype = torch.float64
        return self

    def train_step(self, real, eps_z, eps_real, eps_recon, epoch: int,
                   alpha_kl: float = 0.1, alpha_adv: float = 0.1) -> Dict[str, float]:
        """One iteration of vaegan_code.py:65-135; the three randn_like draws are injected."""
        B = real.size(0)
        dt = getattr(self, "dtype", torch.float32)
        real, eps_z, eps_real, eps_recon = (t.to(dt) for t in (real, eps_z, eps_real, eps_recon))
        mu, logvar = encoder_forward(self.E, real, True)                        # :74
        logvar = torch.clamp(logvar, min=-10, max=10)                           # :75
        std = torch.exp(0.5 * logvar)                                           # :76
        z = (mu + std * eps_z).unsqueeze(-1).unsqueeze(-1)                      # :77-78
        recon = generator_forward(self.G, self.g_spec, z, True)                # :83
        real_labels = torch.full((B,), 0.9, dtype=dt)                           # :88
        fake_labels = torch.full((B,), 0.1, dtype=dt)                           # :89
        real_noisy = real + 0.05 * eps_real                                     # :91
        recon_noisy = recon + 0.05 * eps_recon                                  # :92
        d_losses = []
        for _ in range(2):                                                      # :95
            real_out = discriminator_forward(self.D, self.d_spec, real_noisy, True)
            fake_out = discriminator_forward(self.D, self.d_spec, recon_noisy.detach(), True)
            d_loss = bce_loss(real_out, real_labels) + bce_loss(fake_out, fake_labels)   # :99-101
            self.opt_D.zero_grad()
            d_loss.backward()
            self.opt_D.step()
            d_losses.append(float(d_loss.detach()))
        fake_out = discriminator_forward(self.D, self.d_spec, recon_noisy, True)  # :110
        recon_loss = mse_loss(recon, real)                                      # :113
        kl_loss = kl_sum(mu, logvar) / B                                        # :114
        g_loss_adv = bce_loss(fake_out, real_labels)                            # :115
        total = recon_loss + alpha_kl * min(1.0, epoch / 50) * kl_loss + alpha_adv * g_loss_adv   # :117
        self.opt_E.zero_grad()
        self.opt_G.zero_grad()
        total.backward()
        self.opt_E.step()
        self.opt_G.step()
        return {"recon_loss": float(recon_loss.detach()), "kl_loss": float(kl_loss.detach()), "g_loss_adv": float(g_loss_adv.detach()),
                "d_loss_1": d_losses[0], "d_loss_2": d_losses[1], "total": float(total.detach())}

    @torch.no_grad()
    def denoise(self, img, noise, eps_z):
        """Validation forward, vaegan_code.py:147-171 (eval-mode E and G); noise already scaled by sigma."""
        noisy = torch.clamp(img + noise, -1.0, 1.0)
        mu, logvar = encoder_forward(self.E, noisy, False)
        logvar = torch.clamp(logvar, min=-10, max=10)
        z = (mu + torch.exp(0.5 * logvar) * eps_z).unsqueeze(-1).unsqueeze(-1)
        recon = generator_forward(self.G, self.g_spec, z, False)
        recon_loss = mse_loss(recon, img)
        kl = kl_sum(mu, logvar)                                                 # not divided by B (:166)
        return noisy, recon, float(recon_loss), float(kl)


def validation_epoch(model: "RefVAEGAN", batches, noises, alpha_kl: float = 0.1, n_samples=None):
    """The validation loop of vaegan_code.py:147-191 around RefVAEGAN.denoise (eval-mode E and G):
    val_loss = sum over batches of (mse_mean + alpha_kl * KL_sum) / number of SAMPLES (:167, :187);
    SSIM accumulated over all images as torchmetrics' running sums do (:174, :185; restated recipe, parity unpinned);
    PSNR of the mean squared error over the whole pass (not in the reference).  noises[i] = (sigma*eps, eps_z)."""
    val, ssim_sum, se_sum, seen = 0.0, 0.0, 0.0, 0
    for img, (noise, eps_z) in zip(batches, noises):
        _, recon, rl, kl = model.denoise(img, noise, eps_z)
        val += rl + alpha_kl * kl
        a01, b01 = (recon + 1) / 2, (img + 1) / 2
        ssim_sum += ssim(a01, b01) * img.size(0)
        se_sum += float(((a01.double() - b01.double()) ** 2).mean()) * img.size(0)
        seen += img.size(0)
    n = seen if n_samples is None else n_samples
    import math
    return {"val_loss": val / n, "ssim": ssim_sum / seen, "psnr": 10.0 * math.log10(1.0 / (se_sum / seen)), "samples": seen}


def configure_seed(seed: int) -> None:
    """utils.py:6-14 (host-side part)."""
    import os
    import random
    import numpy as np
    os.environ["PYTHONHASHSEED"] = str(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


# --------------------------------------------------------------------------------------
# Image metrics for the denoise path.  PSNR is not implemented by the reference (README.md:22
# lists it as intended); SSIM in the reference comes from torchmetrics (absent here) --
# "parity unpinned": restated from the published definition used by the call at
# vaegan_code.py:143 (gaussian 11x11, sigma 1.5, k1 .01, k2 .03, data_range 1).
# --------------------------------------------------------------------------------------

def psnr(a: torch.Tensor, b: torch.Tensor, data_range: float = 1.0) -> float:
    mse = torch.mean((a.double() - b.double()) ** 2)
    return float(10.0 * torch.log10(data_range ** 2 / mse))


def ssim(a: torch.Tensor, b: torch.Tensor, data_range: float = 1.0) -> float:
    k = torch.arange(11, dtype=torch.float64) - 5
    g = torch.exp(-(k ** 2) / (2 * 1.5 ** 2))
    g = g / g.sum()
    w = (g[:, None] * g[None, :]).to(torch.float64)
    C = a.size(1)
    w = w.expand(C, 1, 11, 11).contiguous()
    a = a.double()
    b = b.double()
    pad = 5
    ap = F.pad(a, (pad, pad, pad, pad), mode="reflect")
    bp = F.pad(b, (pad, pad, pad, pad), mode="reflect")
    mu_a = F.conv2d(ap, w, groups=C)
    mu_b = F.conv2d(bp, w, groups=C)
    s_aa = F.conv2d(ap * ap, w, groups=C) - mu_a ** 2
    s_bb = F.conv2d(bp * bp, w, groups=C) - mu_b ** 2
    s_ab = F.conv2d(ap * bp, w, groups=C) - mu_a * mu_b
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    m = ((2 * mu_a * mu_b + c1) * (2 * s_ab + c2)) / ((mu_a ** 2 + mu_b ** 2 + c1) * (s_aa + s_bb + c2))
    # torchmetrics crops the reflect-padded border back off before averaging
    m = m[..., pad:-pad, pad:-pad]
    return float(m.reshape(m.size(0), -1).mean(-1).mean())

"""CPU oracle for the reference's three SIBLING training loops, which reuse the hot path's networks and kernels
(SURVEY.md section 8(f), row F4):

  * plain VAE            main_vae.py:60-135   (``train_vae``; Decoder == gan_code.Generator, main_vae.py:10)
  * DCGAN                gan_code.py:162-222  (``train_gan``)
  * weight-clipped WGAN  gan_code.py:261-340  (``train_wgan``)

TEST INFRASTRUCTURE, same rules as ``vaegan_ref.py`` (only tests / smoke / the bench's cpu_baseline leg may import
it).  Every random draw of the loops (``randn_like`` / ``randn``) is injected so that the HIP path, this oracle
and the reference's own modules can be run on identical inputs.  Pinned by ``oracle/gen_golden_siblings.py``:
the same loops run around the reference's imported classes with stock ``torch.optim.Adam`` must agree with this
file bit for bit at S=256 (vectors in ``tests/golden/sibling_*.npz``).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

import vaegan_ref as R


class RefVAE:
    """main_vae.py:81-88: Encoder + Decoder(latent_dim) with PyTorch default init (train_vae never calls
    weights_init), ONE Adam over both parameter lists, lr 1e-3."""

    def __init__(self, img_size: int = 256, latent_dim: int = 100, in_ch: int = 3, lr: float = 1e-3,
                 seed: Optional[int] = 42):
        if seed is not None:
            R.configure_seed(seed)                                                  # main_vae.py:61
        self.g_spec = R.generator_spec(nz=latent_dim, img_size=img_size)
        self.E = R.make_encoder_state([in_ch, img_size, img_size], latent_dim)      # :82
        self.G = R.make_sequential_state(self.g_spec)                               # :83
        self.opt = R.RefAdam(R.require_grads(self.E) + R.require_grads(self.G), lr=lr)   # :84-87

    def train_step(self, img, eps_img, eps_z, epoch: int, noise_max_std: float = 0.5) -> Dict[str, float]:
        """One iteration of main_vae.py:103-127.  eps_img / eps_z stand for the two randn_like draws."""
        noisy = torch.clamp(img + eps_img * noise_max_std, -1.0, 1.0)               # :104-105
        mu, logvar = R.encoder_forward(self.E, noisy, True)                         # :111
        logvar = torch.clamp(logvar, min=-10, max=10)                               # :112
        std = torch.exp(0.5 * logvar)                                               # :113
        z = (mu + std * eps_z).unsqueeze(-1).unsqueeze(-1)                          # :114-115
        recon = R.generator_forward(self.G, self.g_spec, z, True)                   # :116
        recon_loss = R.mse_loss(recon, img)                                         # :119
        kl_loss = R.kl_sum(mu, logvar)                                              # :120 (NOT divided by B)
        total = recon_loss + kl_loss * min(epoch / 50, 1.0) * 1e-5                  # :121
        self.opt.zero_grad()                                                        # :124
        total.backward()
        self.opt.step()
        return {"recon_loss": float(recon_loss.detach()), "kl_loss": float(kl_loss.detach()),
                "total": float(total.detach())}


class _RefGAN:
    """gan_code.py:170-181 / :269-278: Generator(nz=100) and Discriminator(), both through weights_init, two Adams
    with lr 2e-4 and betas (0.5, 0.999)."""

    def __init__(self, img_size: int = 256, nz: int = 100, lr: float = 2e-4, seed: Optional[int] = 42):
        if seed is not None:
            R.configure_seed(seed)
        self.g_spec = R.generator_spec(nz=nz, img_size=img_size)
        self.d_spec = R.discriminator_spec(img_size=img_size)
        self.G = R.make_sequential_state(self.g_spec)
        self.D = R.make_sequential_state(self.d_spec)
        R.weights_init_state(self.G, self.g_spec)
        R.weights_init_state(self.D, self.d_spec)
        self.opt_D = R.RefAdam(R.require_grads(self.D), lr=lr, betas=(0.5, 0.999))
        self.opt_G = R.RefAdam(R.require_grads(self.G), lr=lr, betas=(0.5, 0.999))

    def _g(self, z):
        return R.generator_forward(self.G, self.g_spec, z, True)

    def _d(self, x):
        return R.discriminator_forward(self.D, self.d_spec, x, True)


class RefDCGAN(_RefGAN):
    def train_step(self, real, noise) -> Dict[str, float]:
        """gan_code.py:194-219.  noise [B, nz, 1, 1] stands for the torch.randn of :203.  netD.zero_grad() /
        netG.zero_grad() are module-level: the generator step also leaves gradients in D that the next
        iteration's netD.zero_grad() clears unread."""
        B = real.size(0)
        ones, zeros = torch.ones(B, dtype=real.dtype), torch.zeros(B, dtype=real.dtype)
        self.opt_D.zero_grad()                                                      # :195 netD.zero_grad()
        errD_real = R.bce_loss(self._d(real), ones)                                 # :199-200
        errD_real.backward()                                                        # :201
        fake = self._g(noise)                                                       # :204
        errD_fake = R.bce_loss(self._d(fake.detach()), zeros)                       # :206-207
        errD_fake.backward()                                                        # :208 (accumulates)
        self.opt_D.step()                                                           # :209
        self.opt_G.zero_grad()                                                      # :212 netG.zero_grad()
        errG = R.bce_loss(self._d(fake), ones)                                      # :214-215
        errG.backward()                                                             # :216
        self.opt_G.step()                                                           # :217
        return {"errD_real": float(errD_real.detach()), "errD_fake": float(errD_fake.detach()),
                "errG": float(errG.detach())}


class RefWGAN(_RefGAN):
    def train_step(self, real, critic_noise: List[torch.Tensor], gen_noise, clip_value: float = 0.01) -> Dict[str, float]:
        """gan_code.py:296-328: len(critic_noise) critic iterations (5 in the reference), each followed by the
        weight clamp to +-clip_value, then one generator step.  The critic is the SAME Discriminator, sigmoid
        included (gan_code.py:270)."""
        d_loss = None
        for z in critic_noise:                                                      # :301
            self.opt_D.zero_grad()                                                  # :302
            d_loss_real = -self._d(real).mean()                                     # :305-306
            with torch.no_grad():
                fake = self._g(z)                                                   # :310 (.detach())
            d_loss_fake = self._d(fake).mean()                                      # :311-312
            d_loss = d_loss_real + d_loss_fake                                      # :315
            d_loss.backward()
            self.opt_D.step()                                                       # :317
            with torch.no_grad():
                for p in self.opt_D.params:                                         # :320-321
                    p.clamp_(-clip_value, clip_value)
        self.opt_G.zero_grad()                                                      # :324
        g_loss = -self._d(self._g(gen_noise)).mean()                                # :325-328
        g_loss.backward()                                                           # :330
        self.opt_G.step()
        return {"d_loss": float(d_loss.detach()), "g_loss": float(g_loss.detach())}

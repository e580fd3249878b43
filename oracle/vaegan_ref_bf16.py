"""CPU oracle, bf16-STORAGE mode: the training iteration of vaegan_code.py:65-135 in exact (float64) arithmetic with a
round-to-nearest-even bf16 rounding at exactly the tensors the HIP engine STORES in bf16 -- and nowhere else.

TEST INFRASTRUCTURE (same rule as vaegan_ref.py: only tests/ and tools/ may import it).  Purpose (round-3 review, weak #1):
the bf16 engine's gradients sit 0.4 % ... 20 % from the fp64 oracle; that distance is the arithmetic the engine was ASKED to
do (bf16 storage, f32 accumulate), not necessarily kernel error.  This file restates the asked-for arithmetic, so that the
HIP kernels can be held against it per tensor at a kernel-level bound (tests/test_gpu_configs.py), and it carries the
per-tensor ablation switches (`keep_fp`) that show WHICH stored tensor costs the accuracy (tools/bf16_ablation.py).

What the engine stores in bf16 (engine.py / ops.py / csrc, dtype VG_BF16), i.e. where `q()` / `gq()` sit below:
  forward   * the image batches in NHWC: q(real) (Encoder input), q(real + s*eps) and q(recon + s*eps) (Discriminator inputs)
            * every packed GEMM operand: q(W) of every conv / convT / linear weight and of the Discriminator head
              (biases, BatchNorm gamma / beta and all master weights stay fp32)
            * every raw conv output Y -- the BatchNorm batch statistics are taken from the f32 ACCUMULATORS, before that
              rounding (conv_gemm.hip epilogue), except for the Generator's first layer (taps folded into N: its statistics
              come from a pass over the stored, i.e. rounded, tensor: engine.py `channel_stats`)
            * every activated output q(act(scale*q(Y)+shift)); a layer without BatchNorm stores only q(act(Y))
            * the Encoder head's [mu | logvar] (mulv) and the latent z; the clamped logvar, p = sigmoid(.), the f32 NCHW
              reconstruction tanh(.) and all losses stay f32
  backward  * every data gradient dX written by a dgrad GEMM (the activation mask of a BatchNorm-less layer below is applied
              to the ROUNDED tile and the product rounded again: q(slope * q(dX))), the Discriminator head's dX
            * every BatchNorm-backward output dY = a*dz - b*xhat - c (xhat and act' are re-derived from the STORED q(Y))
            * the gradient of the pre-tanh reconstruction q((d_mse + d_adv) * (1 - recon^2)), and d[mu | logvar]
            * weight / bias / gamma / beta gradients and the (sum dz, sum dz*xhat) reductions are f32: not rounded
The float64 arithmetic between the roundings stands for the engine's f32 accumulation (whose own error shows as the
distance between vaegan_ref.py in fp32 and in fp64: ~1e-3 on the worst tensors).

Each function cites the reference lines it restates, like vaegan_ref.py, whose state layout and constructors it reuses.
"""
from __future__ import annotations

from typing import Dict, Optional, Set

import torch
import torch.nn.functional as F

import vaegan_ref as R


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    """Round-to-nearest-even to bfloat16, returned in x's dtype (f64 -> f32 -> bf16 -> f64)."""
    return x.detach().to(torch.float32).to(torch.bfloat16).to(x.dtype)


class _Q(torch.autograd.Function):
    """Stored in bf16 on the way forward; gradient passes through."""

    @staticmethod
    def forward(ctx, x):
        return bf16_round(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _GQ(torch.autograd.Function):
    """Identity forward; the GRADIENT arriving here is a tensor the engine stores in bf16."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return bf16_round(g)


class _BNAct(torch.autograd.Function):
    """Train-mode BatchNorm2d + (Leaky)ReLU as the engine executes it (bn_act.hip): statistics from the UNROUNDED conv
    output (or from the stored one: `stats_rounded`), normalisation of the STORED q(Y), output stored in bf16; backward
    re-derives xhat and the activation derivative from q(Y) and stores dY in bf16.  Returns (out, mean, biased var)."""

    @staticmethod
    def forward(ctx, Y, gamma, beta, slope, stats_rounded, round_y, round_out, round_dy):
        Yq = bf16_round(Y) if round_y else Y.detach()
        src = Yq if stats_rounded else Y
        mean = src.mean(dim=(0, 2, 3))
        var = src.var(dim=(0, 2, 3), unbiased=False)
        invstd = torch.rsqrt(var + R.BN_EPS)
        scale = (gamma * invstd).view(1, -1, 1, 1)
        shift = (beta - mean * gamma * invstd).view(1, -1, 1, 1)
        z = scale * Yq + shift
        out = torch.where(z > 0, z, z * slope)
        ctx.save_for_backward(Yq, mean, invstd, gamma, z)
        ctx.slope, ctx.round_dy = slope, round_dy
        return (bf16_round(out) if round_out else out), mean.detach(), var.detach()

    @staticmethod
    def backward(ctx, dA, _dm, _dv):
        Yq, mean, invstd, gamma, z = ctx.saved_tensors
        dz = torch.where(z > 0, dA, dA * ctx.slope)
        xhat = (Yq - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
        n = Yq.numel() // Yq.shape[1]
        s1 = dz.sum(dim=(0, 2, 3))
        s2 = (dz * xhat).sum(dim=(0, 2, 3))
        a = (gamma * invstd).view(1, -1, 1, 1)
        dY = a * (dz - (s1 / n).view(1, -1, 1, 1) - xhat * (s2 / n).view(1, -1, 1, 1))
        return (bf16_round(dY) if ctx.round_dy else dY), s2, s1, None, None, None, None, None


class RefVAEGANbf16(R.RefVAEGAN):
    """RefVAEGAN (vaegan_code.py:29-44 construction, same seed-42 state) whose train_step rounds to bf16 where the engine
    stores bf16.  keep_fp: names of storage points to leave UNROUNDED (ablation):
        'w'                      all packed weights              'img'    the three image batches
        'Y:<net>.<i>'            raw conv output of stage i      'A:<net>.<i>'  its activated output
        'dX:<net>.<i>'           data gradient written by stage i's dgrad (the gradient of stage i's INPUT)
        'dY:<net>.<i>'           BatchNorm-backward output of stage i
        'mulv', 'z', 'dmulv', 'dpre'
    or a whole class: 'Y', 'A', 'dX', 'dY', 'fwd' (everything forward), 'bwd' (everything backward).
    net in {E, G, D}; stage indices count conv-like layers from 0 (E: 4 = the fc pair; D: last = head)."""

    def __init__(self, img_size: int = 256, latent_dim: int = 100, in_ch: int = 3, lr: float = 2e-4,
                 seed: Optional[int] = 42, keep_fp: Optional[Set[str]] = None):
        super().__init__(img_size, latent_dim, in_ch, lr, seed)
        self.keep_fp = set(keep_fp or ())
        self.double_()

    # ---- storage points ---------------------------------------------------------------------------------------
    def _keep(self, tag: str) -> bool:
        k = self.keep_fp
        cls = tag.split(":")[0]
        if tag in k or cls in k:
            return True
        fwd = cls in ("w", "img", "Y", "A", "mulv", "z")
        return ("fwd" in k and fwd) or ("bwd" in k and not fwd)

    def q(self, x, tag):
        return x if self._keep(tag) else _Q.apply(x)

    def gq(self, x, tag):
        return x if self._keep(tag) else _GQ.apply(x)

    # ---- networks -----------------------------------------------------------------------------------------------
    def _bn_act(self, st, prefix, Y, slope, net, i, stats_rounded=False):
        out, mean, var = _BNAct.apply(Y, st[f"{prefix}.weight"], st[f"{prefix}.bias"], slope, stats_rounded,
                                      not self._keep(f"Y:{net}.{i}"), not self._keep(f"A:{net}.{i}"),
                                      not self._keep(f"dY:{net}.{i}"))
        with torch.no_grad():                                      # running statistics, nn.BatchNorm2d defaults (:56-58 train mode)
            n = Y.numel() // Y.shape[1]
            st[f"{prefix}.running_mean"].mul_(1 - R.BN_MOMENTUM).add_(R.BN_MOMENTUM * mean)
            st[f"{prefix}.running_var"].mul_(1 - R.BN_MOMENTUM).add_(R.BN_MOMENTUM * var * n / max(n - 1, 1))
            st[f"{prefix}.num_batches_tracked"] += 1
        return out

    def encoder(self, x_q):
        """main_vae.py:50-58 (ConvBlock :27-31): x_q is the stored image batch."""
        st, a = self.E, x_q
        for i in range(4):
            if i > 0:
                a = self.gq(a, f"dX:E.{i}")                        # stage i's dgrad writes the gradient of its input in bf16
            Y = F.conv2d(a, self.q(st[f"cnn.{i}.conv.weight"], "w"), st[f"cnn.{i}.conv.bias"], stride=2)
            a = self._bn_act(st, f"cnn.{i}.bn", Y, 0.01, "E", i)
        h = self.gq(a, "dX:E.4").reshape(a.size(0), -1)
        W = torch.cat([self.q(st["fc_mu.weight"], "w"), self.q(st["fc_logvar.weight"], "w")], 0)
        b = torch.cat([st["fc_mu.bias"], st["fc_logvar.bias"]], 0)
        mulv = self.q(self.gq(F.linear(h, W, b), "dmulv"), "mulv")   # [mu | logvar] stored bf16; its gradient too
        L = self.latent_dim
        return mulv[:, :L], mulv[:, L:]

    def generator(self, z_q):
        """gan_code.py:53-54 over generator_spec; returns (recon f32-equivalent NCHW, pre-tanh handle)."""
        st, spec, a, li = self.G, self.g_spec, z_q, 0
        convs = [i for i, e in enumerate(spec) if e[0] == "convT"]
        for n, i in enumerate(convs):
            _, cin, cout, k, s, p = spec[i]
            a = self.gq(a, f"dX:G.{n}")
            Y = F.conv_transpose2d(a, self.q(st[f"main.{i}.weight"], "w"), None, stride=s, padding=p)
            if n + 1 < len(convs):
                # the 1x1-input first layer folds its taps into N: statistics from the stored (rounded) tensor
                a = self._bn_act(st, f"main.{i + 1}", Y, 0.0, "G", n, stats_rounded=(n == 0))
            else:
                return torch.tanh(self.gq(Y, "dpre"))              # :50 Tanh; f32 NCHW image, its input gradient stored bf16

    def discriminator(self, x_q, need_dx: bool):
        """gan_code.py:88-89 over discriminator_spec; x_q is a stored NHWC-bf16 image batch."""
        st, spec = self.D, self.d_spec
        convs = [i for i, e in enumerate(spec) if e[0] == "conv"]
        a = self.gq(x_q, "dX:D.0") if need_dx else x_q
        for n, i in enumerate(convs):
            _, cin, cout, k, s, p = spec[i]
            w = self.q(st[f"main.{i}.weight"], "w")
            if n == len(convs) - 1:                                # head: Conv2d(C, 1, 4, 1, 0) + Sigmoid, f32 logit / p
                return torch.sigmoid(F.conv2d(self.gq(a, f"dX:D.{n}"), w, None, stride=s, padding=p)).view(-1)
            if n == 0:
                # no BatchNorm: LeakyReLU in the conv epilogue, only q(act(Y)) is stored; its backward is a mask applied by
                # stage 1's dgrad epilogue to its tile AFTER the tile's bf16 rounding (conv_gemm.hip mask_segment): the
                # slope-multiplied elements are rounded twice, q(slope * q(dX))
                Y = F.conv2d(a, w, None, stride=s, padding=p)
                a = self.q(self.gq(F.leaky_relu(self.gq(Y, "dX:D.1"), spec[i + 1][1]), "dX:D.1"), f"A:D.{n}")
            else:
                if n > 1:
                    a = self.gq(a, f"dX:D.{n}")
                Y = F.conv2d(a, w, None, stride=s, padding=p)
                a = self._bn_act(st, f"main.{i + 1}", Y, spec[i + 2][1], "D", n)

    # ---- the iteration ------------------------------------------------------------------------------------------
    def train_step(self, real, eps_z, eps_real, eps_recon, epoch: int, alpha_kl: float = 0.1,
                   alpha_adv: float = 0.1, sigma: float = 0.05) -> Dict[str, float]:
        """vaegan_code.py:65-135 with the three randn_like draws injected (as RefVAEGAN.train_step)."""
        B, dt = real.size(0), torch.float64
        real, eps_z, eps_real, eps_recon = (t.to(dt) for t in (real, eps_z, eps_real, eps_recon))
        mu, logvar = self.encoder(self.q(real, "img"))                                   # :74
        lv = torch.clamp(logvar, min=-10, max=10)                                        # :75 (kept f32 by the engine)
        z = self.q(mu + torch.exp(0.5 * lv) * eps_z, "z").unsqueeze(-1).unsqueeze(-1)    # :76-78, z stored bf16
        recon = self.generator(z)                                                        # :83
        real_labels = torch.full((B,), 0.9, dtype=dt)                                    # :88
        fake_labels = torch.full((B,), 0.1, dtype=dt)                                    # :89
        real_noisy = self.q(real + sigma * eps_real, "img")                              # :91, stored NHWC bf16
        recon_noisy = self.q(recon + sigma * eps_recon, "img")                           # :92
        d_losses = []
        for _ in range(2):                                                               # :95
            real_out = self.discriminator(real_noisy, False)
            fake_out = self.discriminator(recon_noisy.detach(), False)
            d_loss = R.bce_loss(real_out, real_labels) + R.bce_loss(fake_out, fake_labels)   # :99-101
            self.opt_D.zero_grad()
            d_loss.backward()
            if not d_losses:                                                             # gradients of the first update (:103-104)
                grads_D = {k: self.D[k].grad.clone() for k in R.trainable_keys(self.D)}
            self.opt_D.step()
            d_losses.append(float(d_loss.detach()))
        fake_out = self.discriminator(recon_noisy, True)                                 # :110
        recon_loss = R.mse_loss(recon, real)                                             # :113
        kl_loss = R.kl_sum(mu, lv) / B                                                   # :114
        g_loss_adv = R.bce_loss(fake_out, real_labels)                                   # :115
        total = recon_loss + alpha_kl * min(1.0, epoch / 50) * kl_loss + alpha_adv * g_loss_adv   # :117
        self.opt_E.zero_grad()
        self.opt_G.zero_grad()
        total.backward()
        self.last_grads = {"E": {k: self.E[k].grad.clone() for k in R.trainable_keys(self.E)},
                           "G": {k: self.G[k].grad.clone() for k in R.trainable_keys(self.G)}, "D": grads_D}
        self.opt_E.step()
        self.opt_G.step()
        return {"recon_loss": float(recon_loss.detach()), "kl_loss": float(kl_loss.detach()),
                "g_loss_adv": float(g_loss_adv.detach()), "d_loss_1": d_losses[0], "d_loss_2": d_losses[1],
                "total": float(total.detach())}

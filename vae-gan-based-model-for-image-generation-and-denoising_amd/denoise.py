"""Denoising evaluation path of the reference (vaegan_code.py:147-171; BASELINE config 4): eval-mode Encoder ->
reparameterisation -> Generator on ``clamp(img + sigma*eps, -1, 1)``, reconstruction MSE + KL (not divided by
the batch, :166), and image-quality metrics on the [0,1]-rescaled images: SSIM (torchmetrics recipe, parity
unpinned: the package is not installed) and PSNR (listed as intended in the reference README.md:22, implemented
nowhere in it).  All compute is HIP kernels; Inception-based IS/FID need downloaded weights and are out of scope.
"""
import math
from typing import Dict, Iterable, Optional

import torch

from . import geometry as G
from . import ops


@torch.no_grad()
def denoise_eval(encoder, decoder, img: torch.Tensor, sigma: float = 0.05, eps: Optional[torch.Tensor] = None,
                 eps_z: Optional[torch.Tensor] = None, alpha_kl: float = 0.1) -> Dict[str, object]:
    """img: [B,C,S,S] float32 on the MI355X in [-1,1].  Returns the noisy input, the reconstruction (both NCHW
    f32) and a dict of device scalars + host floats: recon_loss, kl_loss (sum), val_loss, psnr, ssim.
    The caller puts encoder / decoder in eval mode (vaegan_code.py:147-148) -- or not: BatchNorm follows
    module.training exactly as in the reference."""
    if not img.is_cuda:
        raise RuntimeError("denoise_eval needs the batch on the MI355X ('cuda'); there is no CPU path")
    dt = encoder._dt
    B, C = img.shape[0], img.shape[1]
    L = encoder.latent_dim
    img = img.contiguous()
    if eps is None or eps_z is None:
        ns = ops.default_noise(img.device)                   # HIP Philox draws keyed by torch's device seed
        ns.advance()
        eps = ns.randn(tuple(img.shape), 0) if eps is None else eps
        eps_z = ns.randn((B, L), 1) if eps_z is None else eps_z
    noisy_h, noisy = ops.noisy_clamp_to_nhwc(img, eps, sigma, G.padc(C, dt), dt)
    mulv, _ = encoder._engine.forward(noisy_h, B, encoder.training, keep=False)
    mulv = mulv.view(B, -1)
    z, lvc = ops.reparam_forward(mulv, eps_z, L, G.padc(decoder.nz, dt), dt)
    pre, _ = decoder._engine.forward(z, B, decoder.training, keep=False)
    recon = ops.nhwc_to_nchw(pre, decoder.nc, dt, apply_tanh=True)
    scal = ops.zeros_f32(4, img.device)
    ops.mse_forward_backward(recon, img, 1.0, scal[0:1], False)            # recon_loss (:165)
    ops.kl_forward(mulv, lvc, L, 1.0, dt, out=scal[1:2])                   # KL sum, no /B (:166)
    ssim_t = ops.ssim(recon, img)
    recon_loss, kl = (float(v) for v in scal[:2].tolist())
    mse01 = recon_loss / 4.0                                                # ((a+1)/2 - (b+1)/2)^2 = (a-b)^2 / 4
    psnr = float("inf") if mse01 == 0 else 10.0 * math.log10(1.0 / mse01)
    return {"noisy": noisy, "recon": recon, "recon_loss": recon_loss, "kl_loss": kl,
            "val_loss": recon_loss + alpha_kl * kl, "psnr": psnr, "ssim": float(ssim_t.item())}


@torch.no_grad()
def validation_epoch(encoder, decoder, loader: Iterable[torch.Tensor], n_samples: Optional[int] = None,
                     sigma: float = 0.05, alpha_kl: float = 0.1, noise_fn=None) -> Dict[str, float]:
    """The per-epoch validation loop of the reference trainer, vaegan_code.py:147-191:

        encoder.eval(); decoder.eval()                                   (:147-148; the discriminator is not used)
        for img in val_loader:   noisy = clamp(img + 0.05*randn, -1, 1) -> E -> reparameterise -> G
            val_loss += mse_mean(recon, img) + alpha_kl * KL_sum          (:165-167: ONE number per batch)
            ssim.update((recon+1)/2, (img+1)/2)                           (:170-174)
        val_loss /= len(val_loader.dataset)                               (:187: divided by the SAMPLE count)
        ssim.compute()                                                    (:185: mean over all validation images)

    loader yields device batches [b,C,S,S] in [-1,1] (data.DeviceLoader; the ragged last batch counts with its own
    size, as torchmetrics' running sums do).  n_samples defaults to the number of images seen (= len(dataset) for a
    full pass).  noise_fn(i, img) -> (eps, eps_z) injects the two draws of batch i (parity tests); by default they are
    generated on the device.  PSNR (not in the reference, README.md:22 lists it as intended) is that of the mean
    squared error over every pixel of the pass.  Inception Score / FID need downloaded InceptionV3 weights: not
    available offline, left out.  Accumulation stays on the device; ONE host sync at the end of the pass.
    Returns python floats: val_loss, ssim, psnr, recon_loss (mean of the batch MSEs), kl_loss (mean of the batch KL
    sums), samples, batches."""
    encoder.eval(), decoder.eval()                                                   # :147-148
    dev = next(encoder.parameters()).device
    acc = ops.zeros_f32(4, dev)                # [sum(recon + a*kl), sum(b * ssim_b), sum(b * mse_b), sum(kl)]
    seen = batches = 0
    dt, L = encoder._dt, encoder.latent_dim
    for i, img in enumerate(loader):
        if not img.is_cuda:
            raise RuntimeError("validation_epoch needs device batches (data.DeviceLoader); there is no CPU path")
        img = img.contiguous()
        b, C = img.shape[0], img.shape[1]
        if noise_fn is not None:
            eps, eps_z = noise_fn(i, img)
        else:
            ns = ops.default_noise(dev)
            ns.advance()
            eps, eps_z = ns.randn(tuple(img.shape), 0), ns.randn((b, L), 1)
        noisy_h, _ = ops.noisy_clamp_to_nhwc(img, eps, sigma, G.padc(C, dt), dt)     # :153-154
        mulv, _ = encoder._engine.forward(noisy_h, b, False, keep=False)
        mulv = mulv.view(b, -1)
        z, lvc = ops.reparam_forward(mulv, eps_z, L, G.padc(decoder.nz, dt), dt)     # :160-162
        pre, _ = decoder._engine.forward(z, b, False, keep=False)
        recon = ops.nhwc_to_nchw(pre, decoder.nc, dt, apply_tanh=True)               # :163
        scal = torch.empty(2, dtype=torch.float32, device=dev)
        ops.mse_forward_backward(recon, img, 1.0, scal[0:1], False)                  # :165
        ops.kl_forward(mulv, lvc, L, 1.0, dt, out=scal[1:2])                         # :166 (sum, not / B)
        ssim_b = ops.ssim(recon, img)                                                # :170-174 (mean over the batch)
        ops.axpy(acc[0:1], scal[0:1], 1.0, out=acc[0:1])
        ops.axpy(acc[0:1], scal[1:2], alpha_kl, out=acc[0:1])                        # :167
        ops.axpy(acc[1:2], ssim_b, float(b), out=acc[1:2])
        ops.axpy(acc[2:3], scal[0:1], float(b), out=acc[2:3])
        ops.axpy(acc[3:4], scal[1:2], 1.0, out=acc[3:4])
        seen += b
        batches += 1
    if batches == 0:
        raise RuntimeError("validation_epoch: the loader yielded no batch")
    val_sum, ssim_sum, mse_sum, kl_sum = (float(v) for v in acc.tolist())           # the one host sync
    n = seen if n_samples is None else int(n_samples)
    mse01 = mse_sum / seen / 4.0
    return {"val_loss": val_sum / n, "ssim": ssim_sum / seen,
            "psnr": float("inf") if mse01 == 0 else 10.0 * math.log10(1.0 / mse01),
            "recon_loss": mse_sum / seen, "kl_loss": kl_sum / batches, "samples": seen, "batches": batches}

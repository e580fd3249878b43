"""Denoising evaluation path of the reference (vaegan_code.py:147-171; BASELINE config 4): eval-mode Encoder ->
reparameterisation -> Generator on ``clamp(img + sigma*eps, -1, 1)``, reconstruction MSE + KL (not divided by
the batch, :166), and image-quality metrics on the [0,1]-rescaled images: SSIM (torchmetrics recipe, parity
unpinned: the package is not installed) and PSNR (listed as intended in the reference README.md:22, implemented
nowhere in it).  All compute is HIP kernels; Inception-based IS/FID need downloaded weights and are out of scope.
"""
import math
from typing import Dict, Optional

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
    if eps is None:
        eps = torch.randn_like(img)
    if eps_z is None:
        eps_z = torch.randn(B, L, device=img.device)
    noisy_h, noisy = ops.noisy_clamp_to_nhwc(img, eps, sigma, G.padc(C, dt), dt)
    mulv, _ = encoder._engine.forward(noisy_h, B, encoder.training, keep=False)
    mulv = mulv.view(B, -1)
    z, lvc = ops.reparam_forward(mulv, eps_z, L, G.padc(decoder.nz, dt), dt)
    pre, _ = decoder._engine.forward(z, B, decoder.training, keep=False)
    recon = ops.nhwc_to_nchw(pre, decoder.nc, dt, apply_tanh=True)
    scal = torch.zeros(4, dtype=torch.float32, device=img.device)
    ops.mse_forward_backward(recon, img, 1.0, scal[0:1], False)            # recon_loss (:165)
    ops.kl_forward(mulv, lvc, L, 1.0, dt, out=scal[1:2])                   # KL sum, no /B (:166)
    ssim_t = ops.ssim(recon, img)
    recon_loss, kl = (float(v) for v in scal[:2].tolist())
    mse01 = recon_loss / 4.0                                                # ((a+1)/2 - (b+1)/2)^2 = (a-b)^2 / 4
    psnr = float("inf") if mse01 == 0 else 10.0 * math.log10(1.0 / mse01)
    return {"noisy": noisy, "recon": recon, "recon_loss": recon_loss, "kl_loss": kl,
            "val_loss": recon_loss + alpha_kl * kl, "psnr": psnr, "ssim": float(ssim_t.item())}

"""Deterministic synthetic inputs shared by the oracle, golden generator and parity tests.

Same recipe as ``oracle/gen_golden.py:make_inputs`` (host generator, explicit seed):
real ~ U[-1,1] (the range Normalize(0.5,0.5) produces, dataset_code.py:147-150),
eps_* ~ N(0,1) standing in for the three randn_like draws of vaegan_code.py:77,91,92.
"""
import torch


def make_inputs(B, S, seed, latent=100):
    g = torch.Generator().manual_seed(seed)
    real = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    eps_z = torch.randn(B, latent, generator=g)
    eps_real = torch.randn(B, 3, S, S, generator=g)
    eps_recon = torch.randn(B, 3, S, S, generator=g)
    return real, eps_z, eps_real, eps_recon


def tstats(t):
    import numpy as np
    t = t.detach().double().flatten().cpu()
    idx = torch.linspace(0, t.numel() - 1, min(16, t.numel())).long()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()]), t[idx].numpy()

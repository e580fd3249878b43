"""Drop-in mirrors of the two loss modules the reference builds at vaegan_code.py:46-47
(``nn.BCELoss()``, ``nn.MSELoss(reduction='mean')``), running the fused HIP loss kernels.
Optional: the stock torch modules also work on the engine's outputs (they are ordinary tensors)."""
import torch
import torch.nn as nn

from . import ops


class _BCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, target):
        if target.numel() != p.numel():
            raise ValueError(f"Using a target size ({tuple(target.shape)}) that is different to the input size "
                             f"({tuple(p.shape)}) is deprecated. Please ensure they have the same size.")
        t0 = float(target.flatten()[0])          # the reference only uses constant soft labels (0.9 / 0.1)
        loss = torch.empty(1, dtype=torch.float32, device=p.device)
        dp = ops.bce_forward_backward(p.contiguous(), t0, 1.0, loss, False, True)
        ctx.save_for_backward(dp)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        return dp * g, None


class BCELoss(nn.Module):
    """nn.BCELoss() for a constant target vector (vaegan_code.py:88-89 uses torch.full labels)."""

    def forward(self, input, target):
        if not bool((target == target.flatten()[0]).all()):
            raise NotImplementedError("vaegan_amd.BCELoss supports the reference's constant soft labels only")
        return _BCEFn.apply(input, target)


class _MSEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        da = ops.mse_forward_backward(a.contiguous(), b.contiguous(), 1.0, loss, True)
        ctx.save_for_backward(da)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (da,) = ctx.saved_tensors
        ga = da * g if ctx.needs_input_grad[0] else None
        gb = -da * g if ctx.needs_input_grad[1] else None
        return ga, gb


class MSELoss(nn.Module):
    def __init__(self, reduction="mean"):
        super().__init__()
        if reduction != "mean":
            raise NotImplementedError("only reduction='mean' (vaegan_code.py:47)")

    def forward(self, input, target):
        return _MSEFn.apply(input, target)

"""vg_head_backward against vg_bce[_pair]_forward_backward + vg_dot_sigmoid_backward + vg_dot_wgrad, output by output."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from importlib import import_module
import vaegan_amd  # noqa
ops = import_module("vae-gan-based-model-for-image-generation-and-denoising_amd.ops")
G = import_module("vae-gan-based-model-for-image-generation-and-denoising_amd.geometry")
torch.manual_seed(0)
for dt, tdt in ((G.F32, torch.float32), (G.BF16, torch.bfloat16)):
    for B, groups in ((8, 2), (128, 2), (128, 1), (37, 1)):
        C, HW = 512, 16
        K = C * HW
        R = B * groups
        p = torch.rand(R, device="cuda") * 0.98 + 0.01
        x = torch.randn(R, 4, 4, C, device="cuda").to(tdt)
        w = (torch.randn(K, device="cuda") * 0.02).to(tdt)
        loss_a = torch.zeros(1, device="cuda"); loss_b = torch.zeros(1, device="cuda")
        dw_a = torch.zeros(1, C, 4, 4, device="cuda"); dw_b = torch.zeros(1, C, 4, 4, device="cuda")
        if groups == 2:
            dp = torch.empty_like(p)
            ops.bce_pair_forward_backward(p, 0.9, 0.1, 1.0, loss_a, dp)
        else:
            dp = ops.bce_forward_backward(p, 0.9, 0.1, loss_a, False, True)
        dx_a, dl_a = ops.dot_sigmoid_backward(p, dp, w, R, K, dt, True, x)
        ops.dot_wgrad(x, dl_a, dw_a, R, K, C, HW, False, dt)
        dx_b = ops.head_backward(p, x, w, B, groups, 0.9, 0.1, 1.0 if groups == 2 else 0.1, loss_b, False, dw_b, False, K, C, HW, dt, True)
        torch.cuda.synchronize()
        print(dt, B, groups, "loss", torch.equal(loss_a, loss_b), float(loss_a), float(loss_b),
              "dx", torch.equal(dx_a, dx_b), float((dx_a.float() - dx_b.float()).abs().max()),
              "dw", torch.equal(dw_a, dw_b), float((dw_a - dw_b).abs().max()))

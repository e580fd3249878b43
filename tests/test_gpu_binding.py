"""GPU: the PyTorch-ROCm custom-op face (torch.ops.vaegan.*, csrc_torch/vaegan_torch_ops.cpp: TORCH_LIBRARY over the
same C entry points) against the ctypes binding: one training iteration and the Adam update must be bit-identical, and
the ops follow PyTorch's error convention (RuntimeError via TORCH_CHECK)."""
import importlib

import pytest
import torch

import vaegan_amd as V
from _inputs import make_inputs
from test_gpu_parity import build

pytestmark = pytest.mark.gpu
DEV = "cuda"
ops = importlib.import_module("vae-gan-based-model-for-image-generation-and-denoising_amd.ops")


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_training_iteration_is_bit_identical_under_both_bindings(dtype):
    res = []
    try:
        for binding in ("ctypes", "torchops"):
            ops.set_binding(binding)
            e, g, d, tr = build(64, dtype=dtype)
            for step in range(2):
                real, ez, er, ec = (t.to(DEV) for t in make_inputs(8, 64, 7064 + step))
                l = tr.train_step(real, 60, ez, er, ec)
            res.append((l[:5].cpu().clone(), tr.opt_E.flat_p.cpu().clone(), tr.opt_G.flat_p.cpu().clone(),
                        tr.opt_D.flat_p.cpu().clone(), tr.opt_G.exp_avg_sq.cpu().clone()))
    finally:
        ops.set_binding("ctypes")
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_custom_ops_are_registered_and_raise_runtime_errors():
    t = ops.torch_ops()
    for name in ("gather_gemm", "wgrad", "adam_step", "bn_act_forward", "pack_weights_multi", "wgrad_ws_bytes"):
        assert hasattr(t, name)
    p = torch.zeros(64, device=DEV)
    st = torch.zeros(4, device=DEV)
    with pytest.raises(RuntimeError, match="buffer sizes differ"):
        t.adam_step(p, torch.zeros(32, device=DEV), p.clone(), p.clone(), 2e-4, 0.9, 0.999, 1e-8, 1.0, st)
    with pytest.raises(RuntimeError):                                       # no CPU implementation is registered
        t.adam_step(torch.zeros(64), torch.zeros(64), torch.zeros(64), torch.zeros(64), 2e-4, 0.9, 0.999, 1e-8, 1.0,
                    torch.zeros(4))
    with pytest.raises(RuntimeError, match="39 integers"):
        t.gather_gemm(p, p, p, None, None, None, p, None, [1, 2, 3], 0.0, 0.0, 0)
    # a working call: torch.optim.Adam's first step on a constant gradient moves every weight by -lr
    g = torch.ones(64, device=DEV)
    m, v = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)
    t.adam_step(p, g, m, v, 2e-4, 0.9, 0.999, 1e-8, 1.0, st)
    torch.testing.assert_close(p.cpu(), torch.full((64,), -2e-4), rtol=1e-5, atol=0)

"""Prints the measured distances the bf16 bounds of tests/test_gpu_configs.py are set from (run on the MI355X box):
  * state after iteration 1 of the bf16 engine vs the reference checksums (S=256, B=4 fixture), worst per class;
  * per-tensor relative Frobenius error of the bf16 S=64 B=128 replayed-graph gradients vs the fp64 oracle."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import vaegan_ref as R  # noqa: E402
from _inputs import make_inputs  # noqa: E402
from test_gpu_parity import DEV, build, check_state_after_first_iteration  # noqa: E402
from test_gpu_configs import frob, oracle_grads  # noqa: E402

torch.set_num_threads(min(16, os.cpu_count() or 1))
gold = np.load(os.path.join(ROOT, "tests", "golden", "steps_S256_B4_e0.npz"))
e, g, d, tr = build(256, dtype="bf16")
real, ez, er, ec = make_inputs(4, 256, 5040)
tr.train_step(real.to(DEV), 0, ez.to(DEV), er.to(DEV), ec.to(DEV))
o64 = R.RefVAEGAN(img_size=256, seed=42).double_()
o64.train_step(real, ez, er, ec, 0)
print("bf16 S=256 B=4 state after iteration 1:", check_state_after_first_iteration(gold, e, g, d, tr, o64, tols=(10.0, 10.0, 10.0)))

S, B, seed = 64, 128, 1234
e, g, d, tr = build(S, dtype="bf16", lr=0.0)
dev_in = [t.to(DEV) for t in make_inputs(B, S, seed)]
for _ in range(3):
    tr.train_step_graphed(dev_in[0], 60, *dev_in[1:])
torch.cuda.synchronize()
hip = {f"{n}.{k}": p.grad.double().cpu() for n, m in (("E", e), ("G", g), ("D", d)) for k, p in m.named_parameters()}
g32, g64 = oracle_grads(S, B, seed, False), oracle_grads(S, B, seed, True)
rows = sorted(((frob(hip[k], r), frob(g32[k], r), k) for k, r in g64.items() if float(r.abs().max()) >= 1e-6), reverse=True)
for err, cal, k in rows:
    print(f"{k:40s} bf16 {err:.2e}   cpu-fp32 {cal:.2e}")

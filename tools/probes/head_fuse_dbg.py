import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from test_gpu_parity import build, DEV
from _inputs import make_inputs
outs = []
for fused in (True, False, True, False):
    e, g, d, tr = build(64, dtype="fp32")
    tr.fuse_head_backward = fused
    ls = []
    for step in range(2):
        real, ez, er, ec = (t.to(DEV) for t in make_inputs(8, 64, 4400 + step))
        ls.append(tr.train_step(real, 60, ez, er, ec)[:5].clone())
    torch.cuda.synchronize()
    outs.append((torch.stack(ls).cpu(), tr.opt_D.flat_p.cpu().clone(), tr.opt_D.flat_g.cpu().clone(), tr.opt_G.flat_p.cpu().clone()))
for i in range(1, 4):
    print("run", i, "vs 0:", [(float((a - b).abs().max())) for a, b in zip(outs[0], outs[i])], (outs[0][0] - outs[i][0]))

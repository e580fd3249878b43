"""BatchNorm(+act) backward timing for the S=64 B=128 step's small-tensor shapes: ops.bn_act_backward as the engine
calls it (one-launch path where it qualifies; VG_BN_FUSED_BWD=0 forces reduce -> finalize -> apply).

    python tools/bn_bench.py [reps=50]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from importlib import import_module

PKG = "vae-gan-based-model-for-image-generation-and-denoising_amd"
ops = import_module(PKG + ".ops")
G = import_module(PKG + ".geometry")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = "cuda"
shapes = [("G0", 2048, 1, 1024), ("G1", 8192, 1, 512), ("G2", 32768, 1, 256), ("D3x2", 2048, 2, 512), ("D3", 2048, 1, 512),
          ("D2x2", 8192, 2, 256), ("D2", 8192, 1, 256), ("D1", 32768, 1, 128), ("E1", 25088, 1, 64), ("E2", 4608, 1, 128),
          ("E3", 512, 1, 256)]
tot = 0.0
for name, rpg, groups, C in shapes:
    rows = rpg * groups
    x = torch.randn(rows, C, device=dev).to(torch.bfloat16)
    dy = torch.randn(rows, C, device=dev).to(torch.bfloat16)
    co = torch.rand(groups, 4, C, device=dev) + 0.5
    gamma = torch.ones(C, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    fn = lambda: ops.bn_act_backward(x, dy, co, rows, C, rows, gamma, 2, 0.2, dg, db, False, G.BF16)
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    gr.replay()
    torch.cuda.synchronize()
    e0.record()
    gr.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    tot += us
    print(f"{name:5s} rows/group {rpg:6d} groups {groups} C {C:5d}: {us:7.2f} us", flush=True)
print(f"total {tot:.1f} us")

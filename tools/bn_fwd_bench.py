"""BatchNorm forward (finalize + normalise + activation) timing for the S=64 B=128 step's shapes: one-launch path
(VG_BN_FUSED_FWD=1, where it qualifies) against finalize -> apply.   python tools/bn_fwd_bench.py [reps=50]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from importlib import import_module
PKG = "vae-gan-based-model-for-image-generation-and-denoising_amd"
ops = import_module(PKG + ".ops"); G = import_module(PKG + ".geometry")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = "cuda"
# name, rows per group, groups, C, slab rows per group (M tile of the producing GEMM)
shapes = [("G0", 2048, 1, 1024, 16), ("G1", 8192, 1, 512, 64), ("G2", 32768, 1, 256, 128), ("G3", 131072, 1, 128, 512),
          ("D3x2", 2048, 2, 512, 32), ("D3", 2048, 1, 512, 32), ("D2x2", 8192, 2, 256, 64), ("D2", 8192, 1, 256, 128),
          ("D1x2", 32768, 2, 128, 256), ("D1", 32768, 1, 128, 256), ("E1", 25088, 1, 64, 196), ("E2", 4608, 1, 128, 36), ("E3", 512, 1, 256, 8)]
for mode in ("1", "0"):
    os.environ["VG_BN_FUSED_FWD"] = mode
    tot = 0.0
    print("== VG_BN_FUSED_FWD=" + mode)
    for name, rpg, groups, C, npg in shapes:
        rows, nparts = rpg * groups, npg * groups
        x = torch.randn(rows, C, device=dev).to(torch.bfloat16)
        stats = torch.rand(nparts * 2 * C, device=dev)
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        def fn():
            r = ops.bn_finalize_act_forward(x, stats, nparts, C, rows, gamma, beta, rm, rv, 0.1, 1e-5, 2, 0.2, G.BF16, groups=groups)
            if r is None:
                co = ops.bn_finalize(stats, nparts, C, rows, gamma, beta, rm, rv, 0.1, 1e-5, dev, groups=groups)
                ops.bn_act_forward(x, co, rows, C, 2, 0.2, G.BF16)
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(reps): fn()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        tot += us
        print(f"{name:5s} rows/group {rpg:6d} groups {groups} C {C:5d} slab rows {npg:4d}: {us:7.2f} us", flush=True)
    print(f"total {tot:.1f} us")

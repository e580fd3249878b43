"""Per-shape timing of the BatchNorm backward: one launch with a grid-wide exchange (csrc/bn_onepass.hip) against the
column-reduce -> finalize -> apply launches (VG_BN_ONEPASS=0).  GPU.  python tools/bn1_bench.py"""
import os, sys, importlib
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vaegan_amd  # noqa
PKG = "vae-gan-based-model-for-image-generation-and-denoising_amd"
ops = importlib.import_module(PKG + ".ops")
G = importlib.import_module(PKG + ".geometry")
DEV = "cuda"
shapes = [("G0", 2048, 1024, 1), ("G1", 8192, 512, 1), ("G2", 32768, 256, 1), ("D1 2B", 65536, 128, 2), ("D2 2B", 16384, 256, 2),
          ("D3 2B", 4096, 512, 2), ("D1 B", 32768, 128, 1), ("D2 B", 8192, 256, 1), ("D3 B", 2048, 512, 1),
          ("E0", 123008, 32, 1), ("E1", 25088, 64, 1), ("E2", 4608, 128, 1), ("E3", 512, 256, 1)]
big = torch.empty(64 << 20, dtype=torch.float32, device=DEV)          # 256 MB: flushes L2 / most of the Infinity Cache between calls
for name, rows, C, groups in shapes:
    x = torch.randn(rows, C, device=DEV).to(torch.bfloat16)
    dy = torch.randn(rows, C, device=DEV).to(torch.bfloat16)
    gamma = torch.ones(C, device=DEV)
    rpg = rows // groups
    xs = x.float().view(groups, rpg, C)
    mean, var = xs.mean(1), xs.var(1, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    co = torch.stack([mean, invstd, invstd, -mean * invstd], 1).contiguous()
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    res = {}
    for mode in ("0", "1"):
        os.environ["VG_BN_ONEPASS"] = mode
        ops.reload_switches()
        for cold in (False, True):
            ts = []
            for it in range(12):
                if cold:
                    big.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.bn_act_backward(x, dy, co, rows, C, rows, gamma, 2, 0.2, dg, db, False, G.BF16)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts = sorted(ts[2:])
            res[(mode, cold)] = ts[len(ts) // 2]
    mb = rows * C * 2 / 1e6
    print(f"{name:6s} rows {rows:7d} C {C:5d} g {groups} |Y| {mb:6.1f} MB   3-launch warm {res[('0', False)]:6.1f} cold {res[('0', True)]:6.1f}   one-launch warm {res[('1', False)]:6.1f} cold {res[('1', True)]:6.1f} us", flush=True)

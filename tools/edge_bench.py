"""Kernel-only timing of the edge layers of one configuration (vg_tnconv, narrow-K conv) and of the gather-GEMM
launches they replace (VG_EDGE=0).   python tools/edge_bench.py [S=64] [B=128]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from importlib import import_module
import vaegan_amd as V
PKG = "vae-gan-based-model-for-image-generation-and-denoising_amd"
ops = import_module(PKG + ".ops"); G = import_module(PKG + ".geometry")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = "cuda"; dt = G.BF16; reps = 20
V.configure_seed(42)
e = V.Encoder([3, S, S], 100, dtype="bf16").to(dev); g = V.Generator(nz=100, img_size=S, dtype="bf16").to(dev)
d = V.Discriminator(img_size=S, dtype="bf16").to(dev)

def timeit(fn, fam):
    fn(); torch.cuda.synchronize()
    t = ops.KernelTimer(); ops.set_timer(t)
    for _ in range(reps): fn()
    torch.cuda.synchronize(); ops.set_timer(None)
    s = t.summary()
    return {k: (v["ms"] * 1e3 / max(v["launches"], 1), v["launches"] // reps) for k, v in s.items() if v["launches"]}

for name, net, i, what, mult in (("G.last fprop", g, len(g._engine.stages) - 1, "fprop", 1), ("D.0 dgrad", d, 0, "dgrad", 1),
                                 ("D.0 fprop", d, 0, "fprop", 2), ("D.0 fprop", d, 0, "fprop", 1), ("E.0 fprop", e, 0, "fprop", 1),
                                 ("G.last dgrad", g, len(g._engine.stages) - 1, "dgrad", 1)):
    eng = net._engine; st = eng.stages[i]; Bx = B * mult
    packs = eng._ensure_packed()
    fl, by = st.alg(Bx, dt)
    gg, pk = eng.spec(i, Bx, what)
    X = torch.randn(gg.B, gg.IH, gg.IW, gg.IC, device=dev).to(torch.bfloat16)
    res = {}
    tn = eng.tn(i, Bx, what)
    if tn is not None:
        res["tnconv"] = timeit(lambda: ops.tnconv(tn[0], X, packs[i]["tn_" + what], alg=(fl, by)), "edge")
    for env in ("1", "0"):
        os.environ["VG_EDGE"] = env
        ops.reload_switches()                      # the library reads its switches once, at load
        bias = packs[i]["bias"] if what == "fprop" else None
        res["gather_gemm VG_EDGE=" + env] = timeit(lambda: ops.gather_gemm(gg, X, packs[i][what], dt, bias=bias, want_stats=(st.bn is not None and what == "fprop"), alg=(fl, by)), None)
    os.environ["VG_EDGE"] = "1"
    ops.reload_switches()
    print(f"{name:14s} B={Bx:4d} alg {by/1e6:7.1f} MB {fl/1e9:6.2f} GF | " + " | ".join(f"{k}: " + ", ".join(f"{f} {t:.1f}us x{n}" for f, (t, n) in v.items()) for k, v in res.items()), flush=True)

"""Per-layer KERNEL timing at the bench workload (HIP events around each GEMM-class kernel on its stream, no
host launch overhead): fprop / dgrad / wgrad TFLOP/s for every stage of E, G, D.

    python tools/layer_bench.py [S=64] [B=128] [dtype=bf16] [reps=20] [filter-regex]

VG_LIB_PATH selects an alternative build of libvaegan_hip.so (ablation builds)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from importlib import import_module
import vaegan_amd as V

PKG = "vae-gan-based-model-for-image-generation-and-denoising_amd"
ops = import_module(PKG + ".ops")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
flt = re.compile(sys.argv[5]) if len(sys.argv) > 5 else None
dev = "cuda"
V.configure_seed(42)
e = V.Encoder([3, S, S], 100, dtype=dtype).to(dev)
g = V.Generator(nz=100, img_size=S, dtype=dtype).to(dev)
d = V.Discriminator(img_size=S, dtype=dtype).to(dev)


def timeit(fn, fam):
    fn()
    torch.cuda.synchronize()
    t = ops.KernelTimer()
    ops.set_timer(t)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ops.set_timer(None)
    s = t.summary()
    fams = [fam] if fam == "wgrad" else [k for k in s if k != "wgrad"]      # edge / fp8 launches count for their layer
    n = sum(s[k]["launches"] for k in fams)
    return sum(s[k]["ms"] for k in fams) * 1e-3 / max(n, 1)


tot = {"fprop": 0, "dgrad": 0, "wgrad": 0}
for name, net, passes in (("E", e, ((1, 1),)), ("G", g, ((1, 1),)), ("D", d, ((2, 2), (1, 1)))):
    eng = net._engine
    dt = eng.dtype
    packs = eng._ensure_packed()
    for i, st in enumerate(eng.stages):
        if st.kind == "head":
            continue
        for mult, count in passes:
            Bx = B * mult
            tag = f"{name}{i}" + ("x2" if mult == 2 else "")
            if flt is not None and not flt.search(tag):
                continue
            fl, by = st.alg(Bx, dt)
            gg, pk = eng.spec(i, Bx, "fprop")
            ggd, pkd = eng.spec(i, Bx, "dgrad")
            wg = eng.spec(i, Bx, "wgrad")
            tdt = ops.TORCH_DT[dt]
            X = torch.randn(gg.B, gg.IH, gg.IW, gg.IC, device=dev).to(tdt)
            DY = torch.randn(ggd.B, ggd.IH, ggd.IW, ggd.IC, device=dev).to(tdt)
            Wshape = st.conv.weight.shape if st.kind != "linear2" else (st.cout, st.conv.weight.shape[1])
            dW = torch.zeros(Wshape, device=dev)
            tf = timeit(lambda: ops.gather_gemm(gg, X, packs[i]["fprop"], dt, want_stats=st.bn is not None and not pk.tap_in_n), "gather_gemm")
            td = timeit(lambda: ops.gather_gemm(ggd, DY, packs[i]["dgrad"], dt), "gather_gemm")
            P, Q = (X, DY) if st.kind == "convT" else (DY, X)
            tw = timeit(lambda: ops.wgrad(wg, P, Q, dW, False, dt), "wgrad")
            k = 2 if (name == "D" and mult == 2) else 1
            tot["fprop"] += tf * k
            tot["dgrad"] += td * k
            tot["wgrad"] += tw * k
            print(f"{tag:5s} {st.kind:7s} {st.cin:4d}->{st.cout:4d} k{st.k} {st.hin:3d}->{st.hout:3d} GF {fl/1e9:7.2f} | "
                  f"fprop {tf*1e6:7.1f}us {fl/tf/1e12:6.1f}TF M={gg.M}x{gg.nphase} N={gg.N} K={gg.Kp} | "
                  f"dgrad {td*1e6:7.1f}us {fl/td/1e12:6.1f}TF M={ggd.M}x{ggd.nphase} N={ggd.N} K={ggd.Kp} | "
                  f"wgrad(main) {tw*1e6:7.1f}us {fl/tw/1e12:6.1f}TF", flush=True)
print("per-step kernel totals (D: 2 grouped iterations + 1 single pass):", {k: f"{v*1e3:.3f} ms" for k, v in tot.items()})

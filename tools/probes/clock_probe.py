"""In-kernel clock and phase times of the patch gather-GEMM (MI355X_MICROARCH.md, DVFS give-back item 6).

Needs the diagnostic build (tools/build_ablate.sh stamps "-DVG_DBG_STAMPS"; VG_LIB_PATH=scratch/libvg_stamps.so): thread 0 of
every workgroup stamps s_memtime / s_memrealtime at entry, before the main loop, after it and at the end.  Each layer's launch
runs back to back for `seconds` on random data, then the stamps of the LAST launch are read:
clock = d(memtime) / d(memrealtime) x 100 MHz over the main loop, median over workgroups.

    VG_LIB_PATH=$PWD/scratch/libvg_stamps.so python tools/probes/clock_probe.py [seconds=2] [layers=G1,G2,G3,G4,D1,D2]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import torch
from importlib import import_module
import vaegan_amd as V

PKG = "vae-gan-based-model-for-image-generation-and-denoising_amd"
ops = import_module(PKG + ".ops")
L = import_module(PKG + "._lib")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
layers = (sys.argv[2] if len(sys.argv) > 2 else "G1,G2,G3,G4,D1,D2").split(",")
lib = L.load()
rd = lib.vg_debug_stamps
rd.restype = ctypes.c_int
rd.argtypes = [ctypes.c_void_p, ctypes.c_int]

V.configure_seed(42)
g = V.Generator(nz=100, img_size=64, dtype="bf16").to("cuda")
d = V.Discriminator(img_size=64, dtype="bf16").to("cuda")
B = 128


def stamps():
    buf = np.zeros(8192 * 8, dtype=np.uint64)
    rc = rd(buf.ctypes.data, buf.size)
    assert rc == 0, rc
    return buf.reshape(8192, 4, 2).astype(np.int64)


for layer in layers:
    net = g if layer[0] == "G" else d
    mult = 2 if layer[0] == "D" else 1
    idx = int(layer[1])
    eng = net._engine
    packs = eng._ensure_packed()
    st = eng.stages[idx]
    for kind in ("fprop", "dgrad"):
        gg, pk = eng.spec(idx, B * mult, kind)
        X = torch.randn(gg.B, gg.IH, gg.IW, gg.IC, device="cuda").to(torch.bfloat16)
        fl, _ = st.alg(B * mult, eng.dtype)
        stats = kind == "fprop" and st.bn is not None
        fn = lambda: ops.gather_gemm(gg, X, packs[idx][kind], eng.dtype, want_stats=stats)
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0, n = time.time(), 0
        e0.record()
        while time.time() - t0 < secs:
            for _ in range(500):
                fn()
            n += 500
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        wall = e0.elapsed_time(e1) / n * 1e3
        s = stamps()
        live = s[:, 3, 1] >= s[:, 3, 1].max() - 100000          # stamped within the last 1 ms (100 MHz ticks)
        s = s[live]
        if len(s) == 0 or (s[:, 2, 1] - s[:, 1, 1]).min() <= 0:
            print(f"{layer} {kind}: {wall:.1f} us/launch sustained; no stamps (not a patch kernel)")
            continue
        clk = (s[:, 2, 0] - s[:, 1, 0]) / (s[:, 2, 1] - s[:, 1, 1]) * 0.1      # GHz
        us = lambda a: np.median(a) / 100.0
        t_first = s[:, 0, 1].min()
        print(f"{layer} {kind}: {wall:.1f} us/launch sustained ({fl / wall / 1e6:.0f} TF/s), {len(s)} workgroups | in-kernel clock "
              f"{np.median(clk):.2f} GHz (min {clk.min():.2f} max {clk.max():.2f}) | per workgroup, median us: entry->loop "
              f"{us(s[:, 1, 1] - s[:, 0, 1]):.2f}, main loop {us(s[:, 2, 1] - s[:, 1, 1]):.2f} ({np.median(s[:, 2, 0] - s[:, 1, 0]):.0f} cyc), "
              f"epilogue {us(s[:, 3, 1] - s[:, 2, 1]):.2f} | entry spread {(s[:, 0, 1].max() - t_first) / 100.0:.2f} us, first entry -> last end "
              f"{(s[:, 3, 1].max() - t_first) / 100.0:.2f} us", flush=True)

"""Operand-pack timing per network (one vg_pack_weights_multi launch each) at the bench workload, HIP-event timed.

    python tools/pack_bench.py [S=64] [dtype=bf16] [reps=20]

Prints microseconds per launch, the bytes the launch has to move at the least (every f32 parameter read once, every
packed operand written once) and the rate that corresponds to.  VG_LIB_PATH selects an alternative build."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from importlib import import_module
import vaegan_amd as V

PKG = "vae-gan-based-model-for-image-generation-and-denoising_amd"
ops = import_module(PKG + ".ops")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = "cuda"
V.configure_seed(42)
nets = (("E", V.Encoder([3, S, S], 100, dtype=dtype).to(dev)), ("G", V.Generator(nz=100, img_size=S, dtype=dtype).to(dev)),
        ("D", V.Discriminator(img_size=S, dtype=dtype).to(dev)))
for name, net in nets:
    eng = net._engine
    packs = eng._ensure_packed()
    src = sum(st.conv.weight.numel() * 4 for st in eng.stages if st.kind != "linear2")
    out = 0
    for i in packs:
        for k, v in packs[i].items():
            if isinstance(v, torch.Tensor) and k in ("fprop", "dgrad", "tn_fprop", "tn_dgrad"):
                out += v.numel() * v.element_size()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    call = lambda: ops.pack_weights_multi(eng._pack_table, eng._pack_n, eng._pack_max, eng.dtype)
    call()
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        call()
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 1e3 / reps
    print(f"{name}: {eng._pack_n} descriptors, {eng._pack_max} tiles, {us:7.1f} us/launch, min bytes {(src + out) / 1e6:6.1f} MB "
          f"(src {src / 1e6:.1f} + out {out / 1e6:.1f}) -> {(src + out) / us / 1e6:5.2f} TB/s", flush=True)

"""In-kernel s_memtime stamps of ggs_kernel (ablation build scratch/libvg_gs_stamps.so via VG_LIB_PATH): cycles a wave
spends per tile in wait+barrier | patch issue | MFMA chunk | tail | epilogue+loop."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from importlib import import_module
import vaegan_amd as V
PKG = "vae-gan-based-model-for-image-generation-and-denoising_amd"
ops = import_module(PKG + ".ops"); G = import_module(PKG + ".geometry")
B = 128
g = V.Generator(nz=100, img_size=64, dtype="bf16").to("cuda")
eng = g._engine; i = 4; st = eng.stages[i]
packs = eng._ensure_packed()
gg, pk = eng.spec(i, B, "fprop")
X = torch.randn(gg.B, gg.IH, gg.IW, gg.IC, device="cuda").to(torch.bfloat16)
for _ in range(3):
    Y, _, _ = ops.gather_gemm(gg, X, packs[i]["fprop"], G.BF16, want_stats=True)
torch.cuda.synchronize()
d = Y.view(-1).view(torch.int64)[: 256 * 8].view(256, 8).cpu().double()
names = ["wait+barrier", "issue", "compute", "tail", "epilogue+loop"]
tot = d[:, :5].sum(1)
print("tiles/WG 16, chunks 64; 100 MHz s_memtime ticks -> us = ticks / 100")
for k, n in enumerate(names):
    print(f"{n:14s} mean {d[:, k].mean() / 100:8.2f} us  min {d[:, k].min() / 100:8.2f}  max {d[:, k].max() / 100:8.2f}")
print(f"total          mean {tot.mean() / 100:8.2f} us")

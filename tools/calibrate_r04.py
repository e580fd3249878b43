"""bf16 engine vs the bf16-STORAGE emulation of the oracle (oracle/vaegan_ref_bf16.py), per parameter tensor.
GPU.  python tools/calibrate_r04.py [S=64] [B=128]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
import vaegan_ref as R, vaegan_ref_bf16 as RB
from _inputs import make_inputs
import vaegan_amd as V
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
torch.set_num_threads(16)
def frob(a, r): return float((a - r).norm() / r.norm().clamp_min(1e-300))
V.configure_seed(42)
e = V.Encoder([3, S, S], 100, dtype="bf16"); g = V.Generator(nz=100, img_size=S, dtype="bf16"); d = V.Discriminator(img_size=S, dtype="bf16")
g.apply(V.weights_init); d.apply(V.weights_init)
for m in (e, g, d): m.to("cuda")
tr = V.VAEGANTrainer(e, g, d, *(V.Adam(m.parameters(), lr=0.0) for m in (e, g, d))); tr.train()
inp = make_inputs(B, S, 1234)
dev = [t.to("cuda") for t in inp]
for _ in range(3): out = tr.train_step_graphed(dev[0], 60, *dev[1:])
torch.cuda.synchronize()
hl = tr.loss_dict(out.clone(), 60)
hip = {f"{n}.{k}": p.grad.double().cpu() for n, m in (("E", e), ("G", g), ("D", d)) for k, p in m.named_parameters()}
t0 = time.time()
o = R.RefVAEGAN(img_size=S, seed=42, lr=0.0).double_(); l64 = o.train_step(*inp, 60)
g64 = {f"{n}.{k}": st[k].grad.double() for n, st in (("E", o.E), ("G", o.G), ("D", o.D)) for k in R.trainable_keys(st)}
em = RB.RefVAEGANbf16(img_size=S, seed=42, lr=0.0); le = em.train_step(*inp, 60)
gem = {f"{n}.{k}": st[k].grad.double() for n, st in (("E", em.E), ("G", em.G), ("D", em.D)) for k in R.trainable_keys(st)}
print("cpu seconds", round(time.time() - t0, 1))
print("losses: hip vs emulation / hip vs fp64 / emulation vs fp64")
for k in V.LOSS_NAMES:
    print(f"  {k:12s} {abs(hl[k]-le[k])/abs(le[k]):.2e}  {abs(hl[k]-l64[k])/abs(l64[k]):.2e}  {abs(le[k]-l64[k])/abs(l64[k]):.2e}")
rows = []
for k, r in g64.items():
    if float(r.abs().max()) < 1e-6: continue
    rows.append((frob(hip[k], gem[k]), frob(hip[k], r), frob(gem[k], r), k))
rows.sort(reverse=True)
print("gradients, relative Frobenius: hip vs emulation | hip vs fp64 | emulation vs fp64")
for a, b, c, k in rows: print(f"  {k:28s} {a:.2e} | {b:.2e} | {c:.2e}")

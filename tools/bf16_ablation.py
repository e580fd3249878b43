"""Which bf16-stored tensor costs the bf16 engine its gradient accuracy?  (round-3 review, weak #1 / round-4 item 2)

    python tools/bf16_ablation.py [B=32] [S=64]

CPU only.  Runs oracle/vaegan_ref_bf16.py (exact arithmetic + bf16 rounding at the engine's storage points) with ONE class
of storage points -- or one single tensor -- left unrounded at a time and prints the relative Frobenius distance of a few
parameter gradients from the fp64 oracle.  The row whose distance collapses names the tensor that carries the error.
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import vaegan_ref as R              # noqa: E402
import vaegan_ref_bf16 as RB        # noqa: E402
from _inputs import make_inputs     # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
torch.set_num_threads(min(8, os.cpu_count() or 1))
inp = make_inputs(B, S, 1234)
WATCH = ["G.main.0.weight", "G.main.3.weight", "G.main.12.weight", "G.main.15.weight", "E.cnn.0.conv.weight",
         "E.fc_mu.weight", "D.main.0.weight", "D.main.8.weight"]


def frob(a, r):
    return float((a - r).norm() / r.norm().clamp_min(1e-300))


o = R.RefVAEGAN(img_size=S, seed=42, lr=0.0).double_()
o.train_step(*inp, 60)
g64 = {f"{n}.{k}": st[k].grad.double().clone() for n, st in (("E", o.E), ("G", o.G)) for k in R.trainable_keys(st)}
# the Discriminator's gradients of its FIRST update (the oracle's .grad holds later passes on top): from the all-exact emulation
ex = RB.RefVAEGANbf16(img_size=S, seed=42, lr=0.0, keep_fp={"fwd", "bwd"})
ex.train_step(*inp, 60)
g64.update({f"D.{k}": v for k, v in ex.last_grads["D"].items()})
WATCH = [w for w in WATCH if w in g64]
nD = sum(1 for e_ in o.d_spec if e_[0] == "conv")
nG = sum(1 for e_ in o.g_spec if e_[0] == "convT")
rows = [("(all storage points rounded)", set()), ("fwd exact", {"fwd"}), ("bwd exact", {"bwd"}), ("weights exact", {"w"}),
        ("images exact", {"img"}), ("raw conv outputs Y exact", {"Y"}), ("activations A exact", {"A"}),
        ("all dX exact", {"dX"}), ("all dY exact", {"dY"}), ("dpre exact", {"dpre"}),
        ("D: dX + dY exact", {f"dX:D.{i}" for i in range(nD)} | {f"dY:D.{i}" for i in range(nD)}),
        ("G: dX + dY + dpre exact", {f"dX:G.{i}" for i in range(nG)} | {f"dY:G.{i}" for i in range(nG)} | {"dpre"})]
rows += [(f"dX:D.{i} exact", {f"dX:D.{i}"}) for i in range(nD)] + [(f"dY:D.{i} exact", {f"dY:D.{i}"}) for i in range(1, nD - 1)]
rows += [(f"dX:G.{i} exact", {f"dX:G.{i}"}) for i in range(nG)] + [(f"dY:G.{i} exact", {f"dY:G.{i}"}) for i in range(nG - 1)]
rows += [("deep small tensors exact (D.3, D.4 / G.0, G.1: dX and dY)",
          {f"d{t}:{n}.{i}" for t in "XY" for n, ii in (("D", (nD - 2, nD - 1)), ("G", (0, 1))) for i in ii})]
print(f"S={S} B={B}; relative Frobenius distance from the fp64 oracle")
print("%-58s" % "left unrounded" + "".join("%22s" % w[-20:] for w in WATCH))
t0 = time.time()
for name, keep in rows:
    e = RB.RefVAEGANbf16(img_size=S, seed=42, lr=0.0, keep_fp=keep)
    e.train_step(*inp, 60)
    got = {f"{n}.{k}": v for n in ("E", "G", "D") for k, v in e.last_grads[n].items()}
    print("%-58s" % name + "".join("%22.2e" % frob(got[w], g64[w]) for w in WATCH), flush=True)
print("seconds:", round(time.time() - t0, 1))

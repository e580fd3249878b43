"""How the CPU oracle (bench.py cpu_baseline) scales with torch threads on this box: S=64 B=128 fp32 training step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import vaegan_ref as R
from _inputs import make_inputs
avail = len(os.sched_getaffinity(0))
print("cores available", avail, flush=True)
inp = make_inputs(128, 64, 1234)
for n in (8, 16, 32, 64, 128, 256):
    if n > avail:
        break
    torch.set_num_threads(n)
    o = R.RefVAEGAN(img_size=64, seed=42)
    o.train_step(*inp, 60)
    t0 = time.time()
    for _ in range(2):
        o.train_step(*inp, 60)
    dt = (time.time() - t0) / 2
    print(f"threads {n:4d}: {dt:.3f} s/step = {128 / dt:.1f} img/s", flush=True)

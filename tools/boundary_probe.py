"""What does one dependent kernel boundary cost inside a replayed hipGraph on this box?  N trivial launches of the library
(vg_rng_advance: one thread) and N small real ones (vg_axpy over 64 K floats) captured into one graph; time per replay / N.
GPU.  python tools/boundary_probe.py"""
import os, sys, time, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vaegan_amd  # noqa
PKG = "vae-gan-based-model-for-image-generation-and-denoising_amd"
ops = importlib.import_module(PKG + ".ops")
L = importlib.import_module(PKG + "._lib")
dev = "cuda"
state = torch.zeros(2, dtype=torch.int64, device=dev)
a = torch.randn(1 << 16, device=dev); b = torch.randn(1 << 16, device=dev); c = torch.empty_like(a)
big_a = torch.randn(1 << 24, device=dev); big_b = torch.randn(1 << 24, device=dev); big_c = torch.empty_like(big_a)
lib = L.load()

def trivial(n):
    for _ in range(n):
        lib.vg_rng_advance(state.data_ptr(), L.stream_ptr())
def small(n):
    for _ in range(n):
        ops.axpy(a, b, 1.0, out=c)
def medium(n):                       # 64 MB read + 64 MB... (3 x 64 MB streams): a ~40 us HBM-bound kernel
    for _ in range(n):
        ops.axpy(big_a, big_b, 1.0, out=big_c)

def replay_us(fn, n, reps=30):
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(2); torch.cuda.synchronize()
        g.capture_begin(); fn(n); g.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

for name, fn in (("trivial (1 thread)", trivial), ("small (axpy 64K floats)", small), ("medium (axpy 16M floats)", medium)):
    t1, t2 = replay_us(fn, 100), replay_us(fn, 300)
    print(f"{name:28s} 100 launches {t1:8.1f} us, 300 launches {t2:8.1f} us -> {(t2 - t1) / 200:6.2f} us per additional dependent launch", flush=True)

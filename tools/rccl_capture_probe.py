"""Can this stack (PyTorch 2.10 / RCCL) capture gradient all-reduces INSIDE a hipGraph, and what does a captured collective
cost with ONE rank?  (round-3 review item 7)   python tools/rccl_capture_probe.py"""
import os, time, sys
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
flat = torch.randn(11 << 18, device=dev)            # 11.5 MB like D's gradient buffer
a = torch.randn(1 << 20, device=dev); b = torch.empty_like(a)
dist.all_reduce(flat); torch.cuda.synchronize()      # communicator warm-up outside any capture

def body(mode, n):
    for i in range(n):
        torch.mul(a, 1.0001, out=b)
        if mode == "sync":
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        elif mode == "async":
            w = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
            torch.add(b, 1.0, out=a)                 # work that may overlap
            w.wait()
        torch.add(b, 1.0, out=a)

def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6

N = 5
res = {}
for mode in ("none", "sync", "async"):
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            body(mode, N)                             # warm-up on the capture stream
            torch.cuda.synchronize()
            g.capture_begin(capture_error_mode="thread_local")
            body(mode, N)
            g.capture_end()
        torch.cuda.current_stream().wait_stream(s)
        res[mode] = timed(g.replay)
        print(f"captured mode={mode}: {res[mode]:.1f} us per replay of {N} iterations -> {res[mode] / N:.1f} us per iteration", flush=True)
    except Exception as ex:
        print(f"capture mode={mode} FAILED: {type(ex).__name__}: {str(ex)[:400]}", flush=True)
        try:
            g.capture_end()
        except Exception:
            pass
for mode in ("none", "sync", "async"):
    print(f"eager mode={mode}: {timed(lambda: body(mode, N)) / N:.1f} us per iteration", flush=True)
dist.destroy_process_group()

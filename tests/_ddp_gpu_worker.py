"""One rank of the multi-process GPU tests (tests/test_gpu_ddp.py).  Launched as a child process with
RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment; backend gloo (several ranks share the one GPU of
the test box, which RCCL refuses).  Runs `steps` iterations of the HIP trainer on this rank's shard of a fixed
global batch and writes losses, reduced gradients, parameters and BatchNorm buffers to an .npz.

usage: python _ddp_gpu_worker.py OUT.npz S GLOBAL_B STEPS SYNC_BN(0|1) GRAPH(0|1) LR
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import vaegan_amd as V  # noqa: E402
from _inputs import make_inputs  # noqa: E402


def main():
    out, S, GB, steps, sync_bn, graph = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), \
        int(sys.argv[5]), int(sys.argv[6])
    lr = float(sys.argv[7])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    backend = os.environ.get("VAEGAN_TEST_BACKEND", "gloo")       # "nccl" (= RCCL) only with one rank per GPU
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        V.configure_seed(42)
        e = V.Encoder([3, S, S], 100)
        g = V.Generator(nz=100, img_size=S)
        d = V.Discriminator(img_size=S)
        g.apply(V.weights_init), d.apply(V.weights_init)
        e.to("cuda"), g.to("cuda"), d.to("cuda")
        oE, oG, oD = (V.Adam(m.parameters(), lr=lr) for m in (e, g, d))
        red = V.GradReducer()
        red.attach(oE, oG, oD)
        red.broadcast_parameters(oE, oG, oD)
        tr = V.VAEGANTrainer(e, g, d, oE, oG, oD, reducer=red, sync_bn=bool(sync_bn))
        tr.train()
        Bl = GB // world
        sl = slice(rank * Bl, (rank + 1) * Bl)
        losses = []
        for s in range(steps):
            real, ez, er, ec = (t[sl].contiguous().to("cuda") for t in make_inputs(GB, S, 9100 + s))
            fn = tr.train_step_graphed if graph else tr.train_step
            losses.append(fn(real, 60, ez, er, ec)[:5].clone())
        torch.cuda.synchronize()
        lo = torch.stack(losses)
        dist.all_reduce(lo)                       # global loss = mean of the per-rank means (equal shards)
        lo /= world
        res = {"losses": lo.cpu().numpy(),
               # gradients in named_parameters() order (the flat buffer's placement is the optimizer's business)
               "grad_E": (torch.cat([p.grad.flatten() for p in e.parameters()]) / world).cpu().numpy(),
               "grad_G": (torch.cat([p.grad.flatten() for p in g.parameters()]) / world).cpu().numpy(),
               "par_E": oE.flat_p.cpu().numpy(), "par_G": oG.flat_p.cpu().numpy(), "par_D": oD.flat_p.cpu().numpy(),
               "stat_collectives": np.array(red.stat_collectives), "collectives": np.array(red.collectives),
               # buckets per optimizer (E, G, D) as the real GradReducer planned them, and hipGraph segments of a replay
               "buckets": np.array([len(red.buckets(o)) for o in (oE, oG, oD)]),
               "segments": np.array(0 if tr._graph is None else len(tr._graph[1]))}
        for name, net in (("E", e), ("G", g), ("D", d)):
            for k, v in net.state_dict().items():
                if "running" in k or "num_batches" in k:
                    res[f"buf_{name}.{k}"] = v.cpu().numpy()
        np.savez(out.replace(".npz", f".r{rank}.npz"), **res)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

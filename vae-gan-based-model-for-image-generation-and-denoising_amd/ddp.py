"""Data-parallel gradient reduction: one process per GPU, RCCL (torch.distributed backend "nccl")
over xGMI.  The reference has no distributed code at all; this is the new capability
BASELINE.json asks for.

Design for the MI355X node (8 GPUs, fully connected, 7 x ~153 GB/s links each): every optimizer
owns ONE flat fp32 gradient buffer (optim.Adam), so a network's gradients are reduced with a single
large all-reduce (E 3.6-43 MB, G 51 MB, D 11 MB) instead of dozens of per-tensor collectives -- the
latency-bound regime on point-to-point links.  SUM is used and the 1/world_size average is folded
into the Adam kernel's grad_scale, so no extra pass touches the gradients.  The generator's
reduction is issued asynchronously and overlaps the encoder's backward (trainer.py).

BatchNorm statistics stay per replica by default (standard DDP semantics, throughput mode).  Loss
normalisers are per-replica means; with equal per-replica batches the averaged gradient equals the
global-batch gradient.  VAEGANTrainer(sync_bn=True) switches every BatchNorm to statistics over the
GLOBAL batch (forward: sum / sum-of-squares, backward: sum dy / sum dy*xhat, each one small f64
all-reduce per layer), which makes an N-rank step equal the single-process step on the concatenated
batch -- the reference's semantics (it is single-process) -- at the price of ~100 latency-bound
collectives per step (parity mode; the iteration then runs eagerly, not as a replayed hipGraph).
"""
from typing import Dict, Optional

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, process_group=None):
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("GradReducer needs an initialised torch.distributed process group "
                               "(backend 'nccl' = RCCL on the MI355X node, 'gloo' for CPU tests)")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self._pending: Dict[int, object] = {}
        self.bytes_reduced = 0
        self.stat_collectives = 0

    def attach(self, *optimizers) -> None:
        """Fold the 1/world_size average into each optimizer's fused step."""
        for o in optimizers:
            o.grad_scale = 1.0 / self.world

    def reduce(self, opt) -> None:
        dist.all_reduce(opt.flat_g, op=dist.ReduceOp.SUM, group=self.group)
        self.bytes_reduced += opt.flat_g.numel() * 4

    def reduce_async(self, opt) -> None:
        self._pending[id(opt)] = dist.all_reduce(opt.flat_g, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.bytes_reduced += opt.flat_g.numel() * 4

    def wait(self, opt) -> None:
        w = self._pending.pop(id(opt), None)
        if w is not None:
            w.wait()

    def all_reduce_sum(self, t: torch.Tensor) -> None:
        """In-place SUM over ranks of a small statistics tensor (synchronised BatchNorm)."""
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        self.stat_collectives += 1

    def broadcast_parameters(self, *optimizers, src: int = 0) -> None:
        """Make every replica start from rank `src`'s weights (one broadcast per flat buffer)."""
        for o in optimizers:
            dist.broadcast(o.flat_p, src=src, group=self.group)

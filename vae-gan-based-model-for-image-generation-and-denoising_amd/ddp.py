"""Data-parallel gradient reduction: one process per GPU, RCCL (torch.distributed backend "nccl")
over xGMI.  The reference has no distributed code at all; this is the new capability
BASELINE.json asks for ("RCCL all-reduce of gradients over xGMI overlapped with backward").

Design for the MI355X node (8 GPUs, fully connected, 7 x ~153 GB/s links each):
  * every optimizer owns ONE flat fp32 gradient buffer (optim.Adam); SUM is used and the 1/world_size
    average is folded into the Adam kernel's grad_scale, so no extra pass touches the gradients;
  * the flat buffer is cut into BUCKETS in reverse layer order -- the order in which the backward pass
    finishes weight gradients -- of >= `bucket_bytes` (default 8 MB: large enough to be bandwidth- rather
    than latency-bound on point-to-point xGMI links, small enough that the first one leaves early).  A bucket's
    all-reduce is launched ASYNCHRONOUSLY the moment its last weight gradient has been enqueued
    (trainer.py cuts its hipGraph there) and runs on RCCL's stream under the rest of the backward pass;
    the optimizer step waits for its buckets only.
  * every bucket launch is a hipGraph CUT, and a cut costs ~29 us of idle GPU (a replayed graph starts ~27 us after
    its predecessor ends: DESIGN.md section 9), so the number of buckets is bounded: a tail smaller than `bucket_bytes`
    is glued to the bucket before it (it would be latency-bound anyway) and at most `max_buckets` (default 2) buckets
    are kept per optimizer, merging from the FRONT so that what is left for the end of the backward pass stays small.
    S=64: D = {11.2 MB} (twice per step), G = {G5..G2 11.0 MB | G1+G0 40.2 MB}, E = {3.6 MB}: 77 MB per step and GPU
    in 5 collectives = 5 cuts (round 2: 9).  A bucket is a contiguous slice of the flat buffer: reducing the slices is
    bit-identical to reducing the whole buffer (elementwise sums; tests/test_ddp_gloo.py).
    MEASURED ON HARDWARE: the overlap itself never was -- no 8-GPU node has been available to the build in rounds 1-3
    (SCALE_r01/r02 are skip records); on one GPU the RCCL calls run with a single rank (tests/test_gpu_ddp.py).

BatchNorm statistics stay per replica by default (standard DDP semantics, throughput mode).  Loss
normalisers are per-replica means; with equal per-replica batches the averaged gradient equals the
global-batch gradient.  VAEGANTrainer(sync_bn=True) switches every BatchNorm to statistics over the
GLOBAL batch (forward: sum / sum-of-squares, backward: sum dy / sum dy*xhat, each one small f64
all-reduce per layer), which makes an N-rank step equal the single-process step on the concatenated
batch -- the reference's semantics (it is single-process) -- at the price of ~100 latency-bound
collectives per step (parity mode).
"""
from typing import Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist

DEFAULT_BUCKET_BYTES = 8 << 20


DEFAULT_MAX_BUCKETS = 2
# Collectives captured INSIDE the hipGraph (round 4, RCCL): a bucket launch is a fork / join inside the graph, not a graph
# cut, so finer buckets cost nothing on the compute stream and leave less of each network's all-reduce exposed: the
# Discriminator's {head + last conv: 8.4 MB} leaves while its first three layers are still in their backward pass and only
# {2.6 MB} is waited for before its Adam step (twice per iteration); the Generator goes in four pieces.
INLINE_BUCKET_BYTES = 2 << 20
INLINE_MAX_BUCKETS = 4


def plan_buckets(offsets: Sequence[int], numels: Sequence[int], ready: Sequence[int], total: int,
                 bucket_bytes: int = DEFAULT_BUCKET_BYTES, max_buckets: int = 0) -> List[Tuple[int, int, int]]:
    """Cut a flat gradient buffer of `total` floats into contiguous buckets.

    offsets/numels: placement of every parameter in the buffer; ready[i]: the backward pass finishes parameter
    i's gradient at event ready[i], events counting DOWN (the stage index of the layer: the last layer's gradients
    come first).  Returns [(lo, hi, event)] in launch order: bucket (lo, hi) is complete once event `event` -- the
    smallest event of any parameter inside -- has happened.  Every float of [0, total) is in exactly one bucket
    (alignment padding rides along).  max_buckets > 0 bounds the number of buckets (= hipGraph cuts): a tail smaller
    than bucket_bytes is glued to the bucket before it, then the first buckets (in launch order) are merged until at
    most max_buckets are left; 0 keeps every bucket the byte rule produces."""
    spans = sorted((offsets[i], offsets[i] + (numels[i] + 3) // 4 * 4, ready[i]) for i in range(len(offsets)))
    # walk the buffer from its END (last layers live there, up to local re-ordering) and close a bucket whenever
    # enough bytes have gathered; the bucket's event is the earliest layer it contains
    buckets, hi, ev, acc = [], total, None, 0
    for lo_p, hi_p, r in reversed(spans):
        ev = r if ev is None else min(ev, r)
        acc = hi - lo_p
        if acc * 4 >= bucket_bytes:
            buckets.append((lo_p, hi, ev))
            hi, ev, acc = lo_p, None, 0
    if hi > 0:
        if ev is None:                                    # only padding left: glue it to the last bucket
            lo_b, hi_b, ev_b = buckets.pop()
            buckets.append((0, hi_b, ev_b))
        else:
            buckets.append((0, hi, ev))
    if max_buckets > 0:
        # Merge in BUFFER order (the buckets tile [0, total): neighbours in the sorted list are adjacent slices by
        # construction), not in launch order: with a locally re-ordered layout two buckets that are neighbours in launch
        # order need not touch, and a min..max span over the gap would overlap a third bucket (its gradients would be
        # all-reduced twice).  For a layout whose events fall monotonically along the buffer both orders coincide.
        def glue(a, b):                                   # a directly below b in the buffer -> one bucket (the later event)
            if a[1] != b[0]:
                raise RuntimeError(f"gradient buckets to merge are not adjacent slices: {a} / {b}")
            return (a[0], b[1], min(a[2], b[2]))
        buckets.sort(key=lambda b: b[0])
        if len(buckets) > 1 and (buckets[0][1] - buckets[0][0]) * 4 < bucket_bytes:
            buckets[0:2] = [glue(buckets[0], buckets[1])]    # a small tail (front of the buffer = end of the backward pass)
        while len(buckets) > max_buckets:                    # merge from the END of the buffer = the first to be launched
            buckets[-2:] = [glue(buckets[-2], buckets[-1])]
    # launch order = by event, descending; a later-closing bucket must never wait on an earlier event than it reports
    buckets.sort(key=lambda b: (-b[2], -b[0]))
    return buckets


class GradReducer:
    def __init__(self, process_group=None, bucket_bytes: int = None, max_buckets: int = None):
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("GradReducer needs an initialised torch.distributed process group "
                               "(backend 'nccl' = RCCL on the MI355X node, 'gloo' for CPU tests)")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self._pending: Dict[int, list] = {}          # id(opt) -> [Work, ...]
        self._buckets: Dict[int, List[Tuple[int, int, int]]] = {}
        self.bytes_reduced = 0
        self.collectives = 0
        self.stat_collectives = 0
        # RCCL's collectives can be recorded into a hipGraph (PyTorch's NCCL work objects fork / join the capturing stream
        # through events: tools/rccl_capture_probe.py); gloo does host-side work and cannot.  VAEGAN_DDP_CAPTURE=0 forces
        # the segmented form (graph cut at every bucket launch) also over RCCL.
        import os
        try:
            backend = dist.get_backend(process_group)
        except Exception:
            backend = ""
        self.capturable = backend == "nccl" and os.environ.get("VAEGAN_DDP_CAPTURE", "1") != "0"
        # per optimizer; in the segmented form every bucket is one hipGraph cut (0: no bound)
        self.bucket_bytes = int(bucket_bytes if bucket_bytes is not None else
                                (INLINE_BUCKET_BYTES if self.capturable else DEFAULT_BUCKET_BYTES))
        self.max_buckets = int(max_buckets if max_buckets is not None else
                               (INLINE_MAX_BUCKETS if self.capturable else DEFAULT_MAX_BUCKETS))

    def attach(self, *optimizers) -> None:
        """Fold the 1/world_size average into each optimizer's fused step."""
        for o in optimizers:
            o.grad_scale = 1.0 / self.world

    # ---- bucketed, overlapped reduction -------------------------------------------------------------------------
    def plan(self, opt, ready: Sequence[int]) -> List[Tuple[int, int, int]]:
        """ready[i] = backward event (stage index, counting down) that completes opt.params[i]'s gradient."""
        b = plan_buckets(opt.offsets, [p.numel() for p in opt.params], ready, opt.flat_g.numel(), self.bucket_bytes,
                         self.max_buckets)
        self._buckets[id(opt)] = b
        return b

    def buckets(self, opt) -> List[Tuple[int, int, int]]:
        return self._buckets.get(id(opt)) or [(0, opt.flat_g.numel(), 0)]

    def launch_bucket(self, opt, k: int) -> None:
        lo, hi, _ = self.buckets(opt)[k]
        w = dist.all_reduce(opt.flat_g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.setdefault(id(opt), []).append(w)
        self.bytes_reduced += (hi - lo) * 4
        self.collectives += 1

    def launch_ready(self, opt, event: int) -> int:
        """Launch every bucket of `opt` that event `event` completes.  Returns how many were launched."""
        n = 0
        for k, (_, _, ev) in enumerate(self.buckets(opt)):
            if ev == event:
                self.launch_bucket(opt, k)
                n += 1
        return n

    # ---- whole-buffer forms (one collective per network) ----------------------------------------------------------
    def reduce(self, opt) -> None:
        dist.all_reduce(opt.flat_g, op=dist.ReduceOp.SUM, group=self.group)
        self.bytes_reduced += opt.flat_g.numel() * 4
        self.collectives += 1

    def reduce_async(self, opt) -> None:
        w = dist.all_reduce(opt.flat_g, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.setdefault(id(opt), []).append(w)
        self.bytes_reduced += opt.flat_g.numel() * 4
        self.collectives += 1

    def wait(self, opt) -> None:
        for w in self._pending.pop(id(opt), []):
            w.wait()

    def drain(self) -> None:
        """Wait for every collective this reducer launched (before a hipGraph capture: trainer.py)."""
        for key in list(self._pending):
            for w in self._pending.pop(key):
                w.wait()

    def outstanding(self) -> int:
        return sum(len(v) for v in self._pending.values())

    def forget_pending(self) -> None:
        """Drop work handles created inside a hipGraph capture that was abandoned (they belong to no executed launch)."""
        self._pending.clear()

    def all_reduce_sum(self, t: torch.Tensor) -> None:
        """In-place SUM over ranks of a small statistics tensor (synchronised BatchNorm)."""
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        self.stat_collectives += 1

    def broadcast_parameters(self, *optimizers, src: int = 0, modules=()) -> None:
        """Make every replica start from rank `src`'s weights (one broadcast per flat buffer) and, for `modules`,
        from its BatchNorm buffers (running_mean / running_var / num_batches_tracked: a checkpoint loaded on one
        rank only would otherwise leave them inconsistent).  The packed GEMM operands derived from the parameters
        are invalidated: the broadcast writes the flat buffer behind the parameter views' backs."""
        from .engine import bump_weights_epoch
        for o in optimizers:
            dist.broadcast(o.flat_p, src=src, group=self.group)
            bump_weights_epoch(o.params)
        for m in modules:
            eng = getattr(m, "_engine", None)
            if eng is not None:
                eng.flush_bn_ticks()
                eng.invalidate()
            for b in m.buffers():
                dist.broadcast(b, src=src, group=self.group)

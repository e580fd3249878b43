"""hipGraph replay of a REFERENCE-SHAPED training step.

The three-line edit of vaegan_code.py (import the engine's modules and Adam) runs the reference's loop body
(vaegan_code.py:65-135: module calls, torch ops between them, ``loss.backward()``, ``optimizer.step()``) with autograd
driving the HIP kernel chains -- about 400 launches per iteration, each marshalled from Python, which makes that loop
host-bound (7.5 ms per iteration at S=64, B=128 against 2.8 ms of kernel time).  ``graphed`` removes the host from the
loop without touching its shape: the step function is captured ONCE into a hipGraph -- forward passes, the autograd
backward passes, ``zero_grad`` / ``step`` of the engine's Adam, the ``torch.randn_like`` draws (torch's graph-safe Philox
offsets) -- and replayed.

    def step(real_images):                        # the body of vaegan_code.py:65-135, verbatim, minus the .item() calls
        ...
        return recon_loss, kl_loss, g_loss_adv, d_loss
    step = vaegan_amd.graphed(step, modules=(encoder, decoder, discriminator), optimizers=(opt_E, opt_Dec, opt_Dis))
    for real_images in loader:
        losses = step(real_images, epoch=epoch)   # device tensors (static buffers: the next call overwrites them)

Rules of a captured step (those of any CUDA/HIP graph): no host synchronisation inside (``.item()``, ``print(tensor)``,
``.cpu()``): return the tensors and read them outside; tensor arguments must keep their shapes (a new shape -- the
ragged last batch -- is captured as a graph of its own); every other argument is a plain hashable scalar and part of
the capture key (``epoch`` changes the KL weight: a new value re-captures).  The first ``warmup`` calls of a signature
run eagerly (they size workspaces and the allocator's pools), the next call captures and replays, later calls replay.
Every call performs exactly one iteration.

Memory: all graphs of one GraphedStep are captured into ONE shared memory pool and at most ``max_graphs`` (default 2:
the full and the ragged last batch) are kept, least recently used first out -- ``step(x, epoch=epoch)`` over 200 epochs
holds two graphs, not 200 private pools.  ``scalar_key`` maps the keyword scalars to what the captured launches really
depend on, so that equal derived values share a graph: the reference's loop only uses ``epoch`` through the KL warm-up
weight ``min(1, epoch / 50)`` (vaegan_code.py:117), hence

    step = vaegan_amd.graphed(step, ..., scalar_key=lambda epoch: min(1.0, epoch / 50))

captures once per warm-up value and never again from epoch 50 on.
"""
from collections import OrderedDict
from typing import Callable, Optional, Sequence

import torch

from .engine import no_gc_while_capturing


def _flatten(out):
    if isinstance(out, torch.Tensor):
        return [out]
    if isinstance(out, (list, tuple)):
        r = []
        for o in out:
            r += _flatten(o)
        return r
    if isinstance(out, dict):
        r = []
        for o in out.values():
            r += _flatten(o)
        return r
    return []


class GraphedStep:
    def __init__(self, fn, modules: Sequence = (), optimizers: Sequence = (), warmup: int = 2, max_graphs: int = 2,
                 scalar_key: Optional[Callable] = None):
        self.fn, self.modules, self.optimizers, self.warmup = fn, tuple(modules), tuple(optimizers), max(1, int(warmup))
        self.max_graphs = max(1, int(max_graphs))
        self.scalar_key = scalar_key
        # key -> (graph, static inputs, outputs, bn-tick deltas, step-count deltas); least recently used first
        self._graphs = OrderedDict()
        self._seen = OrderedDict()  # key -> eager calls so far (bounded like _graphs)
        self._pool = None           # one memory pool for every capture of this step (created with the first one)

    def _engines(self):
        return [m._engine for m in self.modules if getattr(m, "_engine", None) is not None]

    def __call__(self, *tensors, **scalars):
        for t in tensors:
            if not isinstance(t, torch.Tensor) or not t.is_cuda:
                raise RuntimeError("graphed step: positional arguments are device tensors ('cuda'); pass scalars by keyword")
        skey = tuple(sorted(scalars.items())) if self.scalar_key is None else self.scalar_key(**scalars)
        key = (tuple((tuple(t.shape), t.dtype) for t in tensors), skey,
               tuple(m.training for m in self.modules),
               tuple((o.lr, o.betas, o.eps, o.grad_scale) for o in self.optimizers if hasattr(o, "grad_scale")))
        hit = self._graphs.get(key)
        if hit is not None:
            self._graphs.move_to_end(key)
            graph, sin, out, dticks, dsteps = hit
            for s, t in zip(sin, tensors):
                if s.data_ptr() != t.data_ptr():
                    s.copy_(t)
            graph.replay()
            self._bump(dticks, dsteps)
            return out
        n = self._seen.get(key, 0)
        if n < self.warmup:
            self._seen[key] = n + 1
            self._seen.move_to_end(key)
            while len(self._seen) > 8 * self.max_graphs:
                self._seen.popitem(last=False)
            return self.fn(*tensors, **scalars)
        # make room first: an evicted graph hands its share of the pool back before the new capture allocates
        while len(self._graphs) >= self.max_graphs:
            self._graphs.popitem(last=False)
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        # ---- capture ----
        sin = [t.clone() for t in tensors]
        engines = self._engines()
        for e in engines:
            e.invalidate()                              # the captured sequence must contain the operand re-packs
        torch.cuda.synchronize()
        ticks = [e.pending_bn_ticks for e in engines]
        steps = [o.steps for o in self.optimizers]
        graph = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream(device=sin[0].device if sin else None)
        cap.wait_stream(torch.cuda.current_stream())
        try:
            with no_gc_while_capturing(), torch.cuda.stream(cap):
                # global capture mode (the default): the autograd engine runs the backward nodes on its own device thread,
                # which launches into this stream -- their work is captured with the rest
                graph.capture_begin(pool=self._pool)
                try:
                    out = self.fn(*sin, **scalars)
                finally:
                    graph.capture_end()
        except BaseException:
            # nothing was executed: put the host-side mirrors back and let the caller run eagerly / try again
            for e, t in zip(engines, ticks):
                e.pending_bn_ticks = t
                e.invalidate()
            for o, s in zip(self.optimizers, steps):
                o.steps = s
            raise
        torch.cuda.current_stream().wait_stream(cap)
        if any(not t.is_cuda for t in _flatten(out)):
            raise RuntimeError("graphed step: the step function must return device tensors")
        # capture only records: undo the host-side counter changes it made, then replay for real
        dticks = [e.pending_bn_ticks - t for e, t in zip(engines, ticks)]
        dsteps = [o.steps - s for o, s in zip(self.optimizers, steps)]
        for e, t in zip(engines, ticks):
            e.pending_bn_ticks = t
        for o, s in zip(self.optimizers, steps):
            o.steps = s
        self._graphs[key] = (graph, sin, out, dticks, dsteps)
        graph.replay()
        self._bump(dticks, dsteps)
        return out

    def _bump(self, dticks, dsteps) -> None:
        for e, d in zip(self._engines(), dticks):
            e.pending_bn_ticks += d
        for o, d in zip(self.optimizers, dsteps):
            o.steps += d


def graphed(fn, modules: Sequence = (), optimizers: Sequence = (), warmup: int = 2, max_graphs: int = 2,
            scalar_key: Optional[Callable] = None) -> GraphedStep:
    """Wrap a reference-shaped training step for hipGraph replay (see the module docstring).
    modules: the engine networks the step calls (their BatchNorm forward counters and packed operands are host-side
    mirrors the replay keeps in step); optimizers: the vaegan_amd.Adam instances it steps; max_graphs: captured graphs
    kept (least recently used evicted); scalar_key(**scalars) -> hashable: what of the keyword scalars the captured
    launches depend on (default: the scalars themselves)."""
    return GraphedStep(fn, modules, optimizers, warmup, max_graphs, scalar_key)

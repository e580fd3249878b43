"""The VAE-GAN training iteration of the reference (vaegan_code.py:65-135) driven directly on the
HIP kernel chains -- no autograd graph, gradients written straight into the optimizers' flat
buffers.  Same op order, loss weights, zero_grad/step order and BatchNorm modes as the reference
(SURVEY.md A13); the three ``randn_like`` draws (vaegan_code.py:77,91,92) can be injected for
parity runs or are drawn on the device.

    trainer = VAEGANTrainer(encoder, decoder, discriminator, opt_E, opt_Dec, opt_Dis)
    losses = trainer.train_step(real_images, epoch)          # device tensor, no host sync
"""
import os
from typing import Dict, Optional

import torch

from . import geometry as G
from . import ops
from .engine import GradSink, no_gc_while_capturing

LOSS_NAMES = ("recon_loss", "kl_loss", "g_loss_adv", "d_loss_1", "d_loss_2")


class _SyncBNHandoff:
    """What the engines see as `bn_sync` in SyncBN mode: the statistics all-reduce of every BatchNorm goes through the
    trainer's graph cut, so that under hipGraph capture the collective stays OUTSIDE the captured segments (it is
    re-issued between segment replays on the same static f64 buffer) and runs immediately in eager mode."""

    def __init__(self, trainer):
        self.trainer = trainer

    @property
    def world(self):
        return self.trainer.reducer.world

    def all_reduce_sum(self, t):
        self.trainer._cut(lambda: self.trainer.reducer.all_reduce_sum(t))


class VAEGANTrainer:
    def __init__(self, encoder, decoder, discriminator, opt_E, opt_Dec, opt_Dis, alpha_kl: float = 0.1,
                 alpha_adv: float = 0.1, noise_sigma: float = 0.05, real_label: float = 0.9, fake_label: float = 0.1,
                 d_iters: int = 2, elide_dead_grads: bool = False, reducer=None, group_d_passes: bool = True,
                 sync_bn: bool = False):
        self.E, self.G, self.D = encoder, decoder, discriminator
        self.opt_E, self.opt_G, self.opt_D = opt_E, opt_Dec, opt_Dis
        self.alpha_kl, self.alpha_adv, self.sigma = alpha_kl, alpha_adv, noise_sigma          # :49-50, :91-92
        self.real_label, self.fake_label, self.d_iters = real_label, fake_label, d_iters      # :88-89, :95
        # The generator-loss pass through D (vaegan_code.py:110,133) also produces D weight gradients that the
        # next opt_Dis.zero_grad() discards unread.  False = compute them anyway (what the reference executes).
        self.elide_dead_grads = elide_dead_grads
        self.group_d_passes = group_d_passes       # run a D iteration's real+fake passes as one 2B-row launch chain
        # BCE + its gradient + the sigmoid / head backward as ONE launch per Discriminator pass (ops.head_backward;
        # bit-identical to the three separate launches, which False selects)
        self.fuse_head_backward = True
        # noise counter + the three optimizers' step counters / bias corrections in ONE launch at the top of the iteration
        # (ops.step_prologue) instead of one per optimizer step; False: every step() prepares itself
        self.fuse_step_prologue = os.environ.get("VG_STEP_PROLOGUE", "1") != "0"
        # round 4: four pairs of small launches merged (the Encoder's input conversion + the noisy real batch; the MSE's
        # final sum + the KL term; the Encoder's + the Generator's Adam step; the loss slots' memset + the prologue) --
        # bit-identical results; False restores the separate launches (A/B and test switch)
        self.merge_small_launches = os.environ.get("VG_MERGE_SMALL", "1") != "0"
        self.reducer = reducer
        # sync_bn: BatchNorm statistics over the global batch of all ranks (ddp.py) -- an N-rank step then equals
        # the reference's single-process step on the concatenated batch.  Off: per-replica statistics.
        self.sync_bn = bool(sync_bn)
        if self.sync_bn and reducer is None:
            raise ValueError("sync_bn=True needs a ddp.GradReducer (reducer=...)")
        for net in (encoder, decoder, discriminator):
            net._engine.bn_sync = _SyncBNHandoff(self) if self.sync_bn else None
        dts = {encoder._dt, decoder._dt, discriminator._dt}
        if len(dts) != 1:
            raise ValueError("encoder / decoder / discriminator must share one engine dtype")
        self.dt = dts.pop()
        self.latent = encoder.latent_dim
        self.losses = None
        self.noise = None               # ops.NoiseStream for the in-kernel randn_like draws (created on first use)
        self._noise_pinned_seed = None  # torch device seed at the time a checkpoint's noise stream was restored
        self._bucket_plans = {}
        self._graph = None              # (key, [hipGraph segments], [collectives between them], static in, static out)
        self._warm_key = None
        self._cut_hook = None           # set while capturing: splits the iteration into graph segments
        self._inline_failed = False     # capturing the collectives inside the graph failed once: use the segmented form

    def train(self):
        self.E.train(), self.G.train(), self.D.train()                                         # :56-58

    def _noise_stream(self, dev):
        """The generator behind the non-injected draws: torch's device seed (utils.configure_seed /
        torch.cuda.manual_seed) keys it, as it keys torch.randn_like in the reference; re-seeding torch starts a new
        stream (and invalidates a captured graph, which holds the old state buffer)."""
        seed = torch.cuda.initial_seed()
        if self.noise is not None and self._noise_pinned_seed == seed:
            return self.noise           # restored from a checkpoint: pinned until torch is explicitly re-seeded
        self._noise_pinned_seed = None
        if self.noise is None or self.noise.seed != seed:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("noise stream (re)seeded during graph capture")
            self.noise = ops.NoiseStream(dev, seed)
            self._graph = None
        return self.noise

    def _capture_key(self, real, epoch, inject):
        """Everything a captured graph freezes: shapes, every scalar kernel argument (loss weights, labels, noise
        sigma, Adam hyper-parameters, the data-parallel gradient scale), the schedule switches and the buffers the
        launches point at.  A change in any of them re-captures instead of silently replaying stale values."""
        opts = tuple((o.lr, o.betas, o.eps, o.grad_scale, o.flat_p.data_ptr()) for o in (self.opt_E, self.opt_G, self.opt_D))
        return (tuple(real.shape), float(self.alpha_kl * min(1.0, epoch / 50)), inject, self.E.training, self.G.training,
                self.D.training, self.alpha_adv, self.sigma, self.real_label, self.fake_label, self.d_iters,
                self.elide_dead_grads, self.group_d_passes, self.fuse_head_backward, self.fuse_step_prologue,
                self.merge_small_launches,
                id(self.reducer), self.sync_bn, opts,
                None if self.noise is None or inject else self.noise.state.data_ptr())

    # ---- data-parallel gradient hand-off (ddp.GradReducer) -------------------------------------------------------
    def _buckets_for(self, opt, net):
        """Plan `opt`'s gradient buckets once: parameter -> stage whose backward completes its gradient."""
        key = id(opt)
        if key not in self._bucket_plans:
            stage_of = net._engine.param_stage()
            ready = [stage_of.get(id(p), 0) for p in opt.params]
            plan = getattr(self.reducer, "plan", None)
            self._bucket_plans[key] = plan(opt, ready) if plan is not None else None
        return self._bucket_plans[key]

    def _grad_hook(self, opt, net):
        """on_grads callback for engine.backward: after stage i's weight gradients are enqueued, launch (behind a
        graph cut) the all-reduce of every bucket that stage completes.  Stage 0 is left to _finish_reduce, which
        launches the last bucket and waits in ONE cut (no empty graph segment between two cuts)."""
        if self.reducer is None or self._buckets_for(opt, net) is None:
            return None
        events = {ev for _, _, ev in self._bucket_plans[id(opt)] if ev > 0}

        def hook(i):
            if i in events:
                self._cut(lambda: self.reducer.launch_ready(opt, i))
        return hook

    def _finish_reduce(self, opt, net, wait=True, also_wait=()):
        """End of `net`'s backward: launch what is left of opt's buckets, optionally wait for all of them."""
        if self.reducer is None:
            return
        bucketed = self._buckets_for(opt, net) is not None

        def fn():
            if bucketed:
                self.reducer.launch_ready(opt, 0)
            elif wait:
                self.reducer.reduce(opt)
            else:
                self.reducer.reduce_async(opt)
            if wait:
                self.reducer.wait(opt)
            for o in also_wait:
                self.reducer.wait(o)
        self._cut(fn)

    def _cut(self, collective) -> None:
        """A point where the iteration hands gradients to the reducer.  Eager: run the collective now.  While
        capturing: close the current hipGraph segment, remember the collective, open the next segment -- RCCL
        calls stay OUTSIDE the captured graphs and are launched between segment replays."""
        if self._cut_hook is None:
            collective()
        else:
            self._cut_hook(collective)

    def train_step(self, real: torch.Tensor, epoch: int, eps_z: Optional[torch.Tensor] = None,
                   eps_real: Optional[torch.Tensor] = None, eps_recon: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One iteration.  Returns a device tensor [recon_loss, kl_loss, g_loss_adv, d_loss_1, d_loss_2]."""
        steps = [o.steps for o in (self.opt_E, self.opt_G, self.opt_D)]
        try:
            return self._train_step(real, epoch, eps_z, eps_real, eps_recon)
        except BaseException:
            # The step prologue advances the DEVICE step counters / bias corrections of all three optimizers at the top of
            # the iteration, the host mirrors (opt.steps) move with the updates at its end.  An eager iteration that dies
            # in between (OOM, a bad shape in a later pass) must not leave the two apart: a retried iteration would apply
            # t + 2 and state_dict() would save a step count that disagrees with the bias correction in use.  (A capture
            # executes nothing: train_step_graphed restores the host mirrors itself.)
            if not torch.cuda.is_current_stream_capturing():
                for o, st in zip((self.opt_E, self.opt_G, self.opt_D), steps):
                    if o.steps == st:
                        try:
                            o.state_dev[0:1].fill_(float(st))
                        except Exception:
                            pass
            raise

    def _train_step(self, real, epoch, eps_z, eps_real, eps_recon) -> torch.Tensor:
        if not real.is_cuda:
            raise RuntimeError("train_step needs the batch on the MI355X ('cuda'); there is no CPU path")
        E, Gn, D, dt = self.E, self.G, self.D, self.dt
        B, dev = real.shape[0], real.device
        L = self.latent
        real = real.contiguous()
        noise = None
        if eps_z is None or eps_real is None or eps_recon is None:
            # the three randn_like draws (:77, :91, :92) are generated inside the kernels that consume them
            # (Philox keyed by torch's device seed; the iteration counter is bumped on the device, also under replay)
            noise = self._noise_stream(dev)
            eps_z = noise.draw(0) if eps_z is None else eps_z
            eps_real = noise.draw(1) if eps_real is None else eps_real
            eps_recon = noise.draw(2) if eps_recon is None else eps_recon
        # one single-thread launch for everything that only counts: the noise iteration and the step counters / bias
        # corrections of the three Adam steps below (the Discriminator's second update prepares itself)
        prep = self.fuse_step_prologue and self.d_iters >= 1
        # slots 0..4 are written (not accumulated) below when d_iters >= 2; with d_iters = 1 slot 4 (d_loss_2) is never
        # written and with d_iters > 2 it holds the LAST iteration's loss -- zeroed so that it reads 0, not stale memory.
        # With the prologue launch the zeroing rides on it (no memset node in the captured iteration).
        losses = torch.empty(8, dtype=torch.float32, device=dev) if prep else ops.zeros_f32(8, dev)
        if prep:
            ops.step_prologue(noise, [self.opt_D, self.opt_G, self.opt_E], zero=losses)
        elif noise is not None:
            noise.advance()
        sink = GradSink(direct=True)

        # ---- Encode / reparameterise / decode (:74-83) ----
        # instance noise, drawn once per step (:91-92), produced directly in the layout D reads.  Both noisy batches live
        # in one [2B] buffer so that a Discriminator iteration can run real+fake as ONE grouped pass (per-group BatchNorm
        # statistics, running stats updated real-then-fake as the reference's two calls do).  The Encoder's input and the
        # Discriminator's noisy real batch are the same images: one pass over `real` writes both (round 4).
        CP = G.padc(D.nc, dt)
        both = ops.empty_act((2 * B, real.shape[2], real.shape[3], CP), dt, dev)
        pair = ops.nchw_to_nhwc_pair(real, CP, dt, eps_real, self.sigma, both[:B]) \
            if (self.merge_small_launches and G.padc(real.shape[1], dt) == CP) else None
        mulv, ctxE = E.engine_forward(real, x_nhwc=None if pair is None else pair[0])
        ZP = G.padc(Gn.nz, dt)
        z, lvc = ops.reparam_forward(mulv, eps_z, L, ZP, dt)
        real_noisy = both[:B] if pair is not None else \
            ops.nchw_to_nhwc(real, CP, dt, eps=eps_real, sigma=self.sigma, out=both[:B])
        recon_noisy = both[B:]
        if Gn.nc == D.nc and G.padc(Gn.nc, dt) == CP and Gn.fused_tail(B):
            # the last ConvTranspose2d's kernel applies Tanh and writes the NCHW image AND image + sigma*eps in D's layout
            recon, ctxG = Gn.engine_forward(z, B, tail=dict(noise=eps_recon, sigma=self.sigma, out_noisy=recon_noisy))
        else:
            pre, ctxG = Gn.engine_forward(z, B)
            if Gn.nc == D.nc and pre.shape[-1] == CP:
                recon = ops.nhwc_tanh_to_nchw_noisy(pre, Gn.nc, eps_recon, self.sigma, recon_noisy, dt)   # :83 and :92 in one pass
            else:
                recon = ops.nhwc_to_nchw(pre, Gn.nc, dt, apply_tanh=True)
                ops.nchw_to_nhwc(recon, CP, dt, eps=eps_recon, sigma=self.sigma, out=recon_noisy)
        grouped = self.group_d_passes and D._engine.can_group(B, 2, both)
        fused_head = self.fuse_head_backward and 2 * B <= ops.HEAD_BWD_MAXROWS and D._engine.stages[-1].kind == "head"

        # ---- Discriminator updates (:95-105) ----
        for it in range(self.d_iters):
            slot = losses[3 + min(it, 1):4 + min(it, 1)]
            self.opt_D.zero_grad(memset=False)
            if grouped:
                p_both, c_both = D.engine_forward(both, B, groups=2)                           # .detach(): no dx below
                if fused_head:
                    # :98-104: the two BCE terms, their gradient and the head's backward in one launch
                    D._engine.backward(c_both, None, False, sink, on_grads=self._grad_hook(self.opt_D, D),
                                       head_loss=(self.real_label, self.fake_label, 2, 1.0, slot, False))
                else:
                    dp = torch.empty_like(p_both)
                    ops.bce_pair_forward_backward(p_both, self.real_label, self.fake_label, 1.0, slot, dp)   # :98-103
                    D._engine.backward(c_both, dp, False, sink, on_grads=self._grad_hook(self.opt_D, D))
            else:
                p_real, c_real = D.engine_forward(real_noisy, B)
                p_fake, c_fake = D.engine_forward(recon_noisy, B)
                dp_real = ops.bce_forward_backward(p_real, self.real_label, 1.0, slot, False, True)
                dp_fake = ops.bce_forward_backward(p_fake, self.fake_label, 1.0, slot, True, True)
                D._engine.backward(c_real, dp_real, False, sink)
                D._engine.backward(c_fake, dp_fake, False, sink, on_grads=self._grad_hook(self.opt_D, D))   # the accumulating pass
            self._finish_reduce(self.opt_D, D)            # D's buckets overlap the rest of its own backward
            self.opt_D.step(prepared=prep and it == 0)

        # ---- Generator + VAE loss (:110-117) ----
        p_adv, c_adv = D.engine_forward(recon_noisy, B)
        if self.merge_small_launches:
            # :113-114: the MSE's final sum rides on the KL launch (same arithmetic as its own one-wave launch)
            d_recon, mse_tail = ops.mse_forward_backward(recon, real, 1.0, losses[0:1], True, defer_final=True)
            ops.kl_forward(mulv, lvc, L, float(B), dt, out=losses[1:2], mse=mse_tail)
        else:
            d_recon = ops.mse_forward_backward(recon, real, 1.0, losses[0:1], True)           # :113
            ops.kl_forward(mulv, lvc, L, float(B), dt, out=losses[1:2])                       # :114
        dp_adv = None
        if not fused_head:
            dp_adv = ops.bce_forward_backward(p_adv, self.real_label, self.alpha_adv, losses[2:3], False, True)  # :115

        # ---- backward of total = recon + a_kl*min(1,epoch/50)*kl + a_adv*adv, then E and G steps (:131-135) ----
        self.opt_E.zero_grad(memset=False)
        self.opt_G.zero_grad(memset=False)
        d_noisy = D._engine.backward(c_adv, dp_adv, True, sink, param_grads=not self.elide_dead_grads,
                                     head_loss=(self.real_label, 0.0, 1, self.alpha_adv, losses[2:3], False) if fused_head else None)
        # d total / d recon = d MSE + d adv through the instance-noise add (:92), then through tanh: one pass
        d_pre = ops.nchw_grad_add_to_nhwc(d_recon, d_noisy, recon, G.padc(Gn.nc, dt), dt)
        dz = Gn._engine.backward(ctxG, d_pre, True, sink, on_grads=self._grad_hook(self.opt_G, Gn))
        self._finish_reduce(self.opt_G, Gn, wait=False)       # G's last bucket overlaps the encoder's backward
        kl_w = self.alpha_kl * min(1.0, epoch / 50)                                            # :117
        dmulv = ops.reparam_kl_backward(mulv, lvc, eps_z, dz, kl_w / B, L, dt)
        E._engine.backward(ctxE, dmulv.view(B, 1, 1, -1), False, sink, on_grads=self._grad_hook(self.opt_E, E))
        self._finish_reduce(self.opt_E, E, also_wait=(self.opt_G,))
        if prep and self.merge_small_launches:
            self.opt_E.step_pair(self.opt_G)                  # :134-135 in one launch (both prepared by the prologue)
        else:
            self.opt_E.step(prepared=prep)
            self.opt_G.step(prepared=prep)
        self.losses = losses
        return losses

    # ---- hipGraph replay of the whole iteration -------------------------------------------------------------
    def graph_input(self):
        """The static image buffer the captured iteration reads ([B, C, H, W] f32), or None before a capture.  A loader
        that assembles its batches straight into it (data.DeviceLoader.bind_output) hands them to train_step_graphed
        without the device-to-device copy; pass the SAME tensor as `real`."""
        return None if self._graph is None else self._graph[3][0]

    def train_step_graphed(self, real: torch.Tensor, epoch: int, eps_z: Optional[torch.Tensor] = None,
                           eps_real: Optional[torch.Tensor] = None,
                           eps_recon: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Same iteration as train_step, replayed from captured hipGraphs (~220 kernel launches become one graph
        launch per segment).  The returned tensor is the graph's STATIC output buffer: the next call overwrites it
        (clone it to keep a history of losses).  Every call performs exactly one training iteration: the first call with a new
        (shape, KL weight, noise mode) runs eagerly (it also sizes the workspaces), the second captures and
        replays, later calls replay.  Inputs are copied into static buffers; noise is either injected on every
        call or drawn on the device inside the graph (torch's graph-safe Philox state).
        With a gradient reducer the iteration is captured as SEGMENTS that share one memory pool, cut at the points
        where gradients are handed to the reducer (one cut per gradient bucket, ddp.py) and, in SyncBN mode, at every
        BatchNorm's statistics all-reduce; the collectives run eagerly between the segment replays."""
        inject = eps_z is not None
        if inject and (eps_real is None or eps_recon is None):
            raise ValueError("inject all three noise tensors or none")
        if not inject:
            self._noise_stream(real.device)         # exists (and is keyed on the current seed) before the key is formed
        key = self._capture_key(real, epoch, inject)
        if self._graph is not None and self._graph[0] == key:
            _, graphs, cuts, sin, sout = self._graph[:5]
            if real.data_ptr() != sin[0].data_ptr():    # a batch assembled in graph_input() needs no copy
                sin[0].copy_(real)
            if inject:
                sin[1].copy_(eps_z), sin[2].copy_(eps_real), sin[3].copy_(eps_recon)
            self._replay(graphs, cuts)
            self._advance_host_counters()
            self.losses = sout
            return sout
        if self._warm_key != key:
            self._warm_key = key
            self._graph = None
            return self.train_step(real, epoch, eps_z, eps_real, eps_recon)
        sin = [real.clone()] + ([eps_z.clone(), eps_real.clone(), eps_recon.clone()] if inject else [None] * 3)
        # Collectives INSIDE the graph (round 4): RCCL's all-reduces are capturable on this stack (PyTorch 2.10 / RCCL 2.26:
        # tools/rccl_capture_probe.py) -- the asynchronous bucket launches fork onto RCCL's stream and the waits join it
        # back, all as graph dependencies, so the iteration stays ONE graph and no hand-off costs a graph boundary (a cut
        # is ~29 us of idle GPU, 5 per iteration: DESIGN.md section 7).  Reducers that cannot be captured (gloo: host-side
        # work) keep the segmented form; if the inline capture fails on some stack the segmented one is tried next.
        inline = self.reducer is not None and bool(getattr(self.reducer, "capturable", False)) and not self._inline_failed
        try:
            graphs, cuts, sout, dcount = self._capture(sin, epoch, inline)
        except Exception as ex:                     # noqa: BLE001
            if not inline:
                raise
            import sys
            print(f"[vaegan_amd] capturing the gradient collectives inside the hipGraph failed ({ex!r}); "
                  f"falling back to graph segments cut at the collectives", file=sys.stderr)
            self._inline_failed = True
            graphs, cuts, sout, dcount = self._capture(sin, epoch, False)
        self._graph = (key, graphs, cuts, sin, sout, dcount)
        self._replay(graphs, cuts)
        self._advance_host_counters()
        self.losses = sout
        return sout

    def _capture(self, sin, epoch, inline):
        """Capture one iteration on the static inputs `sin`.  inline: the reducer's collectives are recorded into the graph
        (one segment); else the graph is cut at every hand-off to the reducer and the collectives run between the segment
        replays.  Returns (graphs, cuts, static output, per-iteration deltas of the reducer's counters).  Executes nothing;
        on failure the trainer is exactly where it was."""
        for eng in (self.E._engine, self.G._engine, self.D._engine):
            eng.invalidate()                       # the captured sequence must contain the operand re-packs
        # Nothing of torch.distributed may be in flight while a stream is capturing: c10d's watchdog thread polls
        # unfinished collectives with hipEventQuery, which is illegal next to a capture (the abort recorded in round 1).
        # Structural guard rather than luck: wait for every collective this trainer launched, drain the device, and
        # REFUSE to capture if the reducer still reports outstanding work.
        if self.reducer is not None and hasattr(self.reducer, "drain"):
            self.reducer.drain()
        torch.cuda.synchronize()
        if self.reducer is not None and getattr(self.reducer, "outstanding", lambda: 0)() != 0:
            raise RuntimeError("hipGraph capture refused: the gradient reducer still has collectives in flight")
        ticks = [m._engine.pending_bn_ticks for m in (self.E, self.G, self.D)]
        steps = [o.steps for o in (self.opt_E, self.opt_G, self.opt_D)]
        cnames = ("collectives", "bytes_reduced", "stat_collectives")
        counts = [getattr(self.reducer, n, 0) for n in cnames] if self.reducer is not None else None
        graphs, cuts = [], []
        pool = torch.cuda.graph_pool_handle()
        cap = torch.cuda.Stream(device=sin[0].device)
        cap.wait_stream(torch.cuda.current_stream())

        def begin():
            g = torch.cuda.CUDAGraph()
            # thread-local capture mode: torch.distributed's watchdog thread polls finished collectives with
            # hipEventQuery at its own pace; under the default (global) mode such a call from ANOTHER thread while
            # this one captures is an error that takes the process down (seen once in four runs with RCCL)
            g.capture_begin(pool=pool, capture_error_mode="thread_local")
            graphs.append(g)

        def cut(collective):
            if inline:
                collective()                       # recorded: a fork onto / a join from RCCL's stream inside the graph
                return
            graphs[-1].capture_end()               # close this segment, remember the collective, open the next
            cuts.append(collective)
            begin()

        def restore_host_counters():               # capture only records: undo the host-side counter changes it made
            for m, t in zip((self.E, self.G, self.D), ticks):
                m._engine.pending_bn_ticks = t
            for o, st in zip((self.opt_E, self.opt_G, self.opt_D), steps):
                o.steps = st
            deltas = None
            if counts is not None:
                deltas = [getattr(self.reducer, n, 0) - c for n, c in zip(cnames, counts)]
                for n, c in zip(cnames, counts):
                    if hasattr(self.reducer, n):
                        setattr(self.reducer, n, c)
            return deltas

        with no_gc_while_capturing(), torch.cuda.stream(cap):
            self._cut_hook = cut
            try:
                begin()
                sout = self.train_step(sin[0], epoch, sin[1], sin[2], sin[3])
                graphs[-1].capture_end()
            except BaseException:
                # leave no stream behind in capture mode and no half-built state: nothing was executed, so after
                # this the trainer is exactly where it was before the call and can run eagerly (or capture again)
                try:
                    graphs[-1].capture_end()
                except Exception:
                    pass
                restore_host_counters()
                if self.reducer is not None and hasattr(self.reducer, "forget_pending"):
                    self.reducer.forget_pending()
                self._graph, self._warm_key = None, None
                for eng in (self.E._engine, self.G._engine, self.D._engine):
                    eng.invalidate()
                raise
            finally:
                self._cut_hook = None
        torch.cuda.current_stream().wait_stream(cap)
        deltas = restore_host_counters()           # then replay for real
        dcount = dict(zip(cnames, deltas)) if (inline and deltas is not None) else None
        return graphs, cuts, sout, dcount

    @staticmethod
    def _replay(graphs, cuts) -> None:
        for i, g in enumerate(graphs):
            g.replay()
            if i < len(cuts):
                cuts[i]()                       # the collective that separates segment i from segment i + 1

    def _advance_host_counters(self) -> None:
        """What one iteration does to host-side mirrors: BatchNorm forward counts (E 1, G 1, D 2*d_iters+1) and
        optimizer step counts (the authoritative Adam step counter lives on the device); with collectives captured
        inside the graph also the reducer's statistics counters (a replay runs no Python of the reducer)."""
        dcount = self._graph[5] if (self._graph is not None and len(self._graph) > 5) else None
        if dcount and self.reducer is not None:
            for n, d in dcount.items():
                if hasattr(self.reducer, n):
                    setattr(self.reducer, n, getattr(self.reducer, n) + d)
        if self.E.training:
            self.E._engine.pending_bn_ticks += 1
            self.G._engine.pending_bn_ticks += 1
            self.D._engine.pending_bn_ticks += 2 * self.d_iters + 1
        self.opt_E.steps += 1
        self.opt_G.steps += 1
        self.opt_D.steps += self.d_iters

    # ---- checkpoint / resume ----------------------------------------------------------------------------------
    # The reference only ever writes `decoder.state_dict()` (vaegan_code.py:193; main_vae.py:204-205 also the
    # encoder) and reads such files back with torch.load(weights_only=True) + load_state_dict (main_vae.py:246-249,
    # :356-357).  Module state_dicts here have the reference's keys/shapes/dtypes, so those files
    # interchange both ways.  state_dict()/load_state_dict() below add what the reference lacks: all three networks
    # plus the three Adam states in one payload of plain tensors / dicts / numbers (weights_only-loadable), from
    # which training resumes bit-identically.
    def state_dict(self) -> Dict:
        return {"format": 1,
                "encoder": self.E.state_dict(), "decoder": self.G.state_dict(), "discriminator": self.D.state_dict(),
                "opt_E": self.opt_E.state_dict(), "opt_Dec": self.opt_G.state_dict(), "opt_Dis": self.opt_D.state_dict(),
                # generator state of the in-kernel randn_like draws: [seed, iteration counter] (None: never used)
                "noise": None if self.noise is None else self.noise.get_state().cpu()}

    def load_state_dict(self, sd: Dict) -> None:
        if sd.get("format") != 1:
            raise RuntimeError(f"unknown VAE-GAN checkpoint format {sd.get('format')!r}")
        self.E.load_state_dict(sd["encoder"]), self.G.load_state_dict(sd["decoder"])
        self.D.load_state_dict(sd["discriminator"])
        self.opt_E.load_state_dict(sd["opt_E"]), self.opt_G.load_state_dict(sd["opt_Dec"])
        self.opt_D.load_state_dict(sd["opt_Dis"])
        if sd.get("noise") is not None:
            st = sd["noise"]
            dev = next(self.E.parameters()).device
            if self.noise is None:
                self.noise = ops.NoiseStream(dev, int(st[0]))
                self._graph = None
            self.noise.seed = int(st[0])
            self.noise.set_state(st)
            # The restored stream (its seed AND iteration counter) stays in use although the resuming process's torch
            # seed differs from the saved one (or the saved seed was >= 2^63 and is stored masked): only an explicit
            # re-seed of torch AFTER this point (utils.configure_seed / torch.cuda.manual_seed) starts a new stream.
            self._noise_pinned_seed = torch.cuda.initial_seed()

    def save_checkpoint(self, path: str, **extra) -> None:
        """extra: plain numbers / strings / tensors stored next to the state (e.g. epoch=...)."""
        sd = self.state_dict()
        sd["extra"] = dict(extra)
        torch.save(sd, path)

    def load_checkpoint(self, path: str) -> Dict:
        """Loads with torch.load(weights_only=True) (nothing in the file is executed).  Returns the extras."""
        sd = torch.load(path, map_location=next(self.E.parameters()).device, weights_only=True)
        self.load_state_dict(sd)
        return sd.get("extra", {})

    def loss_dict(self, losses: Optional[torch.Tensor] = None, epoch: Optional[int] = None) -> Dict[str, float]:
        """Host copy of the last step's losses (one device sync, like the reference's .item() calls :125-127)."""
        v = (losses if losses is not None else self.losses)[:5].tolist()
        out = dict(zip(LOSS_NAMES, v))
        if epoch is not None:
            out["total"] = out["recon_loss"] + self.alpha_kl * min(1.0, epoch / 50) * out["kl_loss"] \
                + self.alpha_adv * out["g_loss_adv"]
        return out

"""The reference's three sibling training loops on the same HIP kernel chains as the VAE-GAN path
(SURVEY.md section 8(f), row F4):

  VAETrainer    main_vae.py:103-127   plain (denoising) VAE: ONE Adam over encoder + decoder, lr 1e-3
  DCGANTrainer  gan_code.py:194-219   DCGAN: BCE against hard labels 1 / 0, Adam betas (0.5, 0.999)
  WGANTrainer   gan_code.py:296-331   weight-clipped WGAN: 5 critic iterations per generator step

Same conventions as trainer.VAEGANTrainer: direct kernel chains (no autograd graph), gradients land in the
optimizers' flat buffers, every random draw of the reference loop can be injected (parity runs) or is drawn on the
device, losses come back as one device tensor (no host sync), ``step_graphed`` replays the whole iteration from a
captured hipGraph.  There is no CPU path.
"""
from typing import List, Optional, Sequence

import torch

from . import geometry as G
from . import ops
from .engine import GradSink, bump_weights_epoch, no_gc_while_capturing


class _Graphed:
    """Whole-iteration hipGraph replay for a trainer whose ``train_step(*tensors, **scalars)`` takes device tensors
    (static shapes) and plain scalars.  First call with a new signature runs eagerly (it also sizes the
    workspaces), the second captures and replays, later calls replay; every call is exactly one iteration."""

    _nets: Sequence = ()
    _opts: Sequence = ()

    def step_graphed(self, *tensors, **scalars):
        # Draws that are not injected come from the per-device default NoiseStream, whose STATE BUFFER the captured
        # vg_rng_advance / vg_randn launches point at.  utils.configure_seed() (ops.reset_noise) drops that stream: the
        # graph must then be re-captured against the new one -- the buffer's address is part of the key, and the
        # trainer holds a reference to the stream it captured so the old buffer cannot be recycled under a replay.
        noise = None
        if any(t is None for t in tensors):
            dev = next(t for t in tensors if t is not None).device
            noise = ops.default_noise(dev)
        key = (tuple(None if t is None else tuple(t.shape) for t in tensors), tuple(sorted(scalars.items())),
               tuple(n.training for n in self._nets), None if noise is None else noise.state.data_ptr())
        self._gnoise = noise
        st = getattr(self, "_gstate", None)
        if st is not None and st[0] == key:
            _, graph, sin, sout, dticks, dsteps = st[:6]
            for s, t in zip(sin, tensors):
                if s is not None:
                    s.copy_(t)
            graph.replay()
            self._bump(dticks, dsteps)
            return sout
        if getattr(self, "_gwarm", None) != key:
            self._gwarm, self._gstate = key, None
            return self.train_step(*tensors, **scalars)
        sin = [None if t is None else t.clone() for t in tensors]      # None: drawn on the device inside the graph
        for n in self._nets:
            n._engine.invalidate()                     # the captured sequence must contain the operand re-packs
        torch.cuda.synchronize()
        ticks = [n._engine.pending_bn_ticks for n in self._nets]
        steps = [o.steps for o in self._opts]
        graph = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream(device=sin[0].device)
        cap.wait_stream(torch.cuda.current_stream())
        with no_gc_while_capturing(), torch.cuda.stream(cap):
            graph.capture_begin(capture_error_mode="thread_local")   # see trainer.py: other threads (c10d watchdog) may poll events
            sout = self.train_step(*sin, **scalars)
            graph.capture_end()
        torch.cuda.current_stream().wait_stream(cap)
        # capture only records: undo the host-side counter changes it made, then replay for real
        dticks = [n._engine.pending_bn_ticks - t for n, t in zip(self._nets, ticks)]
        dsteps = [o.steps - s for o, s in zip(self._opts, steps)]
        for n, t in zip(self._nets, ticks):
            n._engine.pending_bn_ticks = t
        for o, s in zip(self._opts, steps):
            o.steps = s
        self._gstate = (key, graph, sin, sout, dticks, dsteps, noise)   # noise: keeps the captured state buffer alive
        graph.replay()
        self._bump(dticks, dsteps)
        return sout

    def _bump(self, dticks, dsteps) -> None:
        for n, d in zip(self._nets, dticks):
            n._engine.pending_bn_ticks += d
        for o, d in zip(self._opts, dsteps):
            o.steps += d


def _need_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what} needs the batch on the MI355X ('cuda'); there is no CPU path")


def _same_dtype(*nets) -> int:
    dts = {n._dt for n in nets}
    if len(dts) != 1:
        raise ValueError("the networks must share one engine dtype")
    return dts.pop()


# ---------------------------------------------------------------------------------------------------------------
class VAETrainer(_Graphed):
    """train_vae's loop body (main_vae.py:103-127).  ``optimizer`` is ONE Adam over
    ``list(encoder.parameters()) + list(decoder.parameters())`` (main_vae.py:84-87)."""
    LOSS_NAMES = ("recon_loss", "kl_loss", "total")

    def __init__(self, encoder, decoder, optimizer, noise_max_std: float = 0.5, kl_weight: float = 1e-5):
        self.E, self.G, self.opt = encoder, decoder, optimizer
        self.sigma, self.kl_weight = noise_max_std, kl_weight                       # :66, :121
        self.dt = _same_dtype(encoder, decoder)
        self._nets, self._opts = (encoder, decoder), (optimizer,)

    def train(self):
        self.E.train(), self.G.train()                                             # :98-99

    def train_step(self, img: torch.Tensor, eps_img: Optional[torch.Tensor] = None,
                   eps_z: Optional[torch.Tensor] = None, epoch: int = 0) -> torch.Tensor:
        """-> device tensor [recon_loss, kl_loss (sum, not /B), total]."""
        _need_cuda(img, "VAETrainer.train_step")
        E, Gn, dt = self.E, self.G, self.dt
        B, dev, L = img.shape[0], img.device, E.latent_dim
        img = img.contiguous()
        if eps_img is None or eps_z is None:
            # the reference's randn_like draws, generated in HIP (Philox keyed by torch's device seed; the one-thread
            # advance kernel is captured with the iteration, so graph replays draw fresh noise)
            noise = ops.default_noise(dev)
            noise.advance()
            if eps_img is None:
                eps_img = noise.randn(tuple(img.shape), 0)                         # :104
            if eps_z is None:
                eps_z = noise.randn((B, L), 1)                                     # :114
        losses = ops.zeros_f32(4, dev)
        sink = GradSink(direct=True)
        noisy_h, _ = ops.noisy_clamp_to_nhwc(img, eps_img, self.sigma, G.padc(img.shape[1], dt), dt)   # :104-105
        mulv, ctxE = E._engine.forward(noisy_h, B, E.training, True)               # :111
        mulv = mulv.view(B, -1)
        z, lvc = ops.reparam_forward(mulv, eps_z, L, G.padc(Gn.nz, dt), dt)        # :112-115
        pre, ctxG = Gn.engine_forward(z, B)                                        # :116
        recon = ops.nhwc_to_nchw(pre, Gn.nc, dt, apply_tanh=True)
        d_recon = ops.mse_forward_backward(recon, img, 1.0, losses[0:1], True)     # :119
        ops.kl_forward(mulv, lvc, L, 1.0, dt, out=losses[1:2])                     # :120
        w = min(epoch / 50, 1.0) * self.kl_weight                                  # :121
        torch.add(losses[0:1], losses[1:2], alpha=w, out=losses[2:3])
        self.opt.zero_grad(memset=False)                                           # :124
        d_pre = ops.nchw_grad_to_nhwc(d_recon, recon, G.padc(Gn.nc, dt), dt)
        dz = Gn._engine.backward(ctxG, d_pre, True, sink)
        dmulv = ops.reparam_kl_backward(mulv, lvc, eps_z, dz, w, L, dt)
        E._engine.backward(ctxE, dmulv.view(B, 1, 1, -1), False, sink)
        self.opt.step()                                                            # :126
        return losses


# ---------------------------------------------------------------------------------------------------------------
class _GANBase(_Graphed):
    def __init__(self, netG, netD, optimizerG, optimizerD, elide_dead_grads: bool = False,
                 group_d_passes: bool = True):
        self.G, self.D, self.opt_G, self.opt_D = netG, netD, optimizerG, optimizerD
        # The generator step's backward through D also produces D weight gradients that the next netD.zero_grad()
        # clears unread (module-level zero_grad, gan_code.py:195/:302).  False = compute them anyway, as the
        # reference does.
        self.elide_dead_grads = elide_dead_grads
        self.group_d_passes = group_d_passes
        self.dt = _same_dtype(netG, netD)
        self._nets, self._opts = (netG, netD), (optimizerG, optimizerD)

    def train(self):
        self.G.train(), self.D.train()

    def _gen(self, noise, keep):
        """noise [B,nz,1,1] f32 -> (tanh image NCHW f32, the same image in D's NHWC layout, ctx)."""
        dt, Gn = self.dt, self.G
        B = noise.shape[0]
        zh = ops.nchw_to_nhwc(noise.contiguous(), G.padc(Gn.nz, dt), dt)
        pre, ctx = Gn._engine.forward(zh, B, Gn.training, keep)
        return ops.nhwc_to_nchw(pre, Gn.nc, dt, apply_tanh=True), ctx

    def _gen_backward(self, ctxG, fake, d_fake_nhwc, sink):
        dt, Gn = self.dt, self.G
        d_img = ops.nhwc_to_nchw(d_fake_nhwc, Gn.nc, dt)
        d_pre = ops.nchw_grad_to_nhwc(d_img, fake, G.padc(Gn.nc, dt), dt)
        Gn._engine.backward(ctxG, d_pre, False, sink)


class DCGANTrainer(_GANBase):
    """train_gan's loop body (gan_code.py:194-219).  optimizerD / optimizerG: Adam(lr=2e-4, betas=(0.5, 0.999))
    (gan_code.py:179-180)."""
    LOSS_NAMES = ("errD_real", "errD_fake", "errG")

    def train_step(self, real: torch.Tensor, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        _need_cuda(real, "DCGANTrainer.train_step")
        D, dt = self.D, self.dt
        B, dev = real.shape[0], real.device
        if noise is None:
            ns = ops.default_noise(dev)
            ns.advance()
            noise = ns.randn((B, self.G.nz, 1, 1), 0)                              # :203
        losses = ops.zeros_f32(4, dev)
        sink = GradSink(direct=True)
        CP = G.padc(D.nc, dt)
        # D(real) -> G(noise) -> D(fake.detach()) in the reference; G's forward does not touch D, so the two D
        # passes run as ONE grouped 2B-row pass (per-group BatchNorm statistics, running stats real-then-fake)
        fake, ctxG = self._gen(noise, True)                                        # :204
        both = ops.empty_act((2 * B, real.shape[2], real.shape[3], CP), dt, dev)
        real_h = ops.nchw_to_nhwc(real.contiguous(), CP, dt, out=both[:B])
        fake_h = ops.nchw_to_nhwc(fake, CP, dt, out=both[B:])
        self.opt_D.zero_grad(memset=False)                                         # :195
        if self.group_d_passes and D._engine.can_group(B, 2, both):
            p_both, c_both = D.engine_forward(both, B, groups=2)
            dp = torch.empty_like(p_both)
            ops.bce_forward_backward(p_both[:B], 1.0, 1.0, losses[0:1], False, True, out=dp[:B])   # :199-200
            ops.bce_forward_backward(p_both[B:], 0.0, 1.0, losses[1:2], False, True, out=dp[B:])   # :206-207
            D._engine.backward(c_both, dp, False, sink)                            # :201, :208
        else:
            p_real, c_real = D.engine_forward(real_h, B)
            p_fake, c_fake = D.engine_forward(fake_h, B)
            dp_real = ops.bce_forward_backward(p_real, 1.0, 1.0, losses[0:1], False, True)
            dp_fake = ops.bce_forward_backward(p_fake, 0.0, 1.0, losses[1:2], False, True)
            D._engine.backward(c_real, dp_real, False, sink)
            D._engine.backward(c_fake, dp_fake, False, sink)
        self.opt_D.step()                                                          # :209
        self.opt_G.zero_grad(memset=False)                                         # :212
        p_adv, c_adv = D.engine_forward(fake_h, B)                                 # :214
        dp_adv = ops.bce_forward_backward(p_adv, 1.0, 1.0, losses[2:3], False, True)   # :215
        d_fake = D._engine.backward(c_adv, dp_adv, True, sink, param_grads=not self.elide_dead_grads)   # :216
        self._gen_backward(ctxG, fake, d_fake, sink)
        self.opt_G.step()                                                          # :217
        return losses


class WGANTrainer(_GANBase):
    """train_wgan's loop body (gan_code.py:296-331): ``critic_iters`` critic updates, each followed by the weight
    clamp, then one generator update.  The critic is the same Discriminator, sigmoid included (gan_code.py:270)."""
    LOSS_NAMES = ("d_loss", "g_loss")

    def __init__(self, netG, netD, optimizerG, optimizerD, clip_value: float = 0.01, critic_iters: int = 5, **kw):
        super().__init__(netG, netD, optimizerG, optimizerD, **kw)
        self.clip_value, self.critic_iters = clip_value, critic_iters              # :281-282

    def train_step(self, real: torch.Tensor, critic_noise: Optional[torch.Tensor] = None,
                   gen_noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """critic_noise: [critic_iters, B, nz, 1, 1] (the randn of :309 per critic iteration), gen_noise
        [B, nz, 1, 1] (:325).  -> device tensor [d_loss of the last critic iteration, g_loss]."""
        _need_cuda(real, "WGANTrainer.train_step")
        D, dt = self.D, self.dt
        B, dev, nz = real.shape[0], real.device, self.G.nz
        if critic_noise is None or gen_noise is None:
            ns = ops.default_noise(dev)
            ns.advance()
            if critic_noise is None:
                critic_noise = ns.randn((self.critic_iters, B, nz, 1, 1), 0)
            if gen_noise is None:
                gen_noise = ns.randn((B, nz, 1, 1), 1)
        losses = ops.zeros_f32(4, dev)
        sink = GradSink(direct=True)
        CP = G.padc(D.nc, dt)
        both = ops.empty_act((2 * B, real.shape[2], real.shape[3], CP), dt, dev)
        real_h = ops.nchw_to_nhwc(real.contiguous(), CP, dt, out=both[:B])
        grouped = self.group_d_passes and D._engine.can_group(B, 2, both)
        for it in range(self.critic_iters):                                        # :301
            self.opt_D.zero_grad(memset=False)                                     # :302
            fake, _ = self._gen(critic_noise[it], False)                           # :309-310 (.detach(): forward only)
            fake_h = ops.nchw_to_nhwc(fake, CP, dt, out=both[B:])
            if grouped:
                p_both, c_both = D.engine_forward(both, B, groups=2)
                dp = torch.empty_like(p_both)
                ops.mean_forward_backward(p_both[:B], -1.0, 1.0, losses[0:1], False, True, out=dp[:B])   # :305-306
                ops.mean_forward_backward(p_both[B:], 1.0, 1.0, losses[0:1], True, True, out=dp[B:])     # :311-315
                D._engine.backward(c_both, dp, False, sink)
            else:
                p_real, c_real = D.engine_forward(real_h, B)
                p_fake, c_fake = D.engine_forward(fake_h, B)
                dp_real = ops.mean_forward_backward(p_real, -1.0, 1.0, losses[0:1], False, True)
                dp_fake = ops.mean_forward_backward(p_fake, 1.0, 1.0, losses[0:1], True, True)
                D._engine.backward(c_real, dp_real, False, sink)
                D._engine.backward(c_fake, dp_fake, False, sink)
            self.opt_D.step()                                                      # :317
            ops.clamp_(self.opt_D.flat_p, -self.clip_value, self.clip_value)       # :320-321
            bump_weights_epoch(self.opt_D.params)
        self.opt_G.zero_grad(memset=False)                                         # :324
        fake, ctxG = self._gen(gen_noise, True)                                    # :325-326
        fake_h = ops.nchw_to_nhwc(fake, CP, dt, out=both[B:])
        p_adv, c_adv = D.engine_forward(fake_h, B)                                 # :327
        dp_adv = ops.mean_forward_backward(p_adv, -1.0, 1.0, losses[1:2], False, True)   # :328
        d_fake = D._engine.backward(c_adv, dp_adv, True, sink, param_grads=not self.elide_dead_grads)   # :330
        self._gen_backward(ctxG, fake, d_fake, sink)
        self.opt_G.step()                                                          # :331
        return losses

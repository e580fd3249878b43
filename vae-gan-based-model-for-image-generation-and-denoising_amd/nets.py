"""Host-side mirror of the reference's network classes -- same names, constructor signatures,
module tree and state_dict keys -- whose forward/backward run as HIP kernel
chains (engine.py) instead of ATen:

    ConvBlock, Encoder          main_vae.py:20-58
    Generator, Discriminator    gan_code.py:16-89
    weights_init                gan_code.py:91-97

The leaf layers are thin subclasses of the torch.nn containers (they only own parameters /
buffers and keep the class names ``Conv*`` / ``BatchNorm*`` that ``weights_init`` matches on);
their own forward() refuses to run -- computation happens only through the owning network, on
the MI355X.  Extra keyword arguments (``img_size`` for Generator/Discriminator, ``dtype``)
default to the reference behaviour (256x256, fp32).
"""
import math
from typing import List

import torch
import torch.nn as nn

from . import geometry as G
from . import ops
from ._lib import VG_ACT_LRELU, VG_ACT_NONE, VG_ACT_RELU
from .engine import GradSink, Stage, StackEngine

_DT = {"fp32": G.F32, "float32": G.F32, "f32": G.F32, torch.float32: G.F32,
       "bf16": G.BF16, "bfloat16": G.BF16, torch.bfloat16: G.BF16}
# "fp8": bf16 engine (storage, backward, optimizer masters as for "bf16") whose forward GEMMs of the wide conv layers read
# e4m3 copies of their operands (BASELINE configs[4]: a roofline run, the reference has no fp8 semantics)
_FP8_NAMES = ("fp8", "float8", "e4m3")



DIRECT_PARAM_GRADS = True     # see _StackFn.backward


def _is_fp8(dtype) -> bool:
    return isinstance(dtype, str) and dtype in _FP8_NAMES


def _dtype_code(dtype) -> int:
    if _is_fp8(dtype):
        return G.BF16
    try:
        return _DT[dtype]
    except KeyError:
        raise ValueError(f"dtype must be 'fp32', 'bf16' or 'fp8', got {dtype!r}") from None


def _no_standalone(self, *a, **k):
    raise RuntimeError(f"{type(self).__name__} is a parameter container of the MI355X engine: call the owning "
                       "Encoder / Generator / Discriminator / ConvBlock instead (no ATen/CPU fallback exists)")


class Conv2d(nn.Conv2d):
    forward = _no_standalone


class ConvTranspose2d(nn.ConvTranspose2d):
    forward = _no_standalone


class BatchNorm2d(nn.BatchNorm2d):
    forward = _no_standalone


class Linear(nn.Linear):
    forward = _no_standalone


class LeakyReLU(nn.LeakyReLU):
    forward = _no_standalone


class ReLU(nn.ReLU):
    forward = _no_standalone


class Tanh(nn.Tanh):
    forward = _no_standalone


class Sigmoid(nn.Sigmoid):
    forward = _no_standalone


def weights_init(m):
    """gan_code.py:91-97 (matches on the class-name substrings 'Conv' / 'BatchNorm')."""
    classname = m.__class__.__name__
    if classname.find('Conv') != -1:
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif classname.find('BatchNorm') != -1:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0)


def _check_input(x: torch.Tensor, what: str) -> None:
    if not x.is_cuda:
        raise RuntimeError(f"{what}: input is on {x.device}; this engine only runs on the MI355X ('cuda'). "
                           "Move the module and its inputs with .to('cuda') (no CPU fallback exists)")
    if x.dtype != torch.float32:
        raise RuntimeError(f"{what}: expected a float32 input tensor (the reference's dtype), got {x.dtype}")


class _StackFn(torch.autograd.Function):
    """autograd glue: forward/backward of a whole network as one node (tensors are only handles)."""

    @staticmethod
    def forward(ctx, net, x, *params):
        out, saved = net._forward_impl(x, keep=True)      # grad mode is off inside forward(); the caller decided
        ctx.net, ctx.saved = net, saved
        ctx.nparams = len(params)
        return out

    @staticmethod
    def backward(ctx, *douts):
        net = ctx.net
        if ctx.saved is None:
            raise RuntimeError("backward called on a forward that ran under torch.no_grad()")
        # Parameter gradients.  Default (DIRECT_PARAM_GRADS): for a parameter whose .grad IS its slot of optim.Adam's flat
        # gradient buffer (and that autograd wants a gradient for) the kernels write / accumulate straight into that slot
        # -- overwrite-after-zero_grad as in the direct trainer -- and autograd is handed None for it: no AccumulateGrad
        # add kernel per parameter (91 small ATen launches per reference-shaped iteration).  Every other parameter (no
        # vaegan_amd.Adam, .grad replaced or None, torch.autograd.grad(...) with DIRECT_PARAM_GRADS = False) gets its
        # gradient RETURNED, so AccumulateGrad hooks, torch DDP and foreign optimizers keep working.
        params = net._engine.params()
        wanted = {id(p) for p, need in zip(params, ctx.needs_input_grad[2:]) if need}
        need_p = bool(wanted)
        sink = GradSink(direct="homed" if DIRECT_PARAM_GRADS else False, wanted=wanted)
        dx = net._backward_impl(ctx.saved, douts, ctx.needs_input_grad[1], sink, param_grads=need_p)
        grads = [sink.out.get(id(p)) if id(p) in wanted else None for p in params]
        ctx.saved = None
        return (None, dx, *grads)


class _EngineNet(nn.Module):
    """Common plumbing of the three networks."""
    _engine: StackEngine

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        if getattr(self, "_engine", None) is not None:
            self._engine.invalidate()
        return r

    def state_dict(self, *a, **k):
        if getattr(self, "_engine", None) is not None:
            self._engine.flush_bn_ticks()        # lazily applied num_batches_tracked increments
        return super().state_dict(*a, **k)

    def load_state_dict(self, *a, **k):
        if getattr(self, "_engine", None) is not None:
            self._engine.flush_bn_ticks()
        r = super().load_state_dict(*a, **k)
        if getattr(self, "_engine", None) is not None:
            self._engine.invalidate()
        return r

    def _call_engine(self, x):
        params = self._engine.params()
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
            return _StackFn.apply(self, x, *params)
        out, _ = self._forward_impl(x, keep=False)
        return out


# =================================================================================================
class ConvBlock(_EngineNet):
    """main_vae.py:20-31: Conv2d(k=4, s=2, p=0, bias) -> BatchNorm2d -> LeakyReLU(0.01)."""

    def __init__(self, in_channels, out_channels, kernel_size=4, stride=2, dtype="fp32"):
        super(ConvBlock, self).__init__()
        self.conv = Conv2d(in_channels, out_channels, kernel_size, stride)
        self.bn = BatchNorm2d(out_channels)
        self.leaky_relu = LeakyReLU(inplace=True)
        self._k, self._s = kernel_size, stride
        self._dt = _dtype_code(dtype)
        self._engine = None
        self._hin = None

    def _stage(self, hin: int) -> Stage:
        return Stage("conv", self.conv.in_channels, self.conv.out_channels, self._k, self._s, 0, hin,
                     G.conv_out(hin, self._k, self._s, 0), conv=self.conv, bn=self.bn, act=VG_ACT_LRELU,
                     slope=self.leaky_relu.negative_slope)

    def _forward_impl(self, x, keep):
        B, C, H, W = x.shape
        if H != W:
            raise RuntimeError("square images only")
        if self._engine is None or self._hin != H:
            self._engine = StackEngine([self._stage(H)], self._dt, C)
            self._hin = H
        xh = ops.nchw_to_nhwc(x.contiguous(), G.padc(C, self._dt), self._dt)
        out, saved = self._engine.forward(xh, B, self.training, keep)
        return ops.nhwc_to_nchw(out, self.conv.out_channels, self._dt), (saved if keep else None)

    def _backward_impl(self, saved, douts, need_dx, sink, param_grads=True):
        d = ops.nchw_grad_to_nhwc(douts[0].contiguous(), None, G.padc(self.conv.out_channels, self._dt), self._dt)
        dx = self._engine.backward(saved, d, need_dx, sink, param_grads)
        return ops.nhwc_to_nchw(dx, self.conv.in_channels, self._dt) if need_dx else None

    def forward(self, x):
        _check_input(x, "ConvBlock")
        if self._engine is None or self._hin != x.shape[-1]:
            self._engine = StackEngine([self._stage(x.shape[-1])], self._dt, x.shape[1])
            self._hin = x.shape[-1]
        return self._call_engine(x)


# =================================================================================================
class Encoder(_EngineNet):
    """main_vae.py:34-58.  forward(x[B,C,S,S]) -> (mu[B,latent], logvar[B,latent])."""

    def __init__(self, img_size, latent_dim, dtype="fp32"):
        super(Encoder, self).__init__()
        channels = [img_size[0], 32, 64, 128, 256]
        layers = []
        for i in range(1, len(channels)):
            layers.append(ConvBlock(channels[i - 1], channels[i], dtype=dtype))
        self.cnn = nn.Sequential(*layers)
        self._dt = _dtype_code(dtype)
        self._img = int(img_size[1])
        if img_size[1] != img_size[2]:
            raise RuntimeError("square images only")
        # main_vae.py:43-45 runs self.cnn on zeros(1,C,H,W) in TRAIN mode to read the flattened size.
        # Its observable side effects are reproduced analytically -- the conv of a
        # zero map is its bias, BatchNorm of a constant map is beta (=0), LeakyReLU(0)=0, so every
        # block sees zeros: running_mean = 0.1*bias, running_var = 0.9, num_batches_tracked = 1.
        size = self._img
        self._sizes: List[int] = []
        for blk in self.cnn:
            size = G.conv_out(size, 4, 2, 0)
            self._sizes.append(size)
            with torch.no_grad():
                blk.bn.running_mean.copy_(blk.bn.running_mean * (1 - blk.bn.momentum)
                                          + blk.bn.momentum * blk.conv.bias)
                blk.bn.running_var.copy_(blk.bn.running_var * (1 - blk.bn.momentum) + blk.bn.momentum * 0.0)
                blk.bn.num_batches_tracked += 1
        self.flatten_size = channels[-1] * size * size
        self.latent_dim = latent_dim
        self.fc_mu = Linear(self.flatten_size, latent_dim)
        self.fc_logvar = Linear(self.flatten_size, latent_dim)
        # the two heads run as ONE GEMM on [fc_mu | fc_logvar]: ask the optimizer to home them back to back in its
        # flat buffers (optim.Adam), so the fused operand / gradient is a VIEW and no concatenation is ever launched
        self.fc_logvar.weight._vg_follows = self.fc_mu.weight
        self.fc_logvar.bias._vg_follows = self.fc_mu.bias
        stages = []
        hin = self._img
        for blk, hout in zip(self.cnn, self._sizes):
            stages.append(Stage("conv", blk.conv.in_channels, blk.conv.out_channels, 4, 2, 0, hin, hout,
                                conv=blk.conv, bn=blk.bn, act=VG_ACT_LRELU, slope=blk.leaky_relu.negative_slope))
            hin = hout
        stages.append(Stage("linear2", channels[-1], 2 * latent_dim, hin, 1, 0, hin, 1,
                            conv=self.fc_mu, conv2=self.fc_logvar))
        self._engine = StackEngine(stages, self._dt, img_size[0], fp8_fprop=_is_fp8(dtype))

    # mulv-level API used by trainer.py (no mu/logvar split, no autograd)
    def engine_forward(self, x_nchw, keep=True, x_nhwc=None):
        """x_nhwc: the same batch already in the engine's NHWC layout (the trainer converts it together with the
        Discriminator's noisy copy in one pass)."""
        B, C, H, W = x_nchw.shape
        if H != self._img or W != self._img:
            raise RuntimeError(f"Encoder was built for {self._img}x{self._img} images, got {H}x{W} "
                               f"(mat1 and mat2 shapes cannot be multiplied)")
        xh = x_nhwc if x_nhwc is not None else ops.nchw_to_nhwc(x_nchw.contiguous(), G.padc(C, self._dt), self._dt)
        mulv, saved = self._engine.forward(xh, B, self.training, keep)
        return mulv.view(B, -1), saved

    def _forward_impl(self, x, keep):
        mulv, saved = self.engine_forward(x, keep)
        L = self.latent_dim
        mu = mulv[:, :L].float().contiguous()
        logvar = mulv[:, L:2 * L].float().contiguous()
        return (mu, logvar), ((saved, mulv.shape) if keep else None)

    def _backward_impl(self, saved, douts, need_dx, sink, param_grads=True):
        ctxpack, shape = saved
        dmu, dlv = douts
        L = self.latent_dim
        d = torch.zeros(shape, dtype=ops.TORCH_DT[self._dt], device=(dmu if dmu is not None else dlv).device)
        if dmu is not None:
            d[:, :L] = dmu
        if dlv is not None:
            d[:, L:2 * L] = dlv
        dx = self._engine.backward(ctxpack, d.view(shape[0], 1, 1, -1), need_dx, sink, param_grads)
        return ops.nhwc_to_nchw(dx, self._engine.in_ch, self._dt) if need_dx else None

    def forward(self, x):
        _check_input(x, "Encoder")
        return self._call_engine(x)


# =================================================================================================
def _n_drop(img_size: int) -> int:
    n = int(round(math.log2(256 / img_size)))
    if img_size * (2 ** n) != 256 or not (0 <= n <= 4):
        raise ValueError(f"img_size must be one of 256, 128, 64, 32, 16 (size rule A0), got {img_size}")
    return n


class Generator(_EngineNet):
    """gan_code.py:16-54.  forward(z[B,nz,1,1]) -> image [B,nc,S,S] in (-1,1).

    img_size=256 is the reference network; smaller sizes drop the last log2(256/S) stride-2 stages
    (size rule A0); Sequential indices stay consecutive."""

    def __init__(self, nz=128, ngf=64, nc=3, img_size=256, dtype="fp32"):
        super(Generator, self).__init__()
        n = _n_drop(img_size)
        chans = [ngf * 16, ngf * 8, ngf * 4, ngf * 2, ngf, ngf // 2, ngf // 4]
        chans = chans[:len(chans) - n]
        mods = [ConvTranspose2d(nz, chans[0], 4, 1, 0, bias=False), BatchNorm2d(chans[0]), ReLU(True)]
        for i in range(1, len(chans)):
            mods += [ConvTranspose2d(chans[i - 1], chans[i], 4, 2, 1, bias=False), BatchNorm2d(chans[i]), ReLU(True)]
        mods += [ConvTranspose2d(chans[-1], nc, 3, 1, 1, bias=False), Tanh()]
        self.main = nn.Sequential(*mods)
        self._dt = _dtype_code(dtype)
        self.nz, self.nc, self.img_size = nz, nc, img_size
        stages, h = [], 1
        stages.append(Stage("convT", nz, chans[0], 4, 1, 0, 1, 4, conv=self.main[0], bn=self.main[1], act=VG_ACT_RELU))
        h = 4
        for i in range(1, len(chans)):
            stages.append(Stage("convT", chans[i - 1], chans[i], 4, 2, 1, h, 2 * h, conv=self.main[3 * i],
                                bn=self.main[3 * i + 1], act=VG_ACT_RELU))
            h *= 2
        stages.append(Stage("convT", chans[-1], nc, 3, 1, 1, h, h, conv=self.main[3 * len(chans)]))
        assert h == img_size
        self._engine = StackEngine(stages, self._dt, nz, fp8_fprop=_is_fp8(dtype))

    def engine_forward(self, z_nhwc, B, keep=True, tail=None):
        """z_nhwc: [B,1,1,ZP] engine dtype -> (pre-tanh NHWC image, ctx).
        tail = dict(noise=..., sigma=..., out_noisy=...) (see fused_tail): -> (tanh image NCHW f32, ctx) with Tanh, the
        layout change and the instance-noise add done by the last layer's kernel."""
        return self._engine.forward(z_nhwc, B, self.training, keep, tail=tail)

    def fused_tail(self, B) -> bool:
        """True when the last ConvTranspose2d runs on the edge-layer kernel, which can apply Tanh and emit the NCHW
        image (+ the noisy NHWC copy for the Discriminator) itself."""
        return self._engine.tn(len(self._engine.stages) - 1, B, "fprop") is not None

    def _forward_impl(self, z, keep):
        if z.dim() != 4 or z.shape[1] != self.nz or z.shape[2] != 1 or z.shape[3] != 1:
            raise RuntimeError(f"Generator expects z of shape [B,{self.nz},1,1], got {tuple(z.shape)}")
        B = z.shape[0]
        zh = ops.nchw_to_nhwc(z.contiguous(), G.padc(self.nz, self._dt), self._dt)
        if self.fused_tail(B):
            img, saved = self.engine_forward(zh, B, keep, tail={})
        else:
            pre, saved = self.engine_forward(zh, B, keep)
            img = ops.nhwc_to_nchw(pre, self.nc, self._dt, apply_tanh=True)
        return img, ((saved, img) if keep else None)

    def _backward_impl(self, saved, douts, need_dx, sink, param_grads=True):
        ctxpack, img = saved
        d = ops.nchw_grad_to_nhwc(douts[0].contiguous(), img, G.padc(self.nc, self._dt), self._dt)
        dz = self._engine.backward(ctxpack, d, need_dx, sink, param_grads)
        return ops.nhwc_to_nchw(dz, self.nz, self._dt) if need_dx else None

    def forward(self, input):
        _check_input(input, "Generator")
        return self._call_engine(input)


class Discriminator(_EngineNet):
    """gan_code.py:56-89.  forward(x[B,nc,S,S]) -> probabilities [B] (the reference's .view(-1)).

    img_size=256 is the reference network; smaller sizes drop the first log2(256/S) stages (rule A0)."""

    def __init__(self, ndf=64, nc=3, img_size=256, dtype="fp32"):
        super(Discriminator, self).__init__()
        n = _n_drop(img_size)
        chans = [ndf // 4, ndf // 2, ndf, ndf * 2, ndf * 4, ndf * 8][n:]
        mods = [Conv2d(nc, chans[0], 4, 2, 1, bias=False), LeakyReLU(0.2, inplace=True)]
        for i in range(1, len(chans)):
            mods += [Conv2d(chans[i - 1], chans[i], 4, 2, 1, bias=False), BatchNorm2d(chans[i]),
                     LeakyReLU(0.2, inplace=True)]
        mods += [Conv2d(chans[-1], 1, 4, 1, 0, bias=False), Sigmoid()]
        self.main = nn.Sequential(*mods)
        self._dt = _dtype_code(dtype)
        self.nc, self.img_size = nc, img_size
        stages, h = [], img_size
        stages.append(Stage("conv", nc, chans[0], 4, 2, 1, h, h // 2, conv=self.main[0], act=VG_ACT_LRELU, slope=0.2))
        h //= 2
        for i in range(1, len(chans)):
            stages.append(Stage("conv", chans[i - 1], chans[i], 4, 2, 1, h, h // 2, conv=self.main[3 * i - 1],
                                bn=self.main[3 * i], act=VG_ACT_LRELU, slope=0.2))
            h //= 2
        assert h == 4
        stages.append(Stage("head", chans[-1], 1, 4, 1, 0, 4, 1, conv=self.main[3 * len(chans) - 1]))
        self._engine = StackEngine(stages, self._dt, nc, fp8_fprop=_is_fp8(dtype))

    def engine_forward(self, x_nhwc, B, keep=True, groups=1):
        return self._engine.forward(x_nhwc, B, self.training, keep, groups=groups)

    def _forward_impl(self, x, keep):
        B, C, H, W = x.shape
        if C != self.nc:
            raise RuntimeError(f"Discriminator expects {self.nc} input channels, got {C}")
        if H != self.img_size or W != self.img_size:
            h = H
            for st in self._engine.stages:          # raises the reference's error when the map gets too small
                h = G.conv_out(h, st.k, st.s, st.p)
            raise RuntimeError(f"Discriminator was built for {self.img_size}x{self.img_size} images, got {H}x{W}")
        xh = ops.nchw_to_nhwc(x.contiguous(), G.padc(C, self._dt), self._dt)
        p, saved = self.engine_forward(xh, B, keep)
        return p, (saved if keep else None)

    def _backward_impl(self, saved, douts, need_dx, sink, param_grads=True):
        dx = self._engine.backward(saved, douts[0].contiguous().float(), need_dx, sink, param_grads)
        return ops.nhwc_to_nchw(dx, self.nc, self._dt) if need_dx else None

    def forward(self, input):
        _check_input(input, "Discriminator")
        return self._call_engine(input)

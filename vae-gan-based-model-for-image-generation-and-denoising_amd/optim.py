"""Adam with the constructor / zero_grad / step / state_dict surface of ``torch.optim.Adam`` as the
reference uses it (vaegan_code.py:42-44: ``optim.Adam(net.parameters(), lr=2e-4)``), executed as ONE
HIP kernel over a flat fp32 buffer (vg_adam_step; arithmetic of torch 2.10 ``_single_tensor_adam``,
SURVEY.md A12 / App. A.5).

At construction the parameters are re-homed into one contiguous buffer (``p.data`` becomes a view)
and ``p.grad`` becomes a view into a matching flat gradient buffer, so:
  * ``zero_grad()`` is one memset (gradients read as zeros instead of ``None`` -- the only observable
    difference from the reference, which nobody on the hot path observes);
  * autograd and the direct engine both accumulate in place into that buffer;
  * data-parallel training all-reduces ONE large buffer per network (ddp.py), sized for xGMI.
Construct it AFTER ``.to(device)`` (as vaegan_code.py:29-44 does); moving the module afterwards
would detach the parameters from the flat buffer.
"""
from typing import Iterable

import torch

from . import ops
from .engine import bump_weights_epoch


class Adam:
    def __init__(self, params: Iterable[torch.Tensor], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0,
                 amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("only the torch.optim.Adam defaults used by the reference are implemented "
                                      "(weight_decay=0, amsgrad=False)")
        if lr < 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")                  # torch.optim.Adam's own check and wording
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        dev = self.params[0].device
        for p in self.params:
            if not p.is_cuda or p.dtype != torch.float32:
                raise RuntimeError("vaegan_amd.Adam needs float32 parameters on the MI355X ('cuda'); "
                                   "construct it after .to(device) as vaegan_code.py:29-44 does")
            if p.device != dev:
                raise RuntimeError("all parameters of one optimizer must live on one device")
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False)
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        # Placement in the flat buffers.  Default: parameter order.  A parameter tagged `_vg_follows = other` (the
        # Encoder's fc_logvar after fc_mu, nets.py) is homed directly behind `other`, so that the pair forms one
        # contiguous [2N, ...] operand.  Indices in state_dict() stay those of self.params.
        ids = {id(p): i for i, p in enumerate(self.params)}
        followers = {}
        for i, p in enumerate(self.params):
            q = getattr(p, "_vg_follows", None)
            if q is not None and id(q) in ids and ids[id(q)] != i:
                followers.setdefault(ids[id(q)], []).append(i)
        placed, order = set(), []

        def place(i):
            if i in placed:
                return
            placed.add(i)
            order.append(i)
            for j in followers.get(i, ()):
                place(j)

        for i, p in enumerate(self.params):
            q = getattr(p, "_vg_follows", None)
            if q is not None and id(q) in ids and ids[id(q)] != i and ids[id(q)] not in placed:
                continue                                      # comes right after its leader
            place(i)
        for i in range(len(self.params)):
            place(i)
        self.offsets, n = [0] * len(self.params), 0
        for i in order:
            self.offsets[i] = n
            n += (self.params[i].numel() + 3) // 4 * 4         # keep every parameter 16-byte aligned
        self.numel = n
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.state_dev = torch.zeros(4, dtype=torch.float32, device=dev)   # [t, lr/(1-b1^t), sqrt(1-b2^t), -]
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                self.flat_p[o:o + p.numel()].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + p.numel()].view(p.shape)
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)
                p._vg_fresh = True
                p._vg_homed = p.grad.data_ptr()            # address of this parameter's slot in the flat gradient buffer
        self.grad_scale = 1.0                                   # 1/world_size under data parallelism
        self.steps = 0
        bump_weights_epoch(self.params)

    # -- torch.optim.Optimizer surface ---------------------------------------------------------------
    @property
    def param_groups(self):
        return [dict(params=self.params, lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=0, amsgrad=False)]

    def zero_grad(self, set_to_none: bool = True, memset: bool = True) -> None:
        """Gradients become zero (one memset).  memset=False only re-arms the 'overwrite on next write' flags --
        enough for the direct engine, which writes every gradient exactly once before accumulating."""
        if memset:
            ops.memset_zero(self.flat_g)
        for p in self.params:
            p._vg_fresh = True

    def step(self, closure=None, prepared: bool = False):
        """prepared=True: this iteration's ops.step_prologue() has already advanced the step counter and the bias
        corrections of this optimizer on the device; only the update kernel is launched."""
        if closure is not None:
            raise NotImplementedError("closure is not supported")
        for p, o in zip(self.params, self.offsets):
            g = p.grad
            if g is None or g.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                # someone replaced .grad (e.g. zero_grad(set_to_none=True) from foreign code): re-home it
                with torch.no_grad():
                    view = self.flat_g[o:o + p.numel()].view(p.shape)
                    if g is None:
                        view.zero_()
                    else:
                        view.copy_(g)
                    p.grad = view
        ops.adam_step(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0],
                      self.betas[1], self.eps, self.grad_scale, self.state_dev, prepared=prepared)
        self.steps += 1
        bump_weights_epoch(self.params)

    def step_pair(self, other: "Adam") -> None:
        """self.step(prepared=True); other.step(prepared=True) as ONE launch (both optimizers prepared by this iteration's
        ops.step_prologue); results are bit-identical to the two launches."""
        for o in (self, other):
            for p, off in zip(o.params, o.offsets):
                g = p.grad
                if g is None or g.data_ptr() != o.flat_g.data_ptr() + 4 * off:
                    raise RuntimeError("step_pair needs every .grad homed in the optimizer's flat gradient buffer")
        ops.adam_apply2(self, other)
        for o in (self, other):
            o.steps += 1
            bump_weights_epoch(o.params)

    def state_dict(self):
        st = {}
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            st[i] = dict(step=torch.tensor(float(self.steps)),
                         exp_avg=self.exp_avg[o:o + p.numel()].view(p.shape).clone(),
                         exp_avg_sq=self.exp_avg_sq[o:o + p.numel()].view(p.shape).clone())
        grp = dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=0, amsgrad=False,
                   params=list(range(len(self.params))))
        return dict(state=st, param_groups=[grp])

    def load_state_dict(self, sd):
        """Accepts a torch.optim.Adam state_dict (same format as state_dict() returns)."""
        grp = sd["param_groups"][0]
        self.lr, self.betas, self.eps = float(grp["lr"]), tuple(float(b) for b in grp["betas"]), float(grp["eps"])
        steps = 0
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                s = sd["state"].get(i)
                if s is None:
                    continue
                self.exp_avg[o:o + p.numel()].copy_(s["exp_avg"].reshape(-1))
                self.exp_avg_sq[o:o + p.numel()].copy_(s["exp_avg_sq"].reshape(-1))
                steps = int(float(s["step"]))
            self.steps = steps
            self.state_dev.zero_()
            self.state_dev[0] = float(steps)

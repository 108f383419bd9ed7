"""FlatAdamW: value clip + AdamW + gradient zeroing as ONE streaming HIP kernel per parameter
group (csrc/optim.hip), for the update of lib/solver.py:618-627 / train.py:226-238.

Parameters, gradients and both moments of a group live in flat float32 buffers; every
`p.data` / `p.grad` is a view into them (module state_dict keys and shapes are untouched).
Because gradients are views of one buffer per group they can be all-reduced in place in a few
large slices (ddp.GradBucketReducer.from_flat) and they are never re-allocated, which also makes
the step hipGraph-friendly: the step counter is a device scalar advanced by a captured kernel.

Numerically this is torch.optim.AdamW (amsgrad=False) preceded by clip_grad_value_; the fused
zeroing replaces the `zero_grad()` at the top of the next iteration.
"""
import ctypes

import torch

from . import _lib


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 clip_value=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.clip_value = float(clip_value)
        self._flat = []
        dev = None
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            if not ps:
                self._flat.append(None)
                continue
            for p in ps:
                if not p.is_cuda or p.dtype != torch.float32:
                    raise RuntimeError("FlatAdamW needs float32 parameters on the GPU")
            dev = ps[0].device
            # 4-element (16-byte) alignment of every parameter inside the flat buffers
            offs, total = [], 0
            for p in ps:
                offs.append(total)
                total += (p.numel() + 3) // 4 * 4
            flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
            flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
            for p, off in zip(ps, offs):
                flat_p[off:off + p.numel()].copy_(p.data.reshape(-1))
                p.data = flat_p[off:off + p.numel()].view_as(p)
                p.grad = flat_g[off:off + p.numel()].view_as(p)
            self._flat.append(dict(p=flat_p, g=flat_g, m=torch.zeros_like(flat_p),
                                   v=torch.zeros_like(flat_p), params=ps, offs=offs))
        self._step = torch.zeros((), dtype=torch.float32, device=dev)

    def flat_grad_buffers(self):
        return [f["g"] for f in self._flat if f is not None]

    def zero_grad(self, set_to_none=False):
        """Gradients are zeroed by step() itself; an explicit call zeroes the flat buffers in place
        (the views must survive, so `set_to_none` is ignored)."""
        for f in self._flat:
            if f is not None:
                f["g"].zero_()

    def _rebind(self, f):
        # something (e.g. zero_grad(set_to_none=True) elsewhere) detached a .grad view: restore it
        for p, off in zip(f["params"], f["offs"]):
            view = f["g"][off:off + p.numel()].view_as(p)
            if p.grad is None:
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        dev = self._step.device
        stream = _lib.stream_ptr(dev)
        with torch.cuda.device(dev):
            _lib.call("sig3d_step_increment", _lib.ptr(self._step), stream)
            for group, f in zip(self.param_groups, self._flat):
                if f is None:
                    continue
                self._rebind(f)
                b1, b2 = group["betas"]
                _lib.call("sig3d_adamw_flat", f["p"].numel(), _lib.ptr(f["p"]), _lib.ptr(f["g"]),
                          _lib.ptr(f["m"]), _lib.ptr(f["v"]), _lib.ptr(self._step),
                          ctypes.c_float(group["lr"]), ctypes.c_float(b1), ctypes.c_float(b2),
                          ctypes.c_float(group["eps"]), ctypes.c_float(group["weight_decay"]),
                          ctypes.c_float(self.clip_value), 1, stream)
        return loss

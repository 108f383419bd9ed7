"""One zero-filled scratch region per training step.

A step of the hot path needs ~25 accumulation buffers that must start at zero: the BatchNorm statistic
accumulators of the four SharedMLP stacks (forward and backward), the weight-gradient staging of the encoder,
the dQ images of the split cross-attention backward, the scattered feature gradients of the grouping layers, the
positional MLP's weight gradients.  Each used to be its own `torch.zeros` / `hipMemsetAsync`: ~25 launches of 2-6 us
in a chain whose kernels are 5-40 us long (profiles/r04_a_step.md, "torch glue").  Here they are slices of ONE
buffer that the step zeroes with ONE fill at its start (trainer.train_step / graph_step.GraphedTrainStep call
`begin_step`); the `_z` entry points of the library (include/sig3d_hip.h) take such pre-zeroed outputs.

The region is sized from the demand of the previous step (the first step of a process runs on plain `torch.zeros`);
a request that does not fit -- a larger batch, a second forward pass inside one step -- falls back to `torch.zeros`
and raises the size for the next step.  Outside a step (inference, unit tests of single modules) every request is a
plain `torch.zeros`.  Slices are only valid until the next `begin_step`: nothing handed to the caller of the model
may come from here.

Only the step's own stream may use it: the geometry chains of the NEXT batch run beside the step on their own
streams (geometry.GeometryPipeline) and keep their own buffers.  The region remembers the stream that opened the step; a
request made with another stream current gets a plain `torch.zeros` (not the thread: autograd runs the backward pass
of the same step on its own thread, with the forward's stream current).
"""

import torch

_ALIGN = 256
# False: every request is its own torch.zeros again (tests compare the two forms)
ENABLED = True


class StepZeros:
    def __init__(self):
        self._buf = None          # uint8, one device per process (one process per GPU)
        self._retired = []
        self._cursor = 0
        self._zeroed = 0
        self._demand = 0
        self._want = 0            # demand of the last complete step
        self._active = False
        self.grads_ok = False     # may slices become `.grad` of a parameter?  (begin_step: the caller's promise)
        self.hits = 0
        self.misses = 0
        self._owner = None        # the stream of the step that is open

    def begin_step(self, device, grads_ok=False):
        """On the step's stream, before its first kernel: zero the region (one fill).
        grads_ok: every parameter gradient of this step is consumed (and dropped) before the next begin_step -- a
        FlatAdamW step reads it and sets `.grad = None`; torch optimizers keep `.grad` tensors across steps."""
        device = torch.device(device)
        self.grads_ok = bool(grads_ok)
        self._cursor = self._demand = 0
        self._zeroed = 0
        self._active = ENABLED and device.type == "cuda"
        self._owner = torch.cuda.current_stream(device) if self._active else None
        if not self._active or self._want == 0:
            return
        if self._buf is None or self._buf.device != device or self._buf.numel() < self._want:
            if torch.cuda.is_current_stream_capturing():
                return            # never allocate the shared region from a graph's private pool
            if self._buf is not None:
                self._retired.append(self._buf)   # a captured hipGraph may still hold slices of it: never freed
            self._buf = torch.empty(self._want + self._want // 4, dtype=torch.uint8, device=device)
        self._buf[:self._want].zero_()
        self._zeroed = self._want

    def end_step(self):
        if self._active:
            self._want = self._demand
        self._active = False
        self._owner = None
        self.grads_ok = False

    def zeros(self, shape, dtype, device):
        """torch.zeros(shape, dtype=dtype, device=device), from the region when a step is open and it fits."""
        if isinstance(shape, int):
            shape = (shape,)
        n = 1
        for s in shape:
            n *= int(s)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        if self._active and self._owner != torch.cuda.current_stream(self._owner.device):
            return torch.zeros(shape, dtype=dtype, device=device)      # not the step: a side stream (geometry chain, ...)
        if not self._active or self._buf is None or torch.device(device) != self._buf.device:
            if self._active:
                self._demand += (nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
                self.misses += 1
            return torch.zeros(shape, dtype=dtype, device=device)
        padded = (nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
        self._demand += padded
        if self._cursor + padded > self._zeroed:
            self.misses += 1
            return torch.zeros(shape, dtype=dtype, device=device)
        out = self._buf[self._cursor:self._cursor + nbytes].view(dtype).view(shape)
        self._cursor += padded
        self.hits += 1
        return out

    def owns(self, t):
        """Does `t` lie in the zeroed part of the region (i.e. may a `_z` entry point skip its own fill)?"""
        if self._buf is None or not self._active or t is None:
            return False
        lo = self._buf.data_ptr()
        return lo <= t.data_ptr() < lo + self._zeroed


STEP_ZEROS = StepZeros()


def zeros(shape, dtype=torch.float32, device=None):
    return STEP_ZEROS.zeros(shape, dtype, device)

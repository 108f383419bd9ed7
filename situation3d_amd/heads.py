"""Host side of csrc/heads.hip: the answer and auxiliary MLP heads on the pooled Q-Former output.

`pooled_heads(rows, b, q, aux_reg, answer_cls)` = (aux_reg(pooled), answer_cls(pooled)) with pooled = the mean of the q
query rows of every sample (situation3d/models/sqa_module.py: the heads on the fused query tokens) as three forward and
three backward launches instead of torch's 4 + 8 GEMMs on 8 rows and ~19 launches between them.  The modules' own
parameters are used (same state_dict); anything the kernels do not cover (CPU tensors, more than 16 samples, another
head layout) takes the torch path of the very same modules.
"""
import ctypes
import itertools

import torch
import torch.nn as nn

from . import _lib, scratch

# False: the torch modules (tests compare the two forms; same results up to f32 summation order and the dropout stream)
ENABLED = True
_call_ids = itertools.count(0x5EAD0000)


class _PooledHeadsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows, b, q, w1a, b1a, w2a, b2a, w1c, b1c, w2c, b2c, p_drop, call_id):
        from .qformer import _rng_counter
        dev = rows.device
        hidden, n_aux, n_ans = w1a.shape[1], w2a.shape[0], w2c.shape[0]
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)   # noqa: E731
        pooled, pre, h, aux, ans = e(b, hidden), e(2, b, hidden), e(2, b, hidden), e(b, n_aux), e(b, n_ans)
        with torch.cuda.device(dev):
            _lib.call("sig3d_pooled_heads_fwd", b, q, hidden, n_aux, n_ans, _lib.ptr(rows), _lib.ptr(w1a), _lib.ptr(b1a),
                      _lib.ptr(w2a), _lib.ptr(b2a), _lib.ptr(w1c), _lib.ptr(b1c), _lib.ptr(w2c), _lib.ptr(b2c),
                      ctypes.c_float(p_drop), ctypes.c_uint(call_id), _lib.ptr(_rng_counter(dev)), _lib.ptr(pooled),
                      _lib.ptr(pre), _lib.ptr(h), _lib.ptr(aux), _lib.ptr(ans), _lib.stream_ptr(dev))
        ctx.save_for_backward(pooled, pre, h, w1a, w2a, w1c, w2c)
        ctx.cfg = (b, q, p_drop, call_id, tuple(rows.shape))
        return aux, ans

    @staticmethod
    def backward(ctx, daux, dans):
        from .qformer import _rng_counter
        pooled, pre, h, w1a, w2a, w1c, w2c = ctx.saved_tensors
        b, q, p_drop, call_id, rshape = ctx.cfg
        dev = pooled.device
        hidden, n_aux, n_ans = w1a.shape[1], w2a.shape[0], w2c.shape[0]
        sizes = [hidden * hidden, hidden, n_aux * hidden, n_aux, hidden * hidden, hidden, n_ans * hidden, n_ans]
        grads = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        work = torch.empty(int(_lib.load().sig3d_pooled_heads_work_floats(b, hidden)), dtype=torch.float32, device=dev)
        # the gradient of the whole row matrix: the query rows are written by the kernel, the rest is zero
        # (a slice of the step's zeroed region when a step is open: scratch.py)
        drows = scratch.zeros(rshape, torch.float32, dev) if rshape[0] > b * q else \
            torch.empty(rshape, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("sig3d_pooled_heads_bwd", b, q, hidden, n_aux, n_ans, _lib.ptr(daux.contiguous()),
                      _lib.ptr(dans.contiguous()), _lib.ptr(pooled), _lib.ptr(pre), _lib.ptr(h), _lib.ptr(w1a),
                      _lib.ptr(w2a), _lib.ptr(w1c), _lib.ptr(w2c), ctypes.c_float(p_drop), ctypes.c_uint(call_id),
                      _lib.ptr(_rng_counter(dev)), _lib.ptr(work), _lib.ptr(grads), _lib.ptr(drows), _lib.stream_ptr(dev))
        out, o = [], 0
        for n, shape in zip(sizes, [(hidden, hidden), (hidden,), (n_aux, hidden), (n_aux,), (hidden, hidden), (hidden,),
                                    (n_ans, hidden), (n_ans,)]):
            out.append(grads[o:o + n].view(shape))
            o += n
        return (drows, None, None) + tuple(out) + (None, None)


def _layout(seq, with_dropout):
    mods = list(seq) if isinstance(seq, nn.Sequential) else []
    want = [nn.Linear, nn.GELU, nn.Dropout, nn.Linear] if with_dropout else [nn.Linear, nn.GELU, nn.Linear]
    if len(mods) != len(want) or not all(isinstance(m, t) for m, t in zip(mods, want)):
        return None
    if getattr(mods[1], "approximate", "none") != "none" or mods[0].bias is None or mods[-1].bias is None:
        return None
    return mods


def covered(rows, b, q, aux_reg, answer_cls):
    a, c = _layout(aux_reg, False), _layout(answer_cls, True)
    if not (ENABLED and a and c and rows.is_cuda and rows.dtype == torch.float32 and rows.dim() == 2
            and rows.is_contiguous() and rows.shape[0] >= b * q and 1 <= b <= 16):
        return False
    hidden = rows.shape[1]
    return hidden % 16 == 0 and hidden <= 1024 and a[2].weight.shape[0] <= 1024 and c[3].weight.shape[0] <= 1024 and all(
        tuple(m.weight.shape) == (hidden, hidden) for m in (a[0], c[0])) and a[2].weight.shape[1] == hidden \
        and c[3].weight.shape[1] == hidden


def pooled_heads(rows, b, q, aux_reg, answer_cls):
    """rows (>= b*q, hidden): the first b*q rows are the query rows, sample-major.  -> (aux_scores, answer_scores)."""
    a, c = _layout(aux_reg, False), _layout(answer_cls, True)
    p = float(c[2].p) if c[2].training else 0.0
    return _PooledHeadsFn.apply(rows, b, q, a[0].weight, a[0].bias, a[2].weight, a[2].bias, c[0].weight, c[0].bias,
                                c[3].weight, c[3].bias, p, next(_call_ids) & 0xFFFFFFFF)

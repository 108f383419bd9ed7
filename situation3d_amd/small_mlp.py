"""Host side of csrc/small_mlp.hip: the small dense layers around the Q-Former as single launches.

`pos_embed_add(seq, x, residual)`: residual + seq(x) for seq = nn.Sequential(Linear(cin, hid), GELU(), Linear(hid, cout))
-- the positional MLP of the scene tokens (situation3d/models/sqa_module.py:274-278, applied at :319-321) -- as one
forward launch and two backward launches instead of torch's 4 + ~12.  The module's own parameters are used (same
state_dict); anything the kernels do not cover (CPU tensors, other shapes, a tanh GELU) takes the torch path.
"""

import torch
import torch.nn as nn

from . import _lib, scratch

# False: the torch modules (tests compare the two forms; same results up to f32 summation order)
ENABLED = True


class _PosMLPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, w1, b1, w2, b2, posed=None):
        """posed = (pose (b,7), points (b,T,3), inverse): x is None and is FORMED inside the launch -- the situational
        re-encode of the token positions rides along (sig3d_pos_mlp_fwd_posed); returns (out, x)."""
        hid, cout = w1.shape[0], w2.shape[0]
        dev = residual.device
        rows = residual.shape[0]
        pre = torch.empty((rows, hid), dtype=torch.float32, device=dev)
        out = torch.empty((rows, cout), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            if posed is None:
                cin = x.shape[1]
                _lib.call("sig3d_pos_mlp_fwd", rows, cin, hid, cout, _lib.ptr(x), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2),
                          _lib.ptr(b2), _lib.ptr(residual), _lib.ptr(pre), _lib.ptr(out), _lib.stream_ptr(dev))
            else:
                pose, points, inverse = posed
                b, tokens = points.shape[0], points.shape[1]
                x = torch.empty((rows, 3), dtype=torch.float32, device=dev)
                _lib.call("sig3d_pos_mlp_fwd_posed", b, tokens, hid, cout, int(inverse), _lib.ptr(pose), _lib.ptr(points),
                          _lib.ptr(x), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(residual),
                          _lib.ptr(pre), _lib.ptr(out), _lib.stream_ptr(dev))
        ctx.save_for_backward(x, w1, w2, pre)
        ctx.posed = posed is not None
        if posed is not None:
            ctx.mark_non_differentiable(x)
            return out, x
        return out

    @staticmethod
    def backward(ctx, dy, _dx=None):
        x, w1, w2, pre = ctx.saved_tensors
        rows, cin = x.shape
        hid, cout = w1.shape[0], w2.shape[0]
        dev = x.device
        dy = dy.contiguous()
        dpre = torch.empty((rows, hid), dtype=torch.float32, device=dev)
        n_grads = hid * cin + hid + cout * hid + cout
        if scratch.STEP_ZEROS.grads_ok:     # zeroed with the step's one fill (the slices become `.grad`s: scratch.py)
            grads, entry = scratch.zeros(n_grads, torch.float32, dev), "sig3d_pos_mlp_bwd_z"
        else:
            grads, entry = torch.empty(n_grads, dtype=torch.float32, device=dev), "sig3d_pos_mlp_bwd"
        with torch.cuda.device(dev):
            _lib.call(entry, rows, cin, hid, cout, _lib.ptr(x), _lib.ptr(w2), _lib.ptr(pre), _lib.ptr(dy),
                      _lib.ptr(dpre), _lib.ptr(grads), _lib.stream_ptr(dev))
        o = 0
        dw1 = grads[o:o + hid * cin].view(hid, cin); o += hid * cin
        db1 = grads[o:o + hid]; o += hid
        dw2 = grads[o:o + cout * hid].view(cout, hid); o += cout * hid
        db2 = grads[o:o + cout]
        dx = dpre.mm(w1) if (ctx.needs_input_grad[0] and not ctx.posed) else None
        return dx, (dy if ctx.needs_input_grad[1] else None), dw1, db1, dw2, db2, None


def _covered(seq, x, residual):
    if not (ENABLED and isinstance(seq, nn.Sequential) and len(seq) == 3):
        return False
    l1, act, l2 = seq[0], seq[1], seq[2]
    if not (isinstance(l1, nn.Linear) and isinstance(l2, nn.Linear) and isinstance(act, nn.GELU)):
        return False
    if getattr(act, "approximate", "none") != "none" or l1.bias is None or l2.bias is None:
        return False
    if not (x.is_cuda and x.dtype == torch.float32 and residual.is_cuda and residual.dtype == torch.float32):
        return False
    hid, cin = l1.weight.shape
    return cin <= 4 and hid <= 128 and hid % 4 == 0 and l2.weight.shape[1] == hid and l2.weight.shape[0] % 4 == 0 \
        and x.shape[-1] == cin and residual.shape[-1] == l2.weight.shape[0] and residual.shape[:-1] == x.shape[:-1]


def posed_pos_embed_add(seq, pose, points, residual, inverse=True):
    """residual + seq(situational_transform(pose, points, inverse)) as ONE launch -> (tokens, re-encoded positions), or
    None when the fused kernel does not cover the case (gradients wanted for pose / points, CPU tensors, other shapes):
    the caller then runs situational_transform and pos_embed_add."""
    if torch.is_grad_enabled() and (pose.requires_grad or points.requires_grad):
        return None
    if not (points.dim() == 3 and points.shape[-1] == 3 and pose.dim() == 2 and pose.shape[-1] == 7 and points.is_cuda
            and pose.is_cuda and pose.dtype == torch.float32 and points.dtype == torch.float32
            and residual.dim() == 3 and residual.shape[:2] == points.shape[:2]):
        return None
    probe = points.new_empty(points.shape)        # shape / dtype stand-in for the MLP's input
    if not _covered(seq, probe, residual):
        return None
    l1, l2 = seq[0], seq[2]
    shape = residual.shape
    out, x = _PosMLPFn.apply(None, residual.reshape(-1, shape[-1]).contiguous(), l1.weight, l1.bias, l2.weight, l2.bias,
                             (pose.contiguous(), points.contiguous(), bool(inverse)))
    return out.view(shape), x.view(points.shape)


def pos_embed_add(seq, x, residual):
    """residual + seq(x); x (..., cin), residual (..., cout)."""
    if not _covered(seq, x, residual):
        return residual + seq(x)
    l1, l2 = seq[0], seq[2]
    shape = residual.shape
    out = _PosMLPFn.apply(x.reshape(-1, x.shape[-1]).contiguous(), residual.reshape(-1, shape[-1]).contiguous(),
                          l1.weight, l1.bias, l2.weight, l2.bias)
    return out.view(shape)

"""Fused training-mode SharedMLP + neighbourhood max-pool (host side of csrc/shared_mlp.hip).

What it replaces, semantically unchanged: `self.mlp_module(grouped_features)` followed by
`F.max_pool2d(new_features, [1, nsample])` in PointnetSAModuleVotes.forward
(lib/pointnet2/pointnet2_modules.py:251-262), where mlp_module = pt_utils.SharedMLP =
[Conv2d 1x1 (no bias) -> BatchNorm2d -> ReLU] x L (lib/pointnet2/pytorch_utils.py:11-36).

The module's own parameters and buffers are used (same state_dict, running statistics updated in
place like nn.BatchNorm2d), so this is a pure execution-path change; `can_fuse` decides whether a
given SharedMLP / call qualifies, otherwise the caller keeps the layer-by-layer torch path.
"""
import ctypes

import torch
import torch.nn as nn

from .. import _lib


def _layers(mlp):
    """[(conv, bn)] if `mlp` is a plain post-activation Conv1x1+BN+ReLU stack, else None."""
    out = []
    for block in mlp.children():
        mods = list(block.children())
        if len(mods) != 3:
            return None
        conv, bnw, act = mods
        if not isinstance(conv, nn.Conv2d) or not isinstance(act, nn.ReLU):
            return None
        bns = list(bnw.children())
        if len(bns) != 1 or not isinstance(bns[0], nn.BatchNorm2d):
            return None
        bn = bns[0]
        if conv.kernel_size != (1, 1) or conv.stride != (1, 1) or conv.padding != (0, 0) \
                or conv.bias is not None or conv.groups != 1:
            return None
        if bn.momentum is None or not bn.affine or not bn.track_running_stats:
            return None
        if conv.out_channels % 32 != 0:
            return None
        out.append((conv, bn))
    return out or None


# Below this many (batch x npoint x nsample) positions the 1x1 convolutions go to the library GEMM
# (a few thousand 32-position tiles do not fill the MFMA kernel; measured cross-over on MI355X is
# between the SA2 (262 144 positions) and SA3 (65 536) shapes); BatchNorm / ReLU / pooling stay fused.
MIN_POSITIONS = 100000


def can_fuse(mlp, x, min_positions=None):
    """min_positions is kept for callers that force the decision; any position count qualifies now.
    Training mode: the autograd Function below.  Eval mode: the inference path (running statistics),
    only when no gradient is wanted (torch.no_grad / frozen inputs and weights)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    if _layers(mlp) is None:
        return False
    if mlp.training:
        return True
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in mlp.parameters()))
    return not needs_grad


def _aff_rows(t):
    return t[0], t[1], t[2], t[3]  # scale, shift, mean, invstd


class _FusedMLPMax(torch.autograd.Function):
    """library_gemm=False: every layer is sig3d_mlp_layer_fwd (MFMA GEMM with BatchNorm+ReLU on operand
    load and statistics in the epilogue).  library_gemm=True (small levels): the 1x1 convolutions and
    the input gradients are library GEMMs (torch.bmm), BatchNorm statistics / apply / pooling and the
    whole BatchNorm+ReLU backward stay on the kernels of csrc/shared_mlp.hip -- no MIOpen BatchNorm, no
    separate ReLU / threshold / max-reduce / scatter launches."""

    @staticmethod
    def forward(ctx, x, layers, library_gemm, *flat):
        # flat = (W_1, gamma_1, beta_1, W_2, gamma_2, beta_2, ...) so that autograd tracks them
        dev = x.device
        x = x.contiguous()
        b, c0, p, s = x.shape
        e = p * s
        stream = _lib.stream_ptr(dev)
        ys, affs, ws = [], [], []
        cur, ps, pb = x, None, None
        cmax = max(conv.out_channels for conv, _ in layers)
        with torch.cuda.device(dev):
            # all statistic accumulators of the stack are zeroed by ONE fill (accumulate = 1 below)
            st_all = torch.zeros((len(layers), 2, cmax), dtype=torch.float64, device=dev)
            for i, (conv, bn) in enumerate(layers):
                w = flat[3 * i].reshape(conv.out_channels, conv.in_channels).contiguous()
                gamma, beta = flat[3 * i + 1], flat[3 * i + 2]
                cout, cin = w.shape
                st = st_all[i]
                if library_gemm:
                    if ps is not None:  # materialise relu(bn(y_prev)) for the library GEMM
                        act = torch.empty_like(cur)
                        _lib.call("sig3d_bn_relu_apply", b, cin, e, _lib.ptr(cur), _lib.ptr(ps), _lib.ptr(pb),
                                  _lib.ptr(act), stream)
                    else:
                        act = cur
                    y = torch.bmm(w.unsqueeze(0).expand(b, cout, cin), act.view(b, cin, e)).view(b, cout, p, s)
                    _lib.call("sig3d_channel_stats", b, cout, e, _lib.ptr(y), _lib.ptr(st[0]), _lib.ptr(st[1]),
                              1, stream)
                else:
                    y = torch.empty((b, cout, p, s), dtype=torch.float32, device=dev)
                    _lib.call("sig3d_mlp_layer_fwd", b, cin, cout, e, _lib.ptr(cur), _lib.ptr(w),
                              _lib.ptr(ps), _lib.ptr(pb), _lib.ptr(y), _lib.ptr(st[0]), _lib.ptr(st[1]),
                              1, stream)
                aff = torch.empty((4, cout), dtype=torch.float32, device=dev)
                _lib.call("sig3d_bn_finalize", cout, ctypes.c_double(float(b) * e),
                          ctypes.c_float(bn.eps), ctypes.c_float(bn.momentum), _lib.ptr(st[0]),
                          _lib.ptr(st[1]), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(aff[0]),
                          _lib.ptr(aff[1]), _lib.ptr(aff[2]), _lib.ptr(aff[3]),
                          _lib.ptr(bn.running_mean), _lib.ptr(bn.running_var),
                          _lib.ptr(bn.num_batches_tracked), stream)
                ys.append(y)
                affs.append(aff)
                ws.append(w)
                cur, ps, pb = y, aff[0], aff[1]
            c_last = ws[-1].shape[0]
            out = torch.empty((b, c_last, p), dtype=torch.float32, device=dev)
            arg = torch.empty((b, c_last, p), dtype=torch.int32, device=dev)
            _lib.call("sig3d_bn_relu_maxpool", b, c_last, p, s, _lib.ptr(cur), _lib.ptr(ps),
                      _lib.ptr(pb), _lib.ptr(out), _lib.ptr(arg), stream)
        ctx.save_for_backward(x, arg, *ys, *affs, *ws)
        ctx.nl = len(layers)
        ctx.dims = (b, p, s)
        ctx.library_gemm = library_gemm
        return out

    @staticmethod
    def backward(ctx, grad_out):
        saved = ctx.saved_tensors
        nl = ctx.nl
        x, arg = saved[0], saved[1]
        ys = saved[2:2 + nl]
        affs = saved[2 + nl:2 + 2 * nl]
        ws = saved[2 + 2 * nl:2 + 3 * nl]
        b, p, s = ctx.dims
        e = p * s
        dev = x.device
        stream = _lib.stream_ptr(dev)
        grad_out = grad_out.contiguous()
        grads = [None] * (3 * nl)
        grad_x = None
        with torch.cuda.device(dev):
            dA = None
            # one fill each for every BatchNorm-gradient accumulator and every dW of the stack
            cmax = max(w.shape[0] for w in ws)
            n_dw = sum(w.numel() for w in ws)
            zeros = torch.zeros(nl * 2 * cmax * 8 + n_dw * 4, dtype=torch.uint8, device=dev)
            sums_all = zeros[:nl * 2 * cmax * 8].view(torch.float64).view(nl, 2, cmax)
            dw_all = zeros[nl * 2 * cmax * 8:].view(torch.float32)
            dw_off = 0
            for k in range(nl - 1, -1, -1):
                cout, cin = ws[k].shape
                scale, shift, mean, invstd = _aff_rows(affs[k])
                sums = sums_all[k]
                dY = torch.empty_like(ys[k])
                if k == nl - 1:
                    _lib.call("sig3d_bn_relu_bwd", b, cout, e, s, _lib.ptr(None), _lib.ptr(grad_out),
                              _lib.ptr(arg), _lib.ptr(ys[k]), _lib.ptr(scale), _lib.ptr(shift),
                              _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(sums[0]), _lib.ptr(sums[1]),
                              _lib.ptr(dY), 1, stream)
                else:
                    _lib.call("sig3d_bn_relu_bwd", b, cout, e, s, _lib.ptr(dA), _lib.ptr(None),
                              _lib.ptr(None), _lib.ptr(ys[k]), _lib.ptr(scale), _lib.ptr(shift),
                              _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(sums[0]), _lib.ptr(sums[1]),
                              _lib.ptr(dY), 1, stream)
                prev = ys[k - 1] if k > 0 else x
                pps = affs[k - 1][0] if k > 0 else None
                ppb = affs[k - 1][1] if k > 0 else None
                dW = dw_all[dw_off:dw_off + cout * cin].view(cout, cin)
                dw_off += cout * cin
                _lib.call("sig3d_mlp_layer_dw", b, cin, cout, e, _lib.ptr(dY), _lib.ptr(prev),
                          _lib.ptr(pps), _lib.ptr(ppb), _lib.ptr(dW), 1, stream)
                grads[3 * k] = dW.view(cout, cin, 1, 1)
                if k > 0 or ctx.needs_input_grad[0]:
                    if ctx.library_gemm:
                        dA = torch.bmm(ws[k].t().unsqueeze(0).expand(b, cin, cout), dY.view(b, cout, e)).view(b, cin, p, s)
                    else:
                        wt = ws[k].t().contiguous()  # (cin, cout): dA = W^T dY through the same GEMM
                        dA = torch.empty((b, cin, p, s), dtype=torch.float32, device=dev)
                        _lib.call("sig3d_mlp_layer_fwd", b, cout, cin, e, _lib.ptr(dY), _lib.ptr(wt),
                                  _lib.ptr(None), _lib.ptr(None), _lib.ptr(dA), _lib.ptr(None),
                                  _lib.ptr(None), 0, stream)   # no statistics for an input gradient
                    if k == 0:
                        grad_x = dA
            sums32 = sums_all.to(torch.float32)                # one conversion launch for the whole stack
            for k in range(nl):
                cout = ws[k].shape[0]
                grads[3 * k + 1] = sums32[k, 1, :cout]         # d gamma
                grads[3 * k + 2] = sums32[k, 0, :cout]         # d beta
        return (grad_x, None, None) + tuple(grads)


def _fused_mlp_max_eval(layers, x):
    """Inference: BatchNorm2d.eval() is the affine map scale = gamma / sqrt(running_var + eps),
    shift = beta - running_mean * scale, so a layer is one sig3d_mlp_layer_fwd (previous layer's
    BN+ReLU on operand load, no statistics) and the stack ends in sig3d_bn_relu_maxpool."""
    dev = x.device
    x = x.contiguous()
    b, _, p, s = x.shape
    e = p * s
    stream = _lib.stream_ptr(dev)
    cur, ps, pb = x, None, None
    with torch.no_grad(), torch.cuda.device(dev):
        for conv, bn in layers:
            w = conv.weight.reshape(conv.out_channels, conv.in_channels).contiguous()
            cout, cin = w.shape
            y = torch.empty((b, cout, p, s), dtype=torch.float32, device=dev)
            _lib.call("sig3d_mlp_layer_fwd", b, cin, cout, e, _lib.ptr(cur), _lib.ptr(w), _lib.ptr(ps),
                      _lib.ptr(pb), _lib.ptr(y), _lib.ptr(None), _lib.ptr(None), 0, stream)
            scale = (bn.weight * torch.rsqrt(bn.running_var + bn.eps)).contiguous()
            shift = (bn.bias - bn.running_mean * scale).contiguous()
            cur, ps, pb = y, scale, shift
        c_last = cur.shape[1]
        out = torch.empty((b, c_last, p), dtype=torch.float32, device=dev)
        arg = torch.empty((b, c_last, p), dtype=torch.int32, device=dev)
        _lib.call("sig3d_bn_relu_maxpool", b, c_last, p, s, _lib.ptr(cur), _lib.ptr(ps), _lib.ptr(pb),
                  _lib.ptr(out), _lib.ptr(arg), stream)
    return out


def fused_mlp_max(mlp, x, library_gemm=None):
    """max over nsample of SharedMLP(x): x (B,C,npoint,nsample) -> (B,C_out,npoint).
    library_gemm=None: decided by the position count (MIN_POSITIONS)."""
    layers = _layers(mlp)
    if not mlp.training:
        return _fused_mlp_max_eval(layers, x)
    flat = []
    for conv, bn in layers:
        flat += [conv.weight, bn.weight, bn.bias]
    if library_gemm is None:
        library_gemm = x.shape[0] * x.shape[2] * x.shape[3] < MIN_POSITIONS
    return _FusedMLPMax.apply(x, layers, bool(library_gemm), *flat)

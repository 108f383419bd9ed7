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
import os

import torch
import torch.nn as nn

from .. import _lib, scratch


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
MIN_POSITIONS = int(os.environ.get("SIG3D_MLP_MIN_POSITIONS", "100000"))


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


class CompactLists:
    """Distinct (centre, neighbour) pairs of a ball-query result (csrc/compact.hip): the first n_act[b]
    entries of every row of cidx / centre_of / mult are defined, seg_off[b, j] is the first entry of centre j.
    Buffers are allocated once (static under hipGraph) and refilled by compute()."""

    def __init__(self, batch, npoint, nsample, device):
        e = npoint * nsample
        self.shape = (batch, npoint, nsample)
        self.cidx = torch.zeros(batch, e, dtype=torch.int32, device=device)
        self.centre_of = torch.zeros(batch, e, dtype=torch.int32, device=device)
        self.mult = torch.zeros(batch, e, dtype=torch.float32, device=device)
        self.seg_off = torch.zeros(batch, npoint + 1, dtype=torch.int32, device=device)
        self.n_act = torch.zeros(batch, dtype=torch.int32, device=device)

    def compute(self, idx):
        assert tuple(idx.shape) == self.shape and idx.dtype == torch.int32 and idx.is_contiguous()
        b, m, ns = self.shape
        with torch.cuda.device(idx.device):
            _lib.call("sig3d_compact_neighbour_lists", b, m, ns, _lib.ptr(idx), _lib.ptr(self.cidx),
                      _lib.ptr(self.centre_of), _lib.ptr(self.mult), _lib.ptr(self.seg_off), _lib.ptr(self.n_act),
                      _lib.stream_ptr(idx.device))
        return self

    def tensors(self):
        return self.cidx, self.centre_of, self.mult, self.seg_off, self.n_act


def compact_lists(idx):
    return CompactLists(idx.shape[0], idx.shape[1], idx.shape[2], idx.device).compute(idx.contiguous())


# Module constants below are not environment switches: each alternative lost its A/B (DESIGN.md sections 4f-4i) and stays
# only where a test compares the two forms (monkeypatched there).
# the layers' weight gradients: 1 = the k-streaming split product (sig3d_mlp_layer_dw_stream) on compact levels and
# dense rows up to DW_STREAM_MAX_E positions, 0 = mlp_dw_kernel everywhere, 3 = streaming on every level
DW_STREAM = 1
DW_FOLD_ONCE = True      # one slab fold per SharedMLP stack (sig3d_sum_slabs_multi) instead of one per layer
DW_DX_ONE = True         # a compact level's layer: weight gradient + input gradient as two workgroup ranges of ONE launch
DW_REGROUP = True        # the gathered first layer of a compact level re-materialises its operand for the streaming dW
# dense rows longer than this keep mlp_dw_kernel (it was tuned on the 131 072-position rows of a dense SA1: the dense
# variant of the bench is 0.15 ms slower with the streaming product there)
DW_STREAM_MAX_E = 16384
# SIG3D_COMPACT=0 keeps every set-abstraction level dense
COMPACT = os.environ.get("SIG3D_COMPACT", "1") != "0"
# levels with at least this many (dense) positions run compact; the others keep the library-GEMM hybrid
COMPACT_MIN_POSITIONS = MIN_POSITIONS


class _QueryGroupCompact(torch.autograd.Function):
    """QueryAndGroup's grouped tensor (pointnet2_utils.py:348-359) for the DISTINCT neighbours only:
    out (B, 3+C, npoint, nsample) where only the first n_act[b] positions of every row are written."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, features, cidx, centre_of, n_act, nsample, radius, use_xyz, normalize_xyz):
        dev = _lib.require_device(xyz, new_xyz, features, cidx)
        b, n, _ = xyz.shape
        m = new_xyz.shape[1]
        c = 0 if features is None else features.shape[1]
        c_total = (3 if use_xyz else 0) + c
        out = torch.empty((b, c_total, m, nsample), dtype=torch.float32, device=dev)
        wide = c >= 32 and c % 4 == 0
        feat_pm = None
        with torch.cuda.device(dev):
            if wide:
                feat_pm = torch.empty((b, n, c), dtype=torch.float32, device=dev)
                _lib.call("sig3d_transpose_cn", b, c, n, _lib.ptr(features), _lib.ptr(feat_pm), _lib.stream_ptr(dev))
            _lib.call("sig3d_query_group_compact", b, n, m, c, c, nsample, int(use_xyz), int(normalize_xyz),
                      ctypes.c_float(radius), _lib.ptr(xyz), _lib.ptr(new_xyz), _lib.ptr(features),
                      _lib.ptr(feat_pm), _lib.ptr(cidx), _lib.ptr(centre_of), _lib.ptr(n_act), _lib.ptr(out),
                      _lib.stream_ptr(dev))
        ctx.save_for_backward(cidx, n_act)
        ctx.dims = (b, n, m, c, nsample, c_total, 3 if use_xyz else 0, wide)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        cidx, n_act = ctx.saved_tensors
        b, n, m, c, nsample, c_total, c_off, wide = ctx.dims
        grad_features = None
        if c > 0 and ctx.needs_input_grad[2]:
            grad_out = grad_out.contiguous()
            dev = grad_out.device
            grad_features = torch.empty((b, c, n), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                if wide:
                    grad_pm = torch.empty((b, n, c), dtype=torch.float32, device=dev)
                    _lib.call("sig3d_query_group_compact_grad", b, n, m, c, c, nsample, c_total, c_off,
                              _lib.ptr(grad_out), _lib.ptr(cidx), _lib.ptr(n_act), 1, _lib.ptr(grad_pm),
                              _lib.stream_ptr(dev))
                    _lib.call("sig3d_transpose_cn", b, n, c, _lib.ptr(grad_pm), _lib.ptr(grad_features),
                              _lib.stream_ptr(dev))
                else:
                    _lib.call("sig3d_query_group_compact_grad", b, n, m, c, c, nsample, c_total, c_off,
                              _lib.ptr(grad_out), _lib.ptr(cidx), _lib.ptr(n_act), 0, _lib.ptr(grad_features),
                              _lib.stream_ptr(dev))
        return None, None, grad_features, None, None, None, None, None, None, None


def _aff_rows(t):
    return t[0], t[1], t[2], t[3]  # scale, shift, mean, invstd


class _FusedMLPMax(torch.autograd.Function):
    """library_gemm=False: every layer is sig3d_mlp_layer_fwd (MFMA GEMM with BatchNorm+ReLU on operand
    load and statistics in the epilogue).  library_gemm=True (small levels): the 1x1 convolutions and
    the input gradients are library GEMMs (torch.bmm), BatchNorm statistics / apply / pooling and the
    whole BatchNorm+ReLU backward stay on the kernels of csrc/shared_mlp.hip -- no MIOpen BatchNorm, no
    separate ReLU / threshold / max-reduce / scatter launches."""

    @staticmethod
    def forward(ctx, x, layers, library_gemm, compact, gather, want_pm, x_is_pm, first, *flat):
        # first: None, or (new_xyz, idx, nsample, radius, normalize_xyz, cin): x is then the raw scan POINT-MAJOR
        # (B, N, cpt) = [x y z f0 ...] and the first layer forms its 3 + c <= 8 channel column in registers
        # (csrc/sa_first.hip: SA1) -- no grouped tensor in either direction, nothing differentiable below it
        # want_pm: also return the pooled features POINT-MAJOR (B, npoint, C) -- written by the pooling kernel
        # itself (sig3d_bn_relu_maxpool_pm), for the next level's gathers / the Q-Former's scene tokens
        # x_is_pm (gather mode): x is already the point-major feature copy (B, N, C) of the level below
        # flat = (W_1, gamma_1, beta_1, W_2, gamma_2, beta_2, ...) so that autograd tracks them
        # compact: None, or CompactLists.tensors(): x then holds the DISTINCT neighbours only (first n_act[b]
        # positions of every row) and every kernel below runs in compact mode (csrc/compact.hip)
        # gather: None, or (xyz, new_xyz, idx, nsample, radius, normalize_xyz): x is then the level's INPUT
        # features (B, C, N) and the first layer gathers its operand on load (no grouped
        # (B, 3 + C, npoint, nsample) tensor in either direction); idx = ball-query lists, or the compact lists
        dev = x.device
        ctx.set_materialize_grads(False)   # an unused output (channel-major OR point-major) arrives as None, not zeros
        if compact is not None:
            assert not library_gemm
            c_cidx, c_cent, c_mult, c_seg, c_nact = compact
        x = x.contiguous()
        stream = _lib.stream_ptr(dev)
        feat_pm = None
        if first is not None:
            assert not library_gemm and gather is None
            f_new_xyz, f_idx, s, f_radius, f_norm, f_cin = first
            b, n_src, cpt = x.shape
            p = f_new_xyz.shape[1]
            c0 = f_cin
        elif gather is not None:
            assert not library_gemm
            g_xyz, g_new_xyz, g_idx, s, g_radius, g_norm = gather
            p = g_new_xyz.shape[1]
            if x_is_pm:
                b, n_src, c0 = x.shape
                feat_pm = x
            else:
                b, c0, n_src = x.shape
                feat_pm = torch.empty((b, n_src, c0), dtype=torch.float32, device=dev)
                with torch.cuda.device(dev):
                    _lib.call("sig3d_transpose_cn", b, c0, n_src, _lib.ptr(x), _lib.ptr(feat_pm), stream)
        else:
            b, c0, p, s = x.shape
        e = p * s
        ys, affs, ws = [], [], []
        cur, ps, pb = x, None, None
        cmax = max(conv.out_channels for conv, _ in layers)
        with torch.cuda.device(dev):
            # all statistic accumulators of the stack are zeroed by ONE fill (accumulate = 1 below)
            st_all = scratch.zeros((len(layers), 2, cmax), torch.float64, dev)
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
                elif first is not None and i == 0:
                    y = torch.empty((b, cout, p, s), dtype=torch.float32, device=dev)
                    _lib.call("sig3d_sa_first_layer_fwd", b, n_src, p, s, cpt, f_cin, cout, int(f_norm),
                              ctypes.c_float(f_radius), _lib.ptr(x), _lib.ptr(f_new_xyz), _lib.ptr(f_idx),
                              _lib.ptr(c_cent if compact is not None else None),
                              _lib.ptr(c_nact if compact is not None else None),
                              _lib.ptr(c_mult if compact is not None else None), _lib.ptr(w), _lib.ptr(y),
                              _lib.ptr(st[0]), _lib.ptr(st[1]), 1, stream)
                elif gather is not None and i == 0:
                    y = torch.empty((b, cout, p, s), dtype=torch.float32, device=dev)
                    _lib.call("sig3d_mlp_layer0_gather_fwd", b, n_src, p, s, c0, cout, int(g_norm),
                              ctypes.c_float(g_radius), _lib.ptr(g_xyz), _lib.ptr(g_new_xyz), _lib.ptr(feat_pm),
                              _lib.ptr(g_idx), _lib.ptr(w), _lib.ptr(y), _lib.ptr(st[0]), _lib.ptr(st[1]), 1,
                              _lib.ptr(c_cent if compact is not None else None),
                              _lib.ptr(c_nact if compact is not None else None),
                              _lib.ptr(c_mult if compact is not None else None), stream)
                elif compact is not None:
                    y = torch.empty((b, cout, p, s), dtype=torch.float32, device=dev)
                    _lib.call("sig3d_mlp_layer_fwd_compact", b, cin, cout, e, _lib.ptr(cur), _lib.ptr(w),
                              _lib.ptr(ps), _lib.ptr(pb), _lib.ptr(y), _lib.ptr(st[0]), _lib.ptr(st[1]),
                              1, _lib.ptr(c_nact), _lib.ptr(c_mult), stream)
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
            out_pm = None
            if want_pm:
                out_pm = torch.empty((b, p, c_last), dtype=torch.float32, device=dev)
                _lib.call("sig3d_bn_relu_maxpool_pm", b, c_last, p, s, e, _lib.ptr(cur), _lib.ptr(ps), _lib.ptr(pb),
                          _lib.ptr(c_seg if compact is not None else None), _lib.ptr(out), _lib.ptr(arg),
                          _lib.ptr(out_pm), stream)
            elif compact is not None:
                _lib.call("sig3d_bn_relu_maxpool_compact", b, c_last, p, e, _lib.ptr(cur), _lib.ptr(ps),
                          _lib.ptr(pb), _lib.ptr(c_seg), _lib.ptr(out), _lib.ptr(arg), stream)
            else:
                _lib.call("sig3d_bn_relu_maxpool", b, c_last, p, s, _lib.ptr(cur), _lib.ptr(ps),
                          _lib.ptr(pb), _lib.ptr(out), _lib.ptr(arg), stream)
        # gather mode keeps the point-major feature copy (B, N, C) for the backward pass, not a grouped tensor
        ctx.save_for_backward(feat_pm if gather is not None else x, arg, *ys, *affs, *ws)
        ctx.compact = compact
        ctx.gather = gather
        ctx.first = first
        ctx.nl = len(layers)
        ctx.dims = (b, p, s)
        ctx.library_gemm = library_gemm
        ctx.x_is_pm = bool(x_is_pm)
        return out, out_pm

    @staticmethod
    def backward(ctx, grad_out, grad_out_pm=None):
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
        top_from_pm = None
        if grad_out_pm is not None and grad_out is None:
            # the point-major output's gradient (rows scattered by the level above / the Q-Former's token gradient):
            # turned to the channel-major rows the BatchNorm backward walks INSIDE the top layer's statistics pass
            # (sig3d_bn_relu_bwd_top_from_pm, below) -- no transpose launch
            top_from_pm = grad_out_pm.contiguous()
            grad_out = torch.empty((b, top_from_pm.shape[2], p), dtype=torch.float32, device=dev)
        elif grad_out_pm is not None:   # both layouts were consumed: one transpose, one add
            gp = grad_out_pm.contiguous()
            g_cm = torch.empty((b, gp.shape[2], p), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.call("sig3d_transpose_cn", b, p, gp.shape[2], _lib.ptr(gp), _lib.ptr(g_cm), stream)
            grad_out = grad_out + g_cm
        grad_out = grad_out.contiguous()
        grads = [None] * (3 * nl)
        grad_x = None
        compact = ctx.compact
        gather = ctx.gather
        if compact is not None:
            c_cidx, c_cent, c_mult, c_seg, c_nact = compact
        regroup = None
        if gather is not None:
            g_xyz, g_new_xyz, g_idx, _, g_radius, g_norm = gather
            n_src, c_src = x.shape[1], x.shape[2]          # x is the point-major feature copy here
        with torch.cuda.device(dev):
            if gather is not None and compact is None:
                # Dense lists: the backward products of the first layer run FASTER on a stored operand than with
                # the gather / scatter fused in (SA2 at B = 8, 262 144 positions x 131 channels: weight gradient
                # 148 us stored vs 319 us gathering, input gradient 243 us (product + merged-run scatter kernel)
                # vs 521 us with the atomics in the product's epilogue), so the grouped tensor is RE-MATERIALISED
                # here (30 us) -- recompute-in-backward: it is never kept between the passes.  Compact lists
                # (a few thousand positions, launch-bound) take the fused kernels.
                regroup = torch.empty((b, c_src + 3, p, s), dtype=torch.float32, device=dev)
                _lib.call("sig3d_query_group_fused_pm", b, n_src, p, c_src, c_src, s, 1, int(g_norm),
                          ctypes.c_float(g_radius), _lib.ptr(g_xyz), _lib.ptr(g_new_xyz), _lib.ptr(x),
                          _lib.ptr(g_idx), _lib.ptr(regroup), stream)
            dA = None
            # one fill each for every BatchNorm-gradient accumulator and every dW of the stack
            cmax = max(w.shape[0] for w in ws)
            n_dw = sum(w.numel() for w in ws)
            # (slices of the step's zeroed region when a step is open -- scratch.py; the dW slices become `.grad`s,
            # so they only come from there when the optimizer consumes the gradients inside the step)
            if scratch.STEP_ZEROS.grads_ok:
                zeros = scratch.zeros(nl * 2 * cmax * 8 + n_dw * 4, torch.uint8, dev)
                sums_all = zeros[:nl * 2 * cmax * 8].view(torch.float64).view(nl, 2, cmax)
                dw_all = zeros[nl * 2 * cmax * 8:].view(torch.float32)
            else:
                sums_all = scratch.zeros((nl, 2, cmax), torch.float64, dev)
                dw_all = torch.zeros(n_dw, dtype=torch.float32, device=dev)
            dw_off = 0
            folds = []           # (dW, work, n, slab stride, slabs) of the streaming weight gradients: ONE fold per stack
            dw_entry = "sig3d_mlp_layer_dw_stream_nofold" if DW_FOLD_ONCE else "sig3d_mlp_layer_dw_stream"
            for k in range(nl - 1, -1, -1):
                cout, cin = ws[k].shape
                scale, shift, mean, invstd = _aff_rows(affs[k])
                sums = sums_all[k]
                dY = torch.empty_like(ys[k])
                acc = 1          # the accumulators were zeroed by the one fill above
                if k == nl - 1 and top_from_pm is not None:
                    _lib.call("sig3d_bn_relu_bwd_top_from_pm", b, cout, p, s, e, _lib.ptr(top_from_pm), _lib.ptr(arg),
                              _lib.ptr(ys[k]), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(mean), _lib.ptr(invstd),
                              _lib.ptr(c_seg if compact is not None else None), _lib.ptr(grad_out), _lib.ptr(sums[0]),
                              _lib.ptr(sums[1]), 1, stream)
                    acc = 2      # statistics complete: the launch below skips its top-statistics pass
                if compact is not None:
                    top = k == nl - 1
                    _lib.call("sig3d_bn_relu_bwd_compact", b, cout, e, p, _lib.ptr(None if top else dA),
                              _lib.ptr(grad_out if top else None), _lib.ptr(arg if top else None), _lib.ptr(ys[k]),
                              _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(sums[0]),
                              _lib.ptr(sums[1]), _lib.ptr(dY), acc, _lib.ptr(c_nact), _lib.ptr(c_mult),
                              _lib.ptr(c_cent), _lib.ptr(c_seg), stream)
                elif k == nl - 1:
                    _lib.call("sig3d_bn_relu_bwd", b, cout, e, s, _lib.ptr(None), _lib.ptr(grad_out),
                              _lib.ptr(arg), _lib.ptr(ys[k]), _lib.ptr(scale), _lib.ptr(shift),
                              _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(sums[0]), _lib.ptr(sums[1]),
                              _lib.ptr(dY), acc, stream)
                else:
                    _lib.call("sig3d_bn_relu_bwd", b, cout, e, s, _lib.ptr(dA), _lib.ptr(None),
                              _lib.ptr(None), _lib.ptr(ys[k]), _lib.ptr(scale), _lib.ptr(shift),
                              _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(sums[0]), _lib.ptr(sums[1]),
                              _lib.ptr(dY), 1, stream)
                prev = ys[k - 1] if k > 0 else (regroup if regroup is not None else x)
                pps = affs[k - 1][0] if k > 0 else None
                ppb = affs[k - 1][1] if k > 0 else None
                dW = dw_all[dw_off:dw_off + cout * cin].view(cout, cin)
                dw_off += cout * cin
                dA_next = None       # set by a launch that computes the input gradient along with the weight gradient
                if ctx.first is not None and k == 0:
                    f_new_xyz, f_idx, _, f_radius, f_norm, f_cin = ctx.first
                    _lib.call("sig3d_sa_first_layer_dw", b, x.shape[1], p, s, x.shape[2], f_cin, cout, int(f_norm),
                              ctypes.c_float(f_radius), _lib.ptr(x), _lib.ptr(f_new_xyz), _lib.ptr(f_idx),
                              _lib.ptr(c_cent if compact is not None else None),
                              _lib.ptr(c_nact if compact is not None else None), _lib.ptr(dY), _lib.ptr(dW), 1, stream)
                elif gather is not None and k == 0 and regroup is None and compact is not None and DW_REGROUP \
                        and DW_STREAM and e % 4 == 0 and (cout * cin) % 4 == 0:
                    # compact lists: the grouped tensor of the DISTINCT neighbours is a few MB -- re-materialised here
                    # (point-major rows in, one launch) and streamed through the split product; the gathering weight-
                    # gradient kernel walks index -> row round trips (42 / 86 us at 4 % / 16 % distinct neighbours)
                    rg = torch.empty((b, c_src + 3, e), dtype=torch.float32, device=dev)
                    _lib.call("sig3d_query_group_compact", b, n_src, p, c_src, c_src, s, 1, int(g_norm),
                              ctypes.c_float(g_radius), _lib.ptr(g_xyz), _lib.ptr(g_new_xyz), _lib.ptr(None), _lib.ptr(x),
                              _lib.ptr(c_cidx), _lib.ptr(c_cent), _lib.ptr(c_nact), _lib.ptr(rg), stream)
                    n_work = int(_lib.load().sig3d_mlp_layer_dw_stream_work_floats(b, cin, cout, e))
                    work = torch.empty(max(n_work, 4), dtype=torch.float32, device=dev)
                    _lib.call(dw_entry, b, cin, cout, e, _lib.ptr(dY), _lib.ptr(rg), _lib.ptr(None),
                              _lib.ptr(None), _lib.ptr(c_nact), _lib.ptr(dW), _lib.ptr(work), stream)
                    if DW_FOLD_ONCE:
                        slab = (cout * cin + 3) // 4 * 4
                        folds.append((dW, work, cout * cin, slab, n_work // slab))
                elif gather is not None and k == 0 and regroup is None:
                    _lib.call("sig3d_mlp_layer0_gather_dw", b, n_src, p, s, c_src, cout, int(g_norm),
                              ctypes.c_float(g_radius), _lib.ptr(g_xyz), _lib.ptr(g_new_xyz), _lib.ptr(x),
                              _lib.ptr(g_idx), _lib.ptr(dY), _lib.ptr(dW), 1,
                              _lib.ptr(c_cent if compact is not None else None),
                              _lib.ptr(c_nact if compact is not None else None), stream)
                elif DW_STREAM and e % 4 == 0 and (cout * cin) % 4 == 0 and \
                        (DW_STREAM == 3 or compact is not None or e <= DW_STREAM_MAX_E):
                    # k-streaming split product on the f32 matrix cores (gemm16_core.h, weight-gradient form): both
                    # operands read along their rows, slabs folded in a fixed order (1.5-2.5 x sig3d_mlp_layer_dw)
                    n_work = int(_lib.load().sig3d_mlp_layer_dw_stream_work_floats(b, cin, cout, e))
                    work = torch.empty(max(n_work, 4), dtype=torch.float32, device=dev)
                    if DW_DX_ONE and DW_FOLD_ONCE and compact is not None and k > 0 and not ctx.library_gemm:
                        # the layer's two products over dY as workgroup ranges of one launch (they share only what they read)
                        dA_next = torch.empty((b, cin, p, s), dtype=torch.float32, device=dev)
                        _lib.call("sig3d_mlp_layer_dw_dx", b, cin, cout, e, _lib.ptr(dY), _lib.ptr(prev), _lib.ptr(pps),
                                  _lib.ptr(ppb), _lib.ptr(c_nact), _lib.ptr(ws[k]), _lib.ptr(dW), _lib.ptr(work),
                                  _lib.ptr(dA_next), stream)
                    else:
                        _lib.call(dw_entry, b, cin, cout, e, _lib.ptr(dY), _lib.ptr(prev), _lib.ptr(pps),
                                  _lib.ptr(ppb), _lib.ptr(c_nact if compact is not None else None), _lib.ptr(dW),
                                  _lib.ptr(work), stream)
                    if DW_FOLD_ONCE:
                        slab = (cout * cin + 3) // 4 * 4
                        folds.append((dW, work, cout * cin, slab, n_work // slab))
                elif compact is not None:
                    _lib.call("sig3d_mlp_layer_dw_compact", b, cin, cout, e, _lib.ptr(dY), _lib.ptr(prev),
                              _lib.ptr(pps), _lib.ptr(ppb), _lib.ptr(dW), 1, _lib.ptr(c_nact), stream)
                else:
                    _lib.call("sig3d_mlp_layer_dw", b, cin, cout, e, _lib.ptr(dY), _lib.ptr(prev),
                              _lib.ptr(pps), _lib.ptr(ppb), _lib.ptr(dW), 1, stream)
                grads[3 * k] = dW.view(cout, cin, 1, 1)
                if k == 0 and ctx.first is not None:
                    pass          # the raw scan is not differentiable
                elif k == 0 and regroup is not None:
                    if ctx.needs_input_grad[0]:
                        dA = torch.empty((b, cin, p, s), dtype=torch.float32, device=dev)
                        _lib.call("sig3d_mlp_layer_dx", b, cin, cout, e, _lib.ptr(dY), _lib.ptr(ws[0]), _lib.ptr(dA),
                                  _lib.ptr(None), stream)
                        grad_pm = torch.empty((b, n_src, c_src), dtype=torch.float32, device=dev)
                        _lib.call("sig3d_query_group_fused_grad_pm", b, n_src, p, c_src, c_src, s, cin, 3,
                                  _lib.ptr(dA), _lib.ptr(g_idx), _lib.ptr(grad_pm), stream)
                        if ctx.x_is_pm:
                            grad_x = grad_pm      # the level below takes its gradient point-major
                        else:
                            grad_x = torch.empty((b, c_src, n_src), dtype=torch.float32, device=dev)
                            _lib.call("sig3d_transpose_cn", b, n_src, c_src, _lib.ptr(grad_pm), _lib.ptr(grad_x), stream)
                elif k == 0 and gather is not None:
                    if ctx.needs_input_grad[0]:
                        # W^T dY added straight into the point-major feature gradient at the neighbours' rows
                        grad_pm = scratch.zeros((b, n_src, c_src), torch.float32, dev)
                        _lib.call("sig3d_mlp_layer0_scatter_dx_w", b, n_src, p, s, c_src, cout, _lib.ptr(g_idx),
                                  _lib.ptr(dY), _lib.ptr(ws[0]), _lib.ptr(grad_pm),
                                  _lib.ptr(c_nact if compact is not None else None), stream)
                        if ctx.x_is_pm:
                            grad_x = grad_pm
                        else:
                            grad_x = torch.empty((b, c_src, n_src), dtype=torch.float32, device=dev)
                            _lib.call("sig3d_transpose_cn", b, n_src, c_src, _lib.ptr(grad_pm), _lib.ptr(grad_x), stream)
                elif dA_next is not None:
                    dA = dA_next
                elif k > 0 or ctx.needs_input_grad[0]:
                    if ctx.library_gemm:
                        dA = torch.bmm(ws[k].t().unsqueeze(0).expand(b, cin, cout), dY.view(b, cout, e)).view(b, cin, p, s)
                    else:
                        # dA = W^T dY through the forward kernel, the weight staged transposed (no W^T copy)
                        dA = torch.empty((b, cin, p, s), dtype=torch.float32, device=dev)
                        _lib.call("sig3d_mlp_layer_dx", b, cin, cout, e, _lib.ptr(dY), _lib.ptr(ws[k]), _lib.ptr(dA),
                                  _lib.ptr(c_nact if compact is not None else None), stream)
                    if k == 0:
                        grad_x = dA
            if folds:            # the folds of the stack and the f64 -> f32 conversion of its BatchNorm-gradient sums: one launch
                sums32 = torch.empty(sums_all.shape, dtype=torch.float32, device=dev)
                _lib.sum_slabs_multi(dev, folds, convert=(sums_all, sums32))
            else:
                sums32 = sums_all.to(torch.float32)            # one conversion launch for the whole stack
            for k in range(nl):
                cout = ws[k].shape[0]
                grads[3 * k + 1] = sums32[k, 1, :cout]         # d gamma
                grads[3 * k + 2] = sums32[k, 0, :cout]         # d beta
        return (grad_x, None, None, None, None, None, None, None) + tuple(grads)


def _with_pm(out, out_pm):
    """The channel-major result with its point-major twin riding along as an attribute: the next level (or the
    model, for the scene tokens) picks it up from the very tensor object it is handed; any op in between creates
    a new tensor without it, and the consumer falls back to a transpose."""
    if out_pm is not None:
        out._pm = out_pm
    return out


def point_major_of(features):
    """(B, N, C) twin of channel-major features (B, C, N), if the producer wrote one (see _with_pm)."""
    pm = getattr(features, "_pm", None)
    if pm is not None and pm.dim() == 3 and features.dim() == 3 and \
            (pm.shape[0], pm.shape[2], pm.shape[1]) == tuple(features.shape) and pm.is_contiguous():
        return pm
    return None


def _fused_mlp_max_eval(layers, x, compact=None, gather=None, want_pm=False, x_is_pm=False, first=None):
    """Inference: BatchNorm2d.eval() is the affine map scale = gamma / sqrt(running_var + eps),
    shift = beta - running_mean * scale, so a layer is one sig3d_mlp_layer_fwd (previous layer's
    BN+ReLU on operand load, no statistics) and the stack ends in sig3d_bn_relu_maxpool.
    compact: CompactLists.tensors() when x holds the distinct neighbours only."""
    dev = x.device
    x = x.contiguous()
    stream = _lib.stream_ptr(dev)
    if first is not None:    # x: the raw scan point-major (B, N, cpt); first layer forms its column in registers
        f_new_xyz, f_idx, s, f_radius, f_norm, f_cin = first
        b, n_src, cpt = x.shape
        p = f_new_xyz.shape[1]
    elif gather is not None:   # x: the level's input features (B, C, N); first layer gathers on load
        g_xyz, g_new_xyz, g_idx, s, g_radius, g_norm = gather
        p = g_new_xyz.shape[1]
        if x_is_pm:
            b, n_src, c0 = x.shape
            feat_pm = x
        else:
            b, c0, n_src = x.shape
            feat_pm = torch.empty((b, n_src, c0), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.call("sig3d_transpose_cn", b, c0, n_src, _lib.ptr(x), _lib.ptr(feat_pm), stream)
    else:
        b, _, p, s = x.shape
    e = p * s
    cur, ps, pb = x, None, None
    with torch.no_grad(), torch.cuda.device(dev):
        for li, (conv, bn) in enumerate(layers):
            w = conv.weight.reshape(conv.out_channels, conv.in_channels).contiguous()
            cout, cin = w.shape
            y = torch.empty((b, cout, p, s), dtype=torch.float32, device=dev)
            if first is not None and li == 0:
                _lib.call("sig3d_sa_first_layer_fwd", b, n_src, p, s, cpt, f_cin, cout, int(f_norm), ctypes.c_float(f_radius),
                          _lib.ptr(x), _lib.ptr(f_new_xyz), _lib.ptr(f_idx),
                          _lib.ptr(compact[1] if compact is not None else None),
                          _lib.ptr(compact[4] if compact is not None else None), _lib.ptr(None), _lib.ptr(w), _lib.ptr(y),
                          _lib.ptr(None), _lib.ptr(None), 0, stream)
            elif gather is not None and li == 0:
                _lib.call("sig3d_mlp_layer0_gather_fwd", b, n_src, p, s, c0, cout, int(g_norm), ctypes.c_float(g_radius),
                          _lib.ptr(g_xyz), _lib.ptr(g_new_xyz), _lib.ptr(feat_pm), _lib.ptr(g_idx), _lib.ptr(w),
                          _lib.ptr(y), _lib.ptr(None), _lib.ptr(None), 0,
                          _lib.ptr(compact[1] if compact is not None else None),
                          _lib.ptr(compact[4] if compact is not None else None), _lib.ptr(None), stream)
            elif compact is not None:
                _lib.call("sig3d_mlp_layer_fwd_compact", b, cin, cout, e, _lib.ptr(cur), _lib.ptr(w), _lib.ptr(ps),
                          _lib.ptr(pb), _lib.ptr(y), _lib.ptr(None), _lib.ptr(None), 0, _lib.ptr(compact[4]),
                          _lib.ptr(None), stream)
            else:
                _lib.call("sig3d_mlp_layer_fwd", b, cin, cout, e, _lib.ptr(cur), _lib.ptr(w), _lib.ptr(ps),
                          _lib.ptr(pb), _lib.ptr(y), _lib.ptr(None), _lib.ptr(None), 0, stream)
            scale = (bn.weight * torch.rsqrt(bn.running_var + bn.eps)).contiguous()
            shift = (bn.bias - bn.running_mean * scale).contiguous()
            cur, ps, pb = y, scale, shift
        c_last = cur.shape[1]
        out = torch.empty((b, c_last, p), dtype=torch.float32, device=dev)
        arg = torch.empty((b, c_last, p), dtype=torch.int32, device=dev)
        out_pm = None
        if want_pm:
            out_pm = torch.empty((b, p, c_last), dtype=torch.float32, device=dev)
            _lib.call("sig3d_bn_relu_maxpool_pm", b, c_last, p, s, e, _lib.ptr(cur), _lib.ptr(ps), _lib.ptr(pb),
                      _lib.ptr(compact[3] if compact is not None else None), _lib.ptr(out), _lib.ptr(arg),
                      _lib.ptr(out_pm), stream)
        elif compact is not None:
            _lib.call("sig3d_bn_relu_maxpool_compact", b, c_last, p, e, _lib.ptr(cur), _lib.ptr(ps), _lib.ptr(pb),
                      _lib.ptr(compact[3]), _lib.ptr(out), _lib.ptr(arg), stream)
        else:
            _lib.call("sig3d_bn_relu_maxpool", b, c_last, p, s, _lib.ptr(cur), _lib.ptr(ps), _lib.ptr(pb),
                      _lib.ptr(out), _lib.ptr(arg), stream)
    return _with_pm(out, out_pm)


def fused_mlp_max(mlp, x, library_gemm=None, want_pm=False):
    """max over nsample of SharedMLP(x): x (B,C,npoint,nsample) -> (B,C_out,npoint).
    library_gemm=None: decided by the position count (MIN_POSITIONS).
    want_pm: the result carries its point-major twin (point_major_of)."""
    layers = _layers(mlp)
    if not mlp.training:
        return _fused_mlp_max_eval(layers, x, want_pm=want_pm)
    flat = []
    for conv, bn in layers:
        flat += [conv.weight, bn.weight, bn.bias]
    if library_gemm is None:
        library_gemm = x.shape[0] * x.shape[2] * x.shape[3] < MIN_POSITIONS
    return _with_pm(*_FusedMLPMax.apply(x, layers, bool(library_gemm), None, None, bool(want_pm), False, None, *flat))


GATHER_L0 = True     # False: always store the grouped tensor (tests compare the two forms)


def gather_applies(features, use_xyz):
    """The gathering first layer serves levels with a point-major feature copy: >= 32 channels in multiples of 32
    (SA2-4: 128 / 256 / 256), xyz concatenated.  SA1 (3 input channels, a 29 MB grouped tensor) keeps the stored form."""
    return (GATHER_L0 and features is not None and use_xyz and features.dim() == 3 and features.shape[1] >= 32
            and features.shape[1] % 32 == 0 and features.is_cuda and features.dtype == torch.float32)


FIRST_L0 = True      # False: SA1 keeps the stored grouped tensor (tests compare the two forms)


def attach_scan(features, point_clouds):
    """features: the (B, C, N) channel-major VIEW of point_clouds (B, N, 3 + C)[..., 3:] (no copy made);
    the raw point-major scan rides along so that a first SA level can read whole rows (first_layer_scan)."""
    if point_clouds.is_contiguous() and point_clouds.dtype == torch.float32:
        features._points_pm = point_clouds
    return features


# dense (full) lists: the stored grouped tensor + MFMA layer is faster (13 + 76 us against 114 us at SA1, B = 8:
# the scan kernel writes its 64 output rows as 64 separate 256-byte stores per wave); 1 forces the scan path anyway
FIRST_L0_DENSE = False


def first_layer_scan(mlp, xyz, features, use_xyz, dense=False):
    """The point-major scan (B, N, 3 + C) behind `features` when the first SharedMLP layer can form its column from it
    in registers (csrc/sa_first.hip): 3 + C <= 8 input channels, 64 | output channels, xyz concatenated, nothing
    differentiable below -- else None.  Used on compact lists (dense ones: FIRST_L0_DENSE)."""
    if not (FIRST_L0 and features is not None and use_xyz and features.dim() == 3 and features.is_cuda):
        return None
    if dense and not FIRST_L0_DENSE:
        return None
    pts = getattr(features, "_points_pm", None)
    layers = _layers(mlp)
    if pts is None or layers is None:
        return None
    b, c, n = features.shape
    conv = layers[0][0]
    if not (pts.dim() == 3 and tuple(pts.shape[:2]) == (b, n) and pts.shape[2] == 3 + c and pts.is_contiguous()
            and pts.dtype == torch.float32 and pts.device == features.device and 3 + c <= 8
            and conv.in_channels == 3 + c and conv.out_channels % 64 == 0):
        return None
    if torch.is_grad_enabled() and (features.requires_grad or pts.requires_grad or xyz.requires_grad):
        return None
    return pts


def fused_sa_compact(mlp, xyz, new_xyz, features, compact, nsample, radius, use_xyz, normalize_xyz, want_pm=False):
    """One set-abstraction level over the distinct neighbours only (training mode, MFMA path):
    grouped tensor -> SharedMLP -> max over the neighbourhood, same result as the dense path up to
    floating-point summation order.  compact: CompactLists of this level's ball-query result."""
    layers = _layers(mlp)
    cidx, centre_of, mult, seg_off, n_act = compact.tensors()
    lists = (cidx, centre_of, mult, seg_off, n_act)
    gather, x_is_pm, first = None, False, None
    pts = first_layer_scan(mlp, xyz, features, use_xyz)
    feats = None if (features is None or pts is not None) else features.contiguous()
    if pts is not None:
        first = (new_xyz.contiguous(), cidx, int(nsample), float(radius), bool(normalize_xyz), pts.shape[2])
        x = pts
    elif gather_applies(feats, use_xyz):
        gather = (xyz.contiguous(), new_xyz.contiguous(), cidx, int(nsample), float(radius), bool(normalize_xyz))
        x = point_major_of(features)       # written by the level below's pooling kernel: no transpose launch
        x_is_pm = x is not None
        if x is None:
            x = feats
    else:
        x = _QueryGroupCompact.apply(xyz.contiguous(), new_xyz.contiguous(), feats, cidx, centre_of, n_act, int(nsample),
                                     float(radius), bool(use_xyz) or features is None, bool(normalize_xyz))
    if not mlp.training:
        return _fused_mlp_max_eval(layers, x, lists, gather, want_pm, x_is_pm, first)
    flat = []
    for conv, bn in layers:
        flat += [conv.weight, bn.weight, bn.bias]
    return _with_pm(*_FusedMLPMax.apply(x, layers, False, lists, gather, bool(want_pm), x_is_pm, first, *flat))


def dense_gather_applies(mlp, xyz, features, npoint, nsample, use_xyz):
    """Dense (full-list) levels on the MFMA path whose first layer can gather on load (wide feature rows), or
    form its column from the raw scan (first_layer_scan)."""
    if not xyz.is_cuda or _layers(mlp) is None:
        return False
    if not gather_applies(features, use_xyz) and first_layer_scan(mlp, xyz, features, use_xyz, dense=True) is None:
        return False
    if xyz.shape[0] * npoint * nsample < MIN_POSITIONS:
        return False          # small levels: library-GEMM hybrid on the stored tensor
    if torch.is_grad_enabled() and xyz.requires_grad:
        return False
    if not mlp.training:
        wants_grad = torch.is_grad_enabled() and (features.requires_grad or any(p.requires_grad for p in mlp.parameters()))
        if wants_grad:
            return False
    return True


def fused_sa_dense(mlp, xyz, new_xyz, features, ball_idx, nsample, radius, normalize_xyz, want_pm=False):
    """One dense set-abstraction level without the grouped tensor: QueryAndGroup + SharedMLP + max-pool
    (pointnet2_modules.py:242-262) with the first layer gathering its operand from (xyz, features) through the
    ball-query lists, the weight gradient gathering it again, and the input gradient scattered by the dX product."""
    layers = _layers(mlp)
    idx2 = ball_idx.contiguous().view(ball_idx.shape[0], -1)
    gather, first, x_is_pm = None, None, False
    pts = first_layer_scan(mlp, xyz, features, True, dense=True)
    if pts is not None:
        first = (new_xyz.contiguous(), idx2, int(nsample), float(radius), bool(normalize_xyz), pts.shape[2])
        x = pts
    else:
        gather = (xyz.contiguous(), new_xyz.contiguous(), idx2, int(nsample), float(radius), bool(normalize_xyz))
        x = point_major_of(features)
        x_is_pm = x is not None
        if x is None:
            x = features.contiguous()
    if not mlp.training:
        return _fused_mlp_max_eval(layers, x, None, gather, want_pm, x_is_pm, first)
    flat = []
    for conv, bn in layers:
        flat += [conv.weight, bn.weight, bn.bias]
    return _with_pm(*_FusedMLPMax.apply(x, layers, False, None, gather, bool(want_pm), x_is_pm, first, *flat))


def compact_applies(mlp, xyz, features, npoint, nsample):
    """Compact mode serves the MFMA path (large levels) of plain max-pooled SA layers: training mode, and
    eval mode when no gradient is wanted (the fused inference path)."""
    if not (COMPACT and xyz.is_cuda and _layers(mlp) is not None):
        return False
    if not mlp.training:
        wants_grad = torch.is_grad_enabled() and ((features is not None and features.requires_grad)
                                                  or any(p.requires_grad for p in mlp.parameters()))
        if wants_grad:
            return False
    if features is not None and (not features.is_cuda or features.dtype != torch.float32):
        return False
    differentiable_xyz = torch.is_grad_enabled() and xyz.requires_grad
    return (not differentiable_xyz) and xyz.shape[0] * npoint * nsample >= COMPACT_MIN_POSITIONS

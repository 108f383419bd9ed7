"""Autograd wrappers and groupers over the HIP ops -- host-side mirror of the reference's
lib/pointnet2/pointnet2_utils.py (same public names, call signatures, return tuples).

The six autograd.Functions follow pointnet2_utils.py:51-291 one to one.  `QueryAndGroup`
additionally routes its post-ball-query work (pointnet2_utils.py:348-359: two grouping passes,
centre subtraction, optional /radius, concat) through ONE fused HIP kernel when neither xyz nor
new_xyz needs a gradient; the result is bit-identical to the unfused composition, which is kept
for the differentiable-xyz case.
"""
import ctypes

import torch
import torch.nn as nn
from torch.autograd import Function

from .. import _lib, scratch
from . import _ext


class FurthestPointSampling(Function):
    """pointnet2_utils.py:51-77"""

    @staticmethod
    def forward(ctx, xyz, npoint):
        fps_inds = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(fps_inds)
        return fps_inds

    @staticmethod
    def backward(ctx, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    """pointnet2_utils.py:83-114"""

    @staticmethod
    def forward(ctx, features, idx):
        _, C, N = features.size()
        ctx.for_backwards = (idx, C, N)
        return _ext.gather_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        grad_features = _ext.gather_points_grad(grad_out.contiguous(), idx, N)
        return grad_features, None


gather_operation = GatherOperation.apply


def gather_xyz(xyz, idx):
    """new_xyz (B,M,3) = xyz[b, idx[b,j], :] in one kernel -- what the reference spells as
    gather_operation(xyz.transpose(1,2).contiguous(), idx).transpose(1,2).contiguous()
    (pointnet2_modules.py:233-240).  Bit-identical; used when xyz needs no gradient."""
    dev = _lib.require_device(xyz, idx)
    b, n, _ = xyz.shape
    m = idx.shape[1]
    out = torch.empty((b, m, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.call("sig3d_gather_xyz", b, n, m, _lib.ptr(xyz.contiguous()), _lib.ptr(idx),
                  _lib.ptr(out), _lib.stream_ptr(dev))
    return out


class ThreeNN(Function):
    """pointnet2_utils.py:120-146 (returns sqrt of the squared distances, :140-142)"""

    @staticmethod
    def forward(ctx, unknown, known):
        dist2, idx = _ext.three_nn(unknown, known)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """pointnet2_utils.py:152-203"""

    @staticmethod
    def forward(ctx, features, idx, weight):
        B, c, m = features.size()
        ctx.three_interpolate_for_backward = (idx, weight, m)
        return _ext.three_interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, m = ctx.three_interpolate_for_backward
        grad_features = _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight, m)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    """pointnet2_utils.py:209-254"""

    @staticmethod
    def forward(ctx, features, idx):
        _, C, N = features.size()
        ctx.for_backwards = (idx, N)
        return _ext.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, N = ctx.for_backwards
        grad_features = _ext.group_points_grad(grad_out.contiguous(), idx, N)
        return grad_features, None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    """pointnet2_utils.py:260-288 -- Python order (radius, nsample, xyz, new_xyz); the native
    entry takes new_xyz first (:282)."""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        inds = _ext.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


POINT_MAJOR_MIN_CHANNELS = 32


class _QueryGroupFused(Function):
    """One-kernel version of pointnet2_utils.py:348-359 (xyz / new_xyz treated as constants)."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, features, idx, radius, use_xyz, normalize_xyz, features_pm=None):
        """features_pm: the point-major twin (B, N, C) of the features when their producer wrote one
        (fused_mlp.point_major_of); it is then THE differentiable input (features = None), and its gradient
        is returned point-major too -- no transpose launch in either direction."""
        dev = _lib.require_device(xyz, new_xyz, features, idx, features_pm)
        b, n, _ = xyz.shape
        m, nsample = idx.shape[1], idx.shape[2]
        c = features_pm.shape[2] if features_pm is not None else (0 if features is None else features.shape[1])
        c_total = (3 if use_xyz else 0) + c
        out = torch.empty((b, c_total, m, nsample), dtype=torch.float32, device=dev)
        ctx.from_pm = features_pm is not None
        with torch.cuda.device(dev):
            if c >= POINT_MAJOR_MIN_CHANNELS and c % 4 == 0:
                # wide levels: one coalesced row per neighbour from a point-major copy of the features
                # (8 MB at the bench shapes) instead of c strided 4-byte gathers
                if features_pm is not None:
                    feat_pm = features_pm.contiguous()
                else:
                    feat_pm = torch.empty((b, n, c), dtype=torch.float32, device=dev)
                    _lib.call("sig3d_transpose_cn", b, c, n, _lib.ptr(features), _lib.ptr(feat_pm),
                              _lib.stream_ptr(dev))
                _lib.call("sig3d_query_group_fused_pm", b, n, m, c, c, nsample, int(use_xyz),
                          int(normalize_xyz), ctypes.c_float(radius), _lib.ptr(xyz), _lib.ptr(new_xyz),
                          _lib.ptr(feat_pm), _lib.ptr(idx), _lib.ptr(out), _lib.stream_ptr(dev))
            else:
                _lib.call("sig3d_query_group_fused", b, n, m, c, nsample, int(use_xyz),
                          int(normalize_xyz), ctypes.c_float(radius), _lib.ptr(xyz),
                          _lib.ptr(new_xyz), _lib.ptr(features), _lib.ptr(idx), _lib.ptr(out),
                          _lib.stream_ptr(dev))
        ctx.save_for_backward(idx)
        ctx.dims = (b, n, m, c, nsample, c_total, 3 if use_xyz else 0)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        b, n, m, c, nsample, c_total, c_off = ctx.dims
        grad_features = None
        if ctx.from_pm:
            grad_pm = None
            if c > 0 and ctx.needs_input_grad[7]:
                grad_out = grad_out.contiguous()
                dev = grad_out.device
                grad_pm = scratch.zeros((b, n, c), torch.float32, dev)      # the step's one fill (scratch.py)
                with torch.cuda.device(dev):
                    _lib.call("sig3d_query_group_fused_grad_pm_z", b, n, m, c, c, nsample, c_total, c_off,
                              _lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(grad_pm), _lib.stream_ptr(dev))
            return None, None, None, None, None, None, None, grad_pm
        if c > 0 and ctx.needs_input_grad[2]:
            grad_out = grad_out.contiguous()
            dev = grad_out.device
            grad_features = torch.empty((b, c, n), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                if c >= POINT_MAJOR_MIN_CHANNELS and c % 4 == 0:
                    # every grouped element adds one contiguous row of a point-major gradient (runs of the
                    # padded first index merged in registers), then one transpose back: 3x faster than the
                    # channel-slab LDS scatter at the bench shapes
                    grad_pm = torch.empty((b, n, c), dtype=torch.float32, device=dev)
                    _lib.call("sig3d_query_group_fused_grad_pm", b, n, m, c, c, nsample, c_total, c_off,
                              _lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(grad_pm), _lib.stream_ptr(dev))
                    _lib.call("sig3d_transpose_cn", b, n, c, _lib.ptr(grad_pm), _lib.ptr(grad_features),
                              _lib.stream_ptr(dev))
                else:
                    _lib.call("sig3d_query_group_fused_grad", b, n, m, c, nsample, c_total, c_off,
                              _lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(grad_features),
                              _lib.stream_ptr(dev))
        return None, None, grad_features, None, None, None, None, None


class QueryAndGroup(nn.Module):
    """pointnet2_utils.py:294-376 (same ctor kwargs, same return convention)."""

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False,
                 sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz
        self.sample_uniformly = sample_uniformly
        self.ret_unique_cnt = ret_unique_cnt
        if self.ret_unique_cnt:
            assert self.sample_uniformly

    def forward(self, xyz, new_xyz, features=None, idx=None):
        """`idx` (optional, not in the reference signature): a precomputed ball-query result for
        exactly these (xyz, new_xyz, radius, nsample) -- see geometry.GeometryPlan."""
        if idx is None:
            idx = ball_query(self.radius, self.nsample, xyz, new_xyz)

        if self.sample_uniformly:  # :336-345, host-bound python loop kept for API parity
            unique_cnt = torch.zeros((idx.shape[0], idx.shape[1]))
            for i_batch in range(idx.shape[0]):
                for i_region in range(idx.shape[1]):
                    unique_ind = torch.unique(idx[i_batch, i_region, :])
                    num_unique = unique_ind.shape[0]
                    unique_cnt[i_batch, i_region] = num_unique
                    sample_ind = torch.randint(0, num_unique, (self.nsample - num_unique,),
                                               dtype=torch.long)
                    all_ind = torch.cat((unique_ind, unique_ind[sample_ind]))
                    idx[i_batch, i_region, :] = all_ind

        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"

        differentiable_xyz = torch.is_grad_enabled() and (xyz.requires_grad or new_xyz.requires_grad)
        # host tensors never reach the fused kernel: the unfused composition below then hits
        # _ext, which raises "CPU not supported" exactly like the reference (ball_query.cpp:27-29)
        fused_ok = xyz.is_cuda and (not differentiable_xyz) and \
            (self.use_xyz or features is not None)
        if fused_ok:
            feats = None if features is None else features.contiguous()
            # the producer of `features` may have written them point-major as well (fused_mlp.point_major_of)
            pm = getattr(features, "_pm", None) if features is not None else None
            if pm is not None and not (pm.dim() == 3 and features.dim() == 3 and pm.is_contiguous() and pm.shape[2] % 4 == 0
                                       and pm.shape[2] >= POINT_MAJOR_MIN_CHANNELS
                                       and (pm.shape[0], pm.shape[2], pm.shape[1]) == tuple(features.shape)):
                pm = None
            fused = _QueryGroupFused.apply(xyz.contiguous(), new_xyz.contiguous(), None if pm is not None else feats, idx,
                                           float(self.radius), bool(self.use_xyz) or features is None,
                                           bool(self.normalize_xyz), pm)
            new_features = fused
            grouped_xyz = None
            if self.ret_grouped_xyz:
                if self.use_xyz or features is None:
                    grouped_xyz = fused[:, :3]
                else:
                    grouped_xyz = _QueryGroupFused.apply(xyz.contiguous(), new_xyz.contiguous(),
                                                         None, idx, float(self.radius), True,
                                                         bool(self.normalize_xyz))
        else:
            xyz_trans = xyz.transpose(1, 2).contiguous()
            grouped_xyz = grouping_operation(xyz_trans, idx)  # (B, 3, npoint, nsample)
            grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
            if self.normalize_xyz:
                grouped_xyz = grouped_xyz / self.radius
            if features is not None:
                grouped_features = grouping_operation(features.contiguous(), idx)
                if self.use_xyz:
                    new_features = torch.cat([grouped_xyz, grouped_features], dim=1)
                else:
                    new_features = grouped_features
            else:
                new_features = grouped_xyz

        ret = [new_features]
        if self.ret_grouped_xyz:
            ret.append(grouped_xyz)
        if self.ret_unique_cnt:
            ret.append(unique_cnt)
        if len(ret) == 1:
            return ret[0]
        return tuple(ret)


class GroupAll(nn.Module):
    """pointnet2_utils.py:379-425.  The reference forgets to store `ret_grouped_xyz`
    (AttributeError at :422); here the flag is honoured."""

    def __init__(self, use_xyz=True, ret_grouped_xyz=False):
        super().__init__()
        self.use_xyz = use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is not None:
            grouped_features = features.unsqueeze(2)
            if self.use_xyz:
                new_features = torch.cat([grouped_xyz, grouped_features], dim=1)
            else:
                new_features = grouped_features
        else:
            new_features = grouped_xyz
        if self.ret_grouped_xyz:
            return new_features, grouped_xyz
        return new_features

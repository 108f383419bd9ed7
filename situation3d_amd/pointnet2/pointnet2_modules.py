"""PointNet++ set-abstraction / feature-propagation modules -- host-side mirror of the
reference's lib/pointnet2/pointnet2_modules.py: same class names, keyword-only ctor arguments,
child-module names (=> identical state_dict keys) and return tuples, running on the gfx950 HIP
ops in `pointnet2_utils`.

forward() contracts (reference file:line):
  PointnetSAModuleVotes    pointnet2_modules.py:210-277  -> (new_xyz, new_features, inds[, unique_cnt])
  PointnetSAModule{,MSG}   pointnet2_modules.py:34-75    -> (new_xyz, new_features)
  PointnetSAModuleMSGVotes pointnet2_modules.py:314-358  -> (new_xyz, new_features, inds)
  PointnetFPModule         pointnet2_modules.py:376-421  -> new_features
  PointnetLFPModuleMSG     pointnet2_modules.py:459-501  -> new_features
"""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused_mlp
from . import pointnet2_utils
from . import pytorch_utils as pt_utils


def _sample_centres(xyz, npoint, inds=None):
    """FPS (unless indices are given) + gather of the centre coordinates
    (pointnet2_modules.py:233-240): returns (new_xyz (B,npoint,3) or None, inds)."""
    if npoint is None:
        return None, inds
    if inds is None:
        inds = pointnet2_utils.furthest_point_sample(xyz, npoint)
    if xyz.is_cuda and not (torch.is_grad_enabled() and xyz.requires_grad):
        return pointnet2_utils.gather_xyz(xyz, inds), inds  # one kernel, no transposes
    xyz_flipped = xyz.transpose(1, 2).contiguous()
    new_xyz = pointnet2_utils.gather_operation(xyz_flipped, inds).transpose(1, 2).contiguous()
    return new_xyz, inds


def _max_over_samples(x):
    """F.max_pool2d(x, [1, nsample]).squeeze(-1)  (pointnet2_modules.py:66-69), as a last-dim
    max reduction (same values; the gradient goes to one arg-max element in both forms)."""
    return torch.max(x, dim=3)[0]


def _build_scales(npoint, radii, nsamples, mlps, bn, use_xyz, sample_uniformly):
    groupers, nets = nn.ModuleList(), nn.ModuleList()
    for radius, nsample, mlp_spec in zip(radii, nsamples, mlps):
        groupers.append(
            pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz,
                                          sample_uniformly=sample_uniformly)
            if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
        if use_xyz:
            mlp_spec[0] += 3  # in-place on the caller's list, like the reference (:120-121)
        nets.append(pt_utils.SharedMLP(mlp_spec, bn=bn))
    return groupers, nets


class _PointnetSAModuleBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None):
        new_xyz, _ = _sample_centres(xyz, self.npoint)
        pooled = [_max_over_samples(mlp(grouper(xyz, new_xyz, features)))
                  for grouper, mlp in zip(self.groupers, self.mlps)]
        return new_xyz, torch.cat(pooled, dim=1)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Multi-scale grouping SA layer (pointnet2_modules.py:78-124)."""

    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int],
                 mlps: List[List[int]], bn: bool = True, use_xyz: bool = True,
                 sample_uniformly: bool = False):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.groupers, self.mlps = _build_scales(npoint, radii, nsamples, mlps, bn, use_xyz,
                                                 sample_uniformly)


class PointnetSAModule(PointnetSAModuleMSG):
    """Single-scale SA layer (pointnet2_modules.py:127-161)."""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None,
                 nsample: int = None, bn: bool = True, use_xyz: bool = True):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn,
                         use_xyz=use_xyz)


class PointnetSAModuleVotes(nn.Module):
    """SA layer that also returns the sampled indices (pointnet2_modules.py:164-277)."""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None,
                 nsample: int = None, bn: bool = True, use_xyz: bool = True,
                 pooling: str = 'max', sigma: float = None, normalize_xyz: bool = False,
                 sample_uniformly: bool = False, ret_unique_cnt: bool = False):
        super().__init__()
        self.npoint = npoint
        self.radius = radius
        self.nsample = nsample
        self.pooling = pooling
        self.mlp_module = None
        self.use_xyz = use_xyz
        self.sigma = sigma
        if self.sigma is None:
            self.sigma = self.radius / 2
        self.normalize_xyz = normalize_xyz
        self.ret_unique_cnt = ret_unique_cnt

        if npoint is not None:
            self.grouper = pointnet2_utils.QueryAndGroup(
                radius, nsample, use_xyz=use_xyz, ret_grouped_xyz=True,
                normalize_xyz=normalize_xyz, sample_uniformly=sample_uniformly,
                ret_unique_cnt=ret_unique_cnt)
        else:
            self.grouper = pointnet2_utils.GroupAll(use_xyz, ret_grouped_xyz=True)

        mlp_spec = mlp
        if use_xyz and len(mlp_spec) > 0:
            mlp_spec[0] += 3
        self.mlp_module = pt_utils.SharedMLP(mlp_spec, bn=bn)

    # compact mode is kept for a layer while at most this fraction of its (centre, sample) positions is distinct
    COMPACT_MAX_FRACTION = 0.6
    # True: the fused paths also write the pooled features point-major (B, npoint, C) from the pooling kernel and
    # hang that twin on the returned tensor (fused_mlp.point_major_of) -- the next level's gathers and the
    # Q-Former's scene tokens read rows; set by a backbone that chains levels (model.PointNet2Encoder)
    emit_point_major = False

    def _compact_pays(self, compact):
        """Decided ONCE per layer from the first batch it sees (one host read of the counts, outside any
        hipGraph capture: warm-up steps come first), then static -- the launch structure of a captured step
        cannot follow the data.  Dense scans (full lists) keep the dense kernels."""
        decided = getattr(self, "_compact_decision", None)
        if decided is None:
            if torch.cuda.is_current_stream_capturing():
                return False   # never decide inside a capture
            b, m, ns = compact.shape
            frac = float(compact.n_act.float().mean().item()) / (m * ns)
            decided = self._compact_decision = frac <= self.COMPACT_MAX_FRACTION
        return decided

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None,
                inds: torch.Tensor = None, geometry=None):
        """`geometry` (optional, not in the reference signature): a precomputed
        (inds, new_xyz, ball_idx) triple for this layer from geometry.GeometryPlan."""
        compactable = (self.pooling == 'max' and self.npoint is not None and not self.ret_unique_cnt
                       and not getattr(self.grouper, "sample_uniformly", True)
                       and fused_mlp.compact_applies(self.mlp_module, xyz, features, self.npoint, self.nsample))
        if geometry is not None and compactable:
            # distinct neighbours only: ball query pads short lists with copies of the first hit, and a copy
            # is an identical column all the way up to the max-pool (csrc/compact.hip)
            inds, new_xyz, ball_idx = geometry[:3]
            compact = geometry[3] if len(geometry) > 3 and geometry[3] is not None else fused_mlp.compact_lists(ball_idx)
            compactable = self._compact_pays(compact)
        if geometry is not None and compactable:
            new_features = fused_mlp.fused_sa_compact(self.mlp_module, xyz, new_xyz, features, compact, self.nsample,
                                                      self.grouper.radius, self.grouper.use_xyz,
                                                      self.grouper.normalize_xyz, want_pm=self.emit_point_major)
            return new_xyz, new_features, inds
        # dense MFMA-path levels with a wide feature input: no grouped tensor either (first layer gathers on load)
        dense_gather = (self.pooling == 'max' and self.npoint is not None and not self.ret_unique_cnt
                        and not getattr(self.grouper, "sample_uniformly", True)
                        and fused_mlp.dense_gather_applies(self.mlp_module, xyz, features, self.npoint, self.nsample,
                                                           self.grouper.use_xyz))
        if geometry is not None:
            inds, new_xyz, ball_idx = geometry[:3]
            assert inds.shape[1] == self.npoint
            if dense_gather:
                new_features = fused_mlp.fused_sa_dense(self.mlp_module, xyz, new_xyz, features, ball_idx, self.nsample,
                                                        self.grouper.radius, self.grouper.normalize_xyz,
                                                        want_pm=self.emit_point_major)
                return new_xyz, new_features, inds
            grouped = self.grouper(xyz, new_xyz, features, idx=ball_idx)
        elif compactable:
            if inds is not None:
                assert inds.shape[1] == self.npoint
            new_xyz, inds = _sample_centres(xyz, self.npoint, inds)
            ball_idx = pointnet2_utils.ball_query(self.grouper.radius, self.nsample, xyz, new_xyz)
            compact = fused_mlp.compact_lists(ball_idx)
            if self._compact_pays(compact):
                new_features = fused_mlp.fused_sa_compact(self.mlp_module, xyz, new_xyz, features, compact,
                                                          self.nsample, self.grouper.radius, self.grouper.use_xyz,
                                                          self.grouper.normalize_xyz, want_pm=self.emit_point_major)
                return new_xyz, new_features, inds
            if dense_gather:
                new_features = fused_mlp.fused_sa_dense(self.mlp_module, xyz, new_xyz, features, ball_idx, self.nsample,
                                                        self.grouper.radius, self.grouper.normalize_xyz,
                                                        want_pm=self.emit_point_major)
                return new_xyz, new_features, inds
            grouped = self.grouper(xyz, new_xyz, features, idx=ball_idx)
        else:
            if inds is not None:
                assert inds.shape[1] == self.npoint
            new_xyz, inds = _sample_centres(xyz, self.npoint, inds)
            if dense_gather:
                ball_idx = pointnet2_utils.ball_query(self.grouper.radius, self.nsample, xyz, new_xyz)
                new_features = fused_mlp.fused_sa_dense(self.mlp_module, xyz, new_xyz, features, ball_idx, self.nsample,
                                                        self.grouper.radius, self.grouper.normalize_xyz,
                                                        want_pm=self.emit_point_major)
                return new_xyz, new_features, inds
            grouped = self.grouper(xyz, new_xyz, features)
        if self.ret_unique_cnt:
            grouped_features, grouped_xyz, unique_cnt = grouped
        else:
            grouped_features, grouped_xyz = grouped

        if self.pooling == 'max' and fused_mlp.can_fuse(self.mlp_module, grouped_features):
            # training-mode Conv1x1+BN+ReLU stack and the max over nsample as fused MFMA kernels
            new_features = fused_mlp.fused_mlp_max(self.mlp_module, grouped_features, want_pm=self.emit_point_major)
            if self.ret_unique_cnt:
                return new_xyz, new_features, inds, unique_cnt
            return new_xyz, new_features, inds

        new_features = self.mlp_module(grouped_features)  # (B, mlp[-1], npoint, nsample)
        if self.pooling == 'max':
            new_features = torch.max(new_features, dim=3, keepdim=True)[0]
        elif self.pooling == 'avg':
            new_features = F.avg_pool2d(new_features, kernel_size=[1, new_features.size(3)])
        elif self.pooling == 'rbf':
            # RBF-weighted sum over the neighbourhood, normalised by nsample (:264-271)
            rbf = torch.exp(-1 * grouped_xyz.pow(2).sum(1, keepdim=False) / (self.sigma ** 2) / 2)
            new_features = torch.sum(new_features * rbf.unsqueeze(1), -1, keepdim=True) \
                / float(self.nsample)
        new_features = new_features.squeeze(-1)  # (B, mlp[-1], npoint)

        if self.ret_unique_cnt:
            return new_xyz, new_features, inds, unique_cnt
        return new_xyz, new_features, inds


class PointnetSAModuleMSGVotes(nn.Module):
    """Multi-scale SA layer returning indices (pointnet2_modules.py:279-358)."""

    def __init__(self, *, mlps: List[List[int]], npoint: int, radii: List[float],
                 nsamples: List[int], bn: bool = True, use_xyz: bool = True,
                 sample_uniformly: bool = False):
        super().__init__()
        assert len(mlps) == len(nsamples) == len(radii)
        self.npoint = npoint
        self.groupers, self.mlps = _build_scales(npoint, radii, nsamples, mlps, bn, use_xyz,
                                                 sample_uniformly)

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None,
                inds: torch.Tensor = None):
        new_xyz, inds = _sample_centres(xyz, self.npoint, inds)
        pooled = [_max_over_samples(mlp(grouper(xyz, new_xyz, features)))
                  for grouper, mlp in zip(self.groupers, self.mlps)]
        return new_xyz, torch.cat(pooled, dim=1), inds


class PointnetFPModule(nn.Module):
    """Feature propagation by inverse-distance 3-NN interpolation (pointnet2_modules.py:361-421)."""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn)

    def forward(self, unknown: torch.Tensor, known: torch.Tensor, unknow_feats: torch.Tensor,
                known_feats: torch.Tensor) -> torch.Tensor:
        if known is not None:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            dist_recip = 1.0 / (dist + 1e-8)  # :400-402
            norm = torch.sum(dist_recip, dim=2, keepdim=True)
            weight = dist_recip / norm
            interpolated_feats = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:
            interpolated_feats = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))

        if unknow_feats is not None:
            new_features = torch.cat([interpolated_feats, unknow_feats], dim=1)  # (B, C2+C1, n)
        else:
            new_features = interpolated_feats
        return self.mlp(new_features.unsqueeze(-1)).squeeze(-1)


class PointnetLFPModuleMSG(nn.Module):
    """Learnable feature propagation (pointnet2_modules.py:423-501)."""

    def __init__(self, *, mlps: List[List[int]], radii: List[float], nsamples: List[int],
                 post_mlp: List[int], bn: bool = True, use_xyz: bool = True,
                 sample_uniformly: bool = False):
        super().__init__()
        assert len(mlps) == len(nsamples) == len(radii)
        self.post_mlp = pt_utils.SharedMLP(post_mlp, bn=bn)
        self.groupers, self.mlps = _build_scales(0, radii, nsamples, mlps, bn, use_xyz,
                                                 sample_uniformly)

    def forward(self, xyz2: torch.Tensor, xyz1: torch.Tensor, features2: torch.Tensor,
                features1: torch.Tensor) -> torch.Tensor:
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            new_features = _max_over_samples(mlp(grouper(xyz1, xyz2, features1)))  # (B, C, N2)
            if features2 is not None:
                new_features = torch.cat([new_features, features2], dim=1)
            outs.append(self.post_mlp(new_features.unsqueeze(-1)))
        return torch.cat(outs, dim=1).squeeze(-1)

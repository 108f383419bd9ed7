"""Geometry plan of the point encoder: everything in the SA stack that depends on xyz ONLY.

FPS indices, centre coordinates and ball-query neighbour lists of all set-abstraction levels are
functions of the input coordinates alone (pointnet2_modules.py:233-240 + pointnet2_utils.py:334)
-- no learned parameter and no feature enters them.  FPS is a chain of ~3800 strictly dependent
rounds per step (latency-bound, a few workgroups), so it is the worst possible citizen of the
critical path and the best possible candidate for overlap: a `GeometryPlan` computes the whole
chain into PREALLOCATED buffers through the C ABI (no allocator traffic, safe on a side stream
and inside hipGraph capture), which lets the training step of batch i run concurrently with the
geometry of batch i+1 (graph_step.GraphedTrainStep, `prefetch_geometry=True`) -- the same role
the reference gives its DataLoader workers for the CPU-side voxelisation (lib/sepdataset.py).

The numbers produced are bit-identical to calling the ops inline.
"""
import ctypes
import os

import torch

from . import _lib, timeline
from .pointnet2 import _ext, fused_mlp


# SIG3D_NESTED_FPS=0 runs the dependent rounds on every level (A/B timing; results are identical)
NESTED_FPS = os.environ.get("SIG3D_NESTED_FPS", "1") != "0"


class GeometryPlan:
    """levels: list of (npoint, radius, nsample) from the first SA layer down."""

    def __init__(self, batch, n_points, levels, device):
        self.levels = list(levels)
        self.batch, self.n_points = batch, n_points
        self.inds, self.new_xyz, self.ball_idx, self._temp, self.compact = [], [], [], [], []
        self.fps_proven = []  # per level: (B,) int32, 1 where the nested-FPS proof held (levels >= 1)
        self._bq_work = None    # scratch of the multi-level ball query (allocated once: static under hipGraph)
        self._bq_clean = False  # True once the workspace has been through a call (counters zero again)
        self._bq_levels = None  # its level table (ctypes array of sig3d_bq_level: raw pointers of the plan's buffers)
        n = n_points
        for npoint, radius, nsample in self.levels:
            self.inds.append(torch.zeros(batch, npoint, dtype=torch.int32, device=device))
            self.new_xyz.append(torch.zeros(batch, npoint, 3, dtype=torch.float32, device=device))
            self.ball_idx.append(torch.zeros(batch, npoint, nsample, dtype=torch.int32, device=device))
            self._temp.append(torch.zeros(batch, max(n, 128), dtype=torch.float32, device=device))
            self.fps_proven.append(torch.zeros(batch, dtype=torch.int32, device=device))
            # distinct neighbours of the padded lists (csrc/compact.hip), for the levels that run the MFMA path
            big = fused_mlp.COMPACT and batch * npoint * nsample >= fused_mlp.COMPACT_MIN_POSITIONS
            self.compact.append(fused_mlp.CompactLists(batch, npoint, nsample, device) if big else None)
            n = npoint

    def compute(self, xyz):
        """Launch the chain for xyz (B,N,3) on the current stream; fills the plan's buffers."""
        dev = _lib.require_device(xyz)
        if not xyz.is_contiguous() or xyz.dtype != torch.float32:
            raise RuntimeError("xyz must be a contiguous float tensor")
        b, n, _ = xyz.shape
        assert (b, n) == (self.batch, self.n_points)
        s = _lib.stream_ptr(dev)
        cur = xyz
        with torch.cuda.device(dev):
            timeline.mark("geo:start")
            srcs = []
            for lvl, (npoint, radius, nsample) in enumerate(self.levels):
                if lvl == 0 or not NESTED_FPS:
                    _lib.call("sig3d_furthest_point_sampling", b, n, npoint, _lib.ptr(cur),
                              _lib.ptr(self._temp[lvl]), _lib.ptr(self.inds[lvl]), s)
                else:
                    # `cur` is the FPS-ordered output of the level above: prove inds == 0..m-1 instead
                    # of running the dependent rounds (csrc/sampling.hip, same results for any input)
                    _lib.call("sig3d_furthest_point_sampling_nested", b, n, npoint, _lib.ptr(cur),
                              _lib.ptr(self._temp[lvl]), _lib.ptr(self.inds[lvl]),
                              _lib.ptr(self.fps_proven[lvl]), s)
                timeline.mark("geo:L%d fps" % (lvl + 1))
                _lib.call("sig3d_gather_xyz", b, n, npoint, _lib.ptr(cur), _lib.ptr(self.inds[lvl]),
                          _lib.ptr(self.new_xyz[lvl]), s)
                srcs.append(cur)
                cur, n = self.new_xyz[lvl], npoint
            # the neighbour lists of ALL levels depend on coordinates only: one scatter + one rank launch
            # (csrc/ball_query.hip: centres binned into cells, points streamed once) instead of a chain per level
            ptrs = [t.data_ptr() for t in srcs]
            if self._bq_levels is None or self._bq_srcs != ptrs:   # the level table holds raw pointers
                probs = [(srcs[l], self.new_xyz[l], lv[1], lv[2], self.ball_idx[l]) for l, lv in enumerate(self.levels)]
                self._bq_levels = _lib.bq_levels(probs)
                need = _lib.bq_levels_workspace_bytes(b, self._bq_levels)
                assert need >= 0, "too many centres for one multi-level ball query"
                if self._bq_work is None or self._bq_work.numel() < need:
                    self._bq_work = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
                self._bq_srcs = ptrs
                self._bq_clean = False   # a new workspace / problem list: the next call zeroes the counters itself
            # after one completed call the workspace's counters are zero again (the rank kernel cleans up): no memset
            _lib.call("sig3d_ball_query_levels_ex", b, len(self._bq_levels), self._bq_levels, _lib.ptr(self._bq_work),
                      self._bq_work.numel(), _lib.BQ_CLEAN if self._bq_clean else 0, s)
            self._bq_clean = True
            for lvl in range(len(self.levels)):
                if self.compact[lvl] is not None:
                    self.compact[lvl].compute(self.ball_idx[lvl])
            timeline.mark("geo:lists")
        return self

    def level(self, i):
        return self.inds[i], self.new_xyz[i], self.ball_idx[i], self.compact[i]

    def copy_from(self, other):
        """Hand-over at the end of a step: every tensor of `other` into this plan, ONE launch
        (table_copy.TableCopy) instead of one copy kernel per tensor."""
        pairs = list(zip(self.inds + self.new_xyz + self.ball_idx, other.inds + other.new_xyz + other.ball_idx))
        for mine, theirs in zip(self.compact, other.compact):
            if mine is not None:
                pairs += list(zip(mine.tensors(), theirs.tensors()))
        if getattr(self, "_table_copy", None) is None:
            from .table_copy import TableCopy
            self._table_copy = TableCopy(self.inds[0].device)
        self._table_copy(pairs)

"""Scene voxelisation / de-duplication on the GPU, behind the reference's own names
(SURVEY.md section 8(f) rank 2).

The reference prepares every sample in DataLoader workers with numpy (lib/sepdataset.py:286-302:
augmentation rotations, min shift, `Voxelizer.voxelize` -> `sparse_quantize` -> FNV keys -> np.unique)
and ships the result host->device.  Here the raw scenes stay in HBM and one call of
`sig3d_voxelize` (csrc/voxelize.hip) prepares a whole ragged batch; results are bit-identical to the
numpy path (indices, inverse map, cells), so downstream code sees the same points in the same order.

  fnv_hash_vec, sparse_quantize   lib/openscene/voxelization_utils.py:9-24, 47-140
  Voxelizer                       lib/openscene/voxelizer_dev.py:15-56
  voxelize_batch                  the batched form the step loop uses (no host synchronisation)

There is no CPU path: tensors must live on the GPU and libsig3d_hip.so must be built.
"""
import ctypes

import torch

from . import _lib


def _require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError("%s: CPU not supported (tensor must be on the GPU)" % what)


def fnv_hash_vec(arr):
    """(N, D) floor'd coordinates (any float/int dtype) -> (N,) int64 holding the uint64 FNV key bits
    (voxelization_utils.py:9-24; torch has no uint64 arithmetic, compare with `.view(np.uint64)`)."""
    assert arr.ndim == 2
    _require_gpu(arr, "fnv_hash_vec")
    cells = arr.to(torch.int64).contiguous()
    out = torch.empty(cells.shape[0], dtype=torch.int64, device=arr.device)
    _lib.call("sig3d_fnv_hash_vec", cells.shape[0], cells.shape[1], _lib.ptr(cells), _lib.ptr(out),
              _lib.stream_ptr(arr.device))
    return out


class VoxelBatch:
    """Result of voxelize_batch: flat per-point arrays with scene offsets; the first num_unique[s]
    entries of scene s's segment of inds / cells / feats / labels are valid."""

    def __init__(self, offsets, inds, inverse, num_unique, cells, feats, labels, mins):
        self.offsets, self.inds, self.inverse, self.num_unique = offsets, inds, inverse, num_unique
        self.cells, self.feats, self.labels, self.mins = cells, feats, labels, mins

    def scene(self, s):
        """Host-synchronising view of one scene (the tuple the reference's voxelize returns, with
        return_ind=True): cells (U,3) f64, feats, labels, inverse (N,), inds (U,)."""
        lo, hi = int(self.offsets_host[s]), int(self.offsets_host[s + 1])
        u = int(self.num_unique[s])
        feats = self.feats[lo:lo + u] if self.feats is not None else None
        labels = self.labels[lo:lo + u].long() if self.labels is not None else None
        return (self.cells[lo:lo + u].double(), feats, labels, self.inverse[lo:hi].long(),
                self.inds[lo:lo + u].long())


_work = {}


def _workspace(nbytes, device):
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _work.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _work[key] = buf
    return buf


def voxelize_batch(coords, offsets, feats=None, labels=None, rotations=None, flips=None, voxel_size=None,
                   quantization_size=None, shift_min=True, want_cells=True):
    """coords (total,3) f32/f64 flat batch; offsets: python list / CPU int tensor of B+1 scene starts.
    rotations: optional (B,K,3,3) f64 applied in order as p <- p.R^T (sepdataset.py:267,279,291).
    flips: optional B ints, bit a set = axis a negated first (mirror augmentation, sepdataset.py:246,255).
    voxel_size: cells = floor(p * (1/voxel_size)) (voxelizer_dev.py:29-43); or quantization_size
    (scalar or 3 values): cells = floor(p / q) (voxelization_utils.py:108).
    Everything is enqueued on the current stream; nothing synchronises with the host."""
    _require_gpu(coords, "voxelize_batch")
    assert coords.ndim == 2 and coords.shape[1] == 3 and coords.dtype in (torch.float32, torch.float64)
    assert (voxel_size is None) != (quantization_size is None), "give voxel_size or quantization_size"
    dev = coords.device
    coords = coords.contiguous()
    off_host = torch.as_tensor(offsets, dtype=torch.int32, device="cpu")
    b = off_host.numel() - 1
    total = int(off_host[-1])
    assert total == coords.shape[0] and int(off_host[0]) == 0
    max_n = int((off_host[1:] - off_host[:-1]).max()) if b else 0
    off_dev = off_host.to(dev, non_blocking=True)
    if voxel_size is not None:
        quant, divide = [1 / voxel_size] * 3, 0
    else:
        q = quantization_size
        quant = [float(v) for v in q] if isinstance(q, (list, tuple)) or hasattr(q, "__len__") else [float(q)] * 3
        assert len(quant) == 3, "Quantization size and coordinates size mismatch."
        divide = 1
    quant_c = (ctypes.c_double * 3)(*quant)
    n_rot, rot = 0, None
    if rotations is not None:
        rot = torch.as_tensor(rotations, dtype=torch.float64).reshape(b, -1, 9).to(dev).contiguous()
        n_rot = rot.shape[1]
    flips_dev = None
    if flips is not None:
        flips_dev = torch.as_tensor(flips, dtype=torch.int32).to(dev)
        assert flips_dev.numel() == b
    c_feat = 0
    feats_out = labels_out = labels32 = None
    if feats is not None:
        assert feats.ndim == 2 and feats.shape[0] == total and feats.dtype == torch.float32
        feats = feats.contiguous()
        c_feat = feats.shape[1]
        feats_out = torch.empty_like(feats)
    if labels is not None:
        assert labels.shape[0] == total
        labels32 = labels.to(torch.int32).contiguous()
        labels_out = torch.empty_like(labels32)
    inds = torch.empty(total, dtype=torch.int32, device=dev)
    inverse = torch.empty(total, dtype=torch.int32, device=dev)
    num_unique = torch.empty(b, dtype=torch.int32, device=dev)
    cells = torch.empty(total, 3, dtype=torch.int32, device=dev) if want_cells else None
    mins = torch.empty(b, 3, dtype=torch.float64, device=dev)
    nbytes = _lib.load().sig3d_voxelize_workspace_bytes(b, total, max_n)
    work = _workspace(nbytes, dev)
    _lib.call("sig3d_voxelize", b, max_n, _lib.ptr(off_dev), _lib.ptr(coords),
              1 if coords.dtype == torch.float64 else 0, n_rot, _lib.ptr(rot), _lib.ptr(flips_dev),
              1 if shift_min else 0, divide,
              ctypes.cast(quant_c, ctypes.c_void_p), c_feat, _lib.ptr(feats), _lib.ptr(labels32), _lib.ptr(inds),
              _lib.ptr(inverse), _lib.ptr(num_unique), _lib.ptr(cells), _lib.ptr(feats_out), _lib.ptr(labels_out),
              _lib.ptr(mins), _lib.ptr(work), nbytes, total, _lib.stream_ptr(dev))
    out = VoxelBatch(off_dev, inds, inverse, num_unique, cells, feats_out, labels_out, mins)
    out.offsets_host = off_host
    return out


def sparse_quantize(coords, feats=None, labels=None, ignore_label=255,
                    set_ignore_label_when_collision=False, return_index=False, hash_type="fnv",
                    quantization_size=1):
    """voxelization_utils.py:47-140 for one (N,3) GPU tensor: same arguments, same returns (tensors
    instead of arrays; index tensors are int64).  Only the 'fnv' keys the reference's callers use are
    built on the GPU."""
    use_label = labels is not None
    use_feat = feats is not None
    if not use_label and not use_feat:
        return_index = True
    assert hash_type in ["ravel", "fnv"], \
        "Invalid hash_type. Either ravel, or fnv allowed. You put hash_type=" + hash_type
    if hash_type == "ravel":
        raise NotImplementedError("hash_type='ravel' has no GPU path (no caller in the reference uses it)")
    assert coords.ndim == 2, \
        "The coordinates must be a 2D matrix. The shape of the input is " + str(tuple(coords.shape))
    if coords.shape[1] != 3:
        raise NotImplementedError("the GPU path handles 3-D coordinates")
    if use_feat:
        assert feats.ndim == 2 and coords.shape[0] == feats.shape[0]
    if use_label:
        assert coords.shape[0] == len(labels)
    if isinstance(quantization_size, (list, tuple)) or hasattr(quantization_size, "__len__"):
        assert len(quantization_size) == 3, "Quantization size and coordinates size mismatch."
    elif not isinstance(quantization_size, (int, float)):
        raise ValueError("Not supported type for quantization_size.")
    if coords.dtype not in (torch.float32, torch.float64):
        coords = coords.double()
    n = coords.shape[0]
    vb = voxelize_batch(coords, [0, n], feats if use_feat else None, labels if use_label else None,
                        quantization_size=quantization_size, shift_min=False)
    u = int(vb.num_unique[0])
    inds = vb.inds[:u].long()
    if use_label:
        filtered = vb.labels[:u].to(labels.dtype)
        if set_ignore_label_when_collision:
            counts = torch.bincount(vb.inverse.long(), minlength=u)
            filtered[counts > 1] = ignore_label
        if return_index:
            return inds, filtered
        return vb.cells[:u].double(), vb.feats[:u], filtered
    if return_index:
        return inds, vb.inverse.long()
    if use_feat:
        return vb.cells[:u].double(), vb.feats[:u]
    return vb.cells[:u].double()


class Voxelizer:
    """voxelizer_dev.py:15-56 over GPU tensors (one scene per call, like the reference's; the step loop
    uses voxelize_batch)."""

    def __init__(self, voxel_size=1, ignore_label=255):
        self.voxel_size = voxel_size
        self.ignore_label = ignore_label

    def get_transformation_matrix(self):
        m = torch.eye(4, dtype=torch.float64)
        m[0, 0] = m[1, 1] = m[2, 2] = 1 / self.voxel_size
        return m

    def voxelize(self, coords, feats, labels, center=None, link=None, return_ind=False):
        assert coords.shape[1] == 3 and coords.shape[0] == feats.shape[0] and coords.shape[0]
        n = coords.shape[0]
        vb = voxelize_batch(coords, [0, n], feats, labels, voxel_size=self.voxel_size, shift_min=False)
        u = int(vb.num_unique[0])
        cells = vb.cells[:u].double()
        # voxelizer_dev.py:45 -- the caller must have moved the scene to the origin
        assert float(torch.floor(vb.mins[0] * (1 / self.voxel_size)).sum()) == 0, \
            "Minimum of coordinates are not zeros!"
        inds = vb.inds[:u].long()
        out = (cells, vb.feats[:u], vb.labels[:u].to(labels.dtype), vb.inverse.long())
        if return_ind:
            return out + (inds,)
        if link is not None:
            return out + (link[inds],)
        return out

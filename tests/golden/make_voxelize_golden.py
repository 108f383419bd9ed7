"""Generate tests/golden/voxelize_golden.npz -- run ONLY in the build container (/root/reference).

    python tests/golden/make_voxelize_golden.py

The REFERENCE's own lib/openscene/voxelization_utils.py and voxelizer_dev.py are imported unmodified
(one in-process shim: `collections.Sequence`, removed in Python 3.10, is aliased to
collections.abc.Sequence) and run on seeded scenes; the reference's dataset code around them
(lib/sepdataset.py:286-302: np.dot rotations, min shift) is spelled out inline with the same numpy
calls.  Only inputs and the reference's outputs are stored.
"""
import collections
import collections.abc
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

collections.Sequence = collections.abc.Sequence
sys.path.insert(0, REF)
from lib.openscene import voxelization_utils as ref_utils  # noqa: E402
from lib.openscene.voxelizer_dev import Voxelizer as RefVoxelizer  # noqa: E402


def rotz(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def rotx(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def roty(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def scene(rng, n, dup=0.0):
    p = (rng.random((n, 3)) * np.array([8.0, 8.0, 3.0])).astype(np.float32)
    if dup:
        k = int(n * dup)
        p[rng.integers(0, n, k)] = p[rng.integers(0, n, k)]
    return p


def main():
    rng = np.random.default_rng(20240917)
    out = {}
    # fnv_hash_vec known answers (float cells, as sparse_quantize passes them)
    cells = np.floor(rng.random((257, 3)) * 400.0)
    out["fnv_cells"] = cells
    out["fnv_keys"] = ref_utils.fnv_hash_vec(cells)
    cells4 = np.floor(rng.random((64, 4)) * 1000.0)
    out["fnv_cells4"] = cells4
    out["fnv_keys4"] = ref_utils.fnv_hash_vec(cells4)

    cases = [("plain_f32", 5000, 0.02, 0.0, []),
             ("coarse_f32", 3000, 0.25, 0.0, []),
             ("dups_f32", 4000, 0.05, 0.3, []),
             ("rot_f64", 5000, 0.02, 0.0, [rotx(0.03), roty(-0.05), rotz(0.07)]),
             ("rotz_f64", 2500, 0.1, 0.1, [rotz(-0.08)])]
    names = []
    for name, n, voxel, dup, rots in cases:
        p = scene(rng, n, dup)
        feats = rng.random((n, 3)).astype(np.float32)
        labels = rng.integers(0, 20, n).astype(np.int64)
        out[name + "_points"] = p.copy()
        out[name + "_feats"] = feats
        out[name + "_labels"] = labels
        out[name + "_voxel"] = np.float64(voxel)
        out[name + "_rots"] = np.array(rots, dtype=np.float64).reshape(-1, 3, 3)
        pc = p
        for r in rots:  # sepdataset.py:267,279,291
            pc = np.dot(pc[:, 0:3], np.transpose(r))
        mins = pc.min(0)  # sepdataset.py:298-299
        pc = pc - mins
        cells_u, feats_u, labels_u, inverse, inds = RefVoxelizer(voxel_size=voxel).voxelize(
            pc, feats, labels, return_ind=True)
        out[name + "_mins"] = np.asarray(mins, dtype=np.float64)
        out[name + "_cells"] = cells_u
        out[name + "_feats_out"] = feats_u
        out[name + "_labels_out"] = labels_u
        out[name + "_inverse"] = np.asarray(inverse)
        out[name + "_inds"] = np.asarray(inds)
        names.append(name)
    out["cases"] = np.array(names)

    # sparse_quantize on its own (division by the cell size, labels with collisions)
    p = scene(rng, 3000, 0.2) - 2.0   # negative coordinates too
    labels = rng.integers(0, 20, 3000).astype(np.int64)
    feats = rng.random((3000, 3)).astype(np.float32)
    out["sq_points"], out["sq_labels"], out["sq_feats"] = p, labels, feats
    inds, inv = ref_utils.sparse_quantize(p, return_index=True, quantization_size=0.1)
    out["sq_inds"], out["sq_inverse"] = inds, inv
    inds_l, lab = ref_utils.sparse_quantize(p, feats, labels, return_index=True, quantization_size=0.1,
                                            set_ignore_label_when_collision=True)
    out["sq_inds_l"], out["sq_labels_l"] = inds_l, lab
    c, f = ref_utils.sparse_quantize(p, feats, quantization_size=[0.1, 0.2, 0.3])
    out["sq_cells_aniso"], out["sq_feats_aniso"] = c, f
    path = os.path.join(HERE, "voxelize_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()

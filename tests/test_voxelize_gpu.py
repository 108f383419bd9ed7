"""GPU parity of csrc/voxelize.hip (through the C ABI, via situation3d_amd.voxelizer): bit-exact
against the reference-generated golden vectors and against the numpy oracle on larger / degenerate
scenes.  Integer and index results: exact.  float64 cells: exact.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "voxelize_golden.npz")


def _g():
    return {k: v for k, v in np.load(GOLD, allow_pickle=False).items()}


def _vox():
    from situation3d_amd import voxelizer
    return voxelizer


def _ref():
    from oracle import voxelize_ref
    return voxelize_ref


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t.to(dtype) if dtype is not None else t


def test_fnv_hash_vec_vs_reference_vectors():
    g = _g()
    for cells, keys in ((g["fnv_cells"], g["fnv_keys"]), (g["fnv_cells4"], g["fnv_keys4"])):
        out = _vox().fnv_hash_vec(_t(cells)).cpu().numpy().view(np.uint64)
        assert np.array_equal(out, keys)


def test_batch_matches_reference_cases():
    """All five reference cases as ONE ragged batch would need one voxel size; run them per voxel size
    as 1-scene batches, then the two 0.02 m cases together."""
    g = _g()
    V = _vox()
    for name in g["cases"]:
        pts = g[name + "_points"]
        rots = g[name + "_rots"]
        vb = V.voxelize_batch(_t(pts), [0, len(pts)], _t(g[name + "_feats"]), _t(g[name + "_labels"]),
                              rotations=rots[None] if len(rots) else None, voxel_size=float(g[name + "_voxel"]))
        cells, feats, labels, inverse, inds = vb.scene(0)
        assert np.array_equal(vb.mins[0].cpu().numpy(), g[name + "_mins"]), name
        assert np.array_equal(inds.cpu().numpy(), g[name + "_inds"]), name
        assert np.array_equal(inverse.cpu().numpy(), g[name + "_inverse"]), name
        assert np.array_equal(cells.cpu().numpy(), g[name + "_cells"]), name
        assert np.array_equal(feats.cpu().numpy(), g[name + "_feats_out"]), name
        assert np.array_equal(labels.cpu().numpy(), g[name + "_labels_out"]), name


def test_ragged_batch_of_reference_cases():
    g = _g()
    names = ["plain_f32", "dups_f32", "coarse_f32"]
    pts = [g[n + "_points"] for n in names]
    off = np.concatenate([[0], np.cumsum([len(p) for p in pts])])
    # one voxel size per call: use each case's own by running the batch three times
    for k, name in enumerate(names):
        vb = _vox().voxelize_batch(_t(np.concatenate(pts)), off.tolist(), voxel_size=float(g[name + "_voxel"]))
        cells, _, _, inverse, inds = vb.scene(k)
        assert np.array_equal(inds.cpu().numpy(), g[name + "_inds"])
        assert np.array_equal(inverse.cpu().numpy(), g[name + "_inverse"])
        assert np.array_equal(cells.cpu().numpy(), g[name + "_cells"])


def test_voxelizer_class_matches_reference():
    g = _g()
    name = "plain_f32"
    pts = g[name + "_points"]
    pts = pts - pts.min(0)   # float32 shift, as sepdataset.py:298-299 on an un-augmented scene
    out = _vox().Voxelizer(voxel_size=float(g[name + "_voxel"])).voxelize(
        _t(pts), _t(g[name + "_feats"]), _t(g[name + "_labels"]), return_ind=True)
    cells, feats, labels, inverse, inds = [o.cpu().numpy() for o in out]
    assert cells.dtype == np.float64 and inds.dtype == np.int64 and labels.dtype == np.int64
    assert np.array_equal(cells, g[name + "_cells"]) and np.array_equal(inds, g[name + "_inds"])
    assert np.array_equal(inverse, g[name + "_inverse"]) and np.array_equal(feats, g[name + "_feats_out"])
    assert np.array_equal(labels, g[name + "_labels_out"])
    with pytest.raises(AssertionError, match="Minimum of coordinates"):
        _vox().Voxelizer(voxel_size=0.02).voxelize(_t(pts + 1.0), _t(g[name + "_feats"]), _t(g[name + "_labels"]))


def test_sparse_quantize_matches_reference():
    g = _g()
    V = _vox()
    p, labels, feats = _t(g["sq_points"]), _t(g["sq_labels"]), _t(g["sq_feats"])
    inds, inv = V.sparse_quantize(p, return_index=True, quantization_size=0.1)
    assert np.array_equal(inds.cpu().numpy(), g["sq_inds"]) and np.array_equal(inv.cpu().numpy(), g["sq_inverse"])
    inds_l, lab = V.sparse_quantize(p, feats, labels, return_index=True, quantization_size=0.1,
                                    set_ignore_label_when_collision=True)
    assert np.array_equal(inds_l.cpu().numpy(), g["sq_inds_l"]) and np.array_equal(lab.cpu().numpy(), g["sq_labels_l"])
    c, f = V.sparse_quantize(p, feats, quantization_size=[0.1, 0.2, 0.3])
    assert np.array_equal(c.cpu().numpy(), g["sq_cells_aniso"]) and np.array_equal(f.cpu().numpy(), g["sq_feats_aniso"])
    with pytest.raises(RuntimeError, match="CPU not supported"):
        V.sparse_quantize(p.cpu(), return_index=True)


@pytest.mark.parametrize("n,voxel", [(1, 0.02), (2047, 0.02), (2048, 0.05), (2049, 0.02), (40000, 0.02),
                                     (250000, 0.02), (250000, 0.3)])
def test_large_and_edge_sizes_vs_oracle(n, voxel):
    rng = np.random.default_rng(n)
    pts = (rng.random((n, 3)) * np.array([8.0, 8.0, 3.0])).astype(np.float32)
    cells, inds, inverse, mins = _ref().prepare_scene(pts, (), voxel)
    vb = _vox().voxelize_batch(_t(pts), [0, n], voxel_size=voxel)
    c, _, _, inv, ind = vb.scene(0)
    assert np.array_equal(ind.cpu().numpy(), inds) and np.array_equal(inv.cpu().numpy(), inverse)
    assert np.array_equal(c.cpu().numpy(), cells) and np.array_equal(vb.mins[0].cpu().numpy(), mins)


def test_degenerate_scenes_vs_oracle():
    """All points in one cell (a single run of n equal keys), heavy duplication, an empty scene in the
    middle of the batch, float64 input."""
    rng = np.random.default_rng(5)
    same = np.tile(np.array([[1.0, 2.0, 0.5]], dtype=np.float64), (5000, 1)) + rng.random((5000, 3)) * 1e-4
    heavy = rng.integers(0, 4, (7000, 3)).astype(np.float64) * 0.1 + 0.01
    third = rng.random((3000, 3)) * 4.0
    off = [0, 5000, 5000, 12000, 15000]   # scene 1 is empty
    flat = np.concatenate([same, heavy, third])
    vb = _vox().voxelize_batch(_t(flat), off, voxel_size=0.05)
    assert int(vb.num_unique[1]) == 0
    for s, pts in ((0, same), (2, heavy), (3, third)):
        cells, inds, inverse, mins = _ref().prepare_scene(pts, (), 0.05)
        c, _, _, inv, ind = vb.scene(s)
        assert np.array_equal(ind.cpu().numpy(), inds) and np.array_equal(inv.cpu().numpy(), inverse), s
        assert np.array_equal(c.cpu().numpy(), cells), s
    assert int(vb.num_unique[0]) <= 2 and int(vb.num_unique[2]) == 64


def test_round_trip_properties_at_full_size():
    """Size-independent properties on a B=8 x 150k batch: inverse maps every point to a kept point of
    the same cell; kept points are first occurrences; cells are distinct; keys ascend."""
    rng = np.random.default_rng(11)
    b, n = 8, 150000
    pts = (rng.random((b * n, 3)) * np.array([8.0, 8.0, 3.0])).astype(np.float32)
    off = [i * n for i in range(b + 1)]
    V = _vox()
    tp = _t(pts)
    vb = V.voxelize_batch(tp, off, voxel_size=0.05)
    for s in (0, 3, 7):
        cells, _, _, inv, ind = vb.scene(s)
        seg = tp[s * n:(s + 1) * n]
        allc = torch.floor((seg - seg.min(0).values).double() * (1 / 0.05))
        assert torch.equal(cells[inv], allc)
        first = torch.full((len(ind),), n, dtype=torch.int64, device=DEV)
        first.scatter_reduce_(0, inv, torch.arange(n, device=DEV), reduce="amin")
        assert torch.equal(first, ind)
        keys = V.fnv_hash_vec(cells).cpu().numpy().view(np.uint64)
        assert (keys[1:] > keys[:-1]).all()


def test_prepare_batch_with_augmentation_vs_oracle():
    """Flips + three rotations + min shift + voxelise for a ragged batch, and the device-side pose shift."""
    from situation3d_amd import scene_prep as sp
    rng = np.random.default_rng(21)
    sizes = [30000, 12345, 40000]
    scenes = [(rng.random((n, 3)) * np.array([8.0, 8.0, 3.0]) - np.array([4.0, 4.0, 0.0])).astype(np.float32)
              for n in sizes]
    off = np.concatenate([[0], np.cumsum(sizes)]).tolist()
    augs = [sp.SceneAugmentation(1, [("x", 0.03), ("y", -0.05), ("z", 0.07)]),
            sp.SceneAugmentation(2, [("x", -0.01), ("y", 0.02), ("z", -0.08)]),
            sp.SceneAugmentation(3, [("x", 0.08), ("y", 0.0), ("z", 0.01)])]
    sits = [(rng.normal(size=3), np.array([0.0, 0.0, np.sin(0.3), np.cos(0.3)])) for _ in sizes]
    flat = np.concatenate(scenes)
    feats = rng.random((len(flat), 3)).astype(np.float32)
    labels = rng.integers(0, 20, len(flat))
    vb, aux = sp.prepare_batch(_t(flat), off, _t(feats), _t(labels), sits, augs, voxel_size=0.02)
    assert aux.shape == (3, 7) and aux.dtype == torch.float32
    for s, pts in enumerate(scenes):
        mats = [sp.axis_rotation(ax, t) for ax, t in augs[s].rotations]
        cells, inds, inverse, mins = _ref().prepare_scene(pts, mats, 0.02, flips=augs[s].flips)
        c, f, lab, inv, ind = vb.scene(s)
        assert np.array_equal(vb.mins[s].cpu().numpy(), mins), s
        assert np.array_equal(ind.cpu().numpy(), inds) and np.array_equal(inv.cpu().numpy(), inverse), s
        assert np.array_equal(c.cpu().numpy(), cells), s
        assert np.array_equal(f.cpu().numpy(), feats[off[s]:off[s + 1]][inds]), s
        assert np.array_equal(lab.cpu().numpy(), labels[off[s]:off[s + 1]][inds]), s
        coord, quat = sp.augment_situation(*sits[s], augs[s])
        want = np.concatenate([coord - mins, quat]).astype(np.float32)
        assert np.allclose(aux[s].cpu().numpy(), want, atol=1e-6), s   # f32 output of an f64 subtraction
    # eval split: no augmentation, float32 min shift
    vb, aux = sp.prepare_batch(_t(flat), off, None, None, sits, None, voxel_size=0.02)
    for s, pts in enumerate(scenes):
        cells, inds, inverse, mins = _ref().prepare_scene(pts, (), 0.02)
        c, _, _, inv, ind = vb.scene(s)
        assert np.array_equal(ind.cpu().numpy(), inds) and np.array_equal(c.cpu().numpy(), cells), s

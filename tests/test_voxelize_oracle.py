"""CPU suite: the numpy restatement of the scene-preparation path (oracle/voxelize_ref.py) against
vectors produced by the REFERENCE's own voxelization_utils.py / voxelizer_dev.py
(tests/golden/make_voxelize_golden.py, run in the build container).  Nothing here reads /root/reference.
"""
import os

import numpy as np

from oracle import voxelize_ref as ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "voxelize_golden.npz")


def _g():
    return {k: v for k, v in np.load(GOLD, allow_pickle=False).items()}


def test_fnv_keys_match_reference():
    g = _g()
    assert np.array_equal(ref.fnv_hash_vec(g["fnv_cells"]), g["fnv_keys"])
    assert np.array_equal(ref.fnv_hash_vec(g["fnv_cells4"]), g["fnv_keys4"])
    # known answer by hand: one cell (0,0,0) is three multiplications of the offset basis
    h = 14695981039346656037
    for _ in range(3):
        h = (h * 1099511628211) % (1 << 64)
    assert int(ref.fnv_hash_vec(np.zeros((1, 3)))[0]) == h


def test_prepare_scene_matches_reference_cases():
    g = _g()
    for name in g["cases"]:
        cells, inds, inverse, mins = ref.prepare_scene(g[name + "_points"], g[name + "_rots"],
                                                       float(g[name + "_voxel"]))
        assert np.array_equal(mins, g[name + "_mins"]), name
        assert np.array_equal(inds, g[name + "_inds"]), name
        assert np.array_equal(inverse, g[name + "_inverse"]), name
        assert np.array_equal(cells, g[name + "_cells"]), name
        # properties np.unique guarantees: first occurrence, and the inverse reconstructs every cell
        assert np.array_equal(cells[inverse][inds], cells)
        first = np.full(len(inds), len(inverse))
        np.minimum.at(first, inverse, np.arange(len(inverse)))
        assert np.array_equal(first, inds), name


def test_sparse_quantize_matches_reference():
    g = _g()
    p, labels, feats = g["sq_points"], g["sq_labels"], g["sq_feats"]
    inds, inv = ref.sparse_quantize(p, return_index=True, quantization_size=0.1)
    assert np.array_equal(inds, g["sq_inds"]) and np.array_equal(inv, g["sq_inverse"])
    inds_l, lab = ref.sparse_quantize(p, feats, labels, return_index=True, quantization_size=0.1,
                                      set_ignore_label_when_collision=True)
    assert np.array_equal(inds_l, g["sq_inds_l"]) and np.array_equal(lab, g["sq_labels_l"])
    assert (lab == 255).any()
    c, f = ref.sparse_quantize(p, feats, quantization_size=[0.1, 0.2, 0.3])
    assert np.array_equal(c, g["sq_cells_aniso"]) and np.array_equal(f, g["sq_feats_aniso"])


def test_align_pose_known_answers():
    # identity alignment: position + centre, orientation unchanged
    coord, rot = ref.align_pose([1.0, 2.0, 3.0, 0, 0, 0, 1.0], [0.5, 0.5, 0.5], np.eye(4))
    assert np.allclose(coord, [1.5, 2.5, 3.5]) and np.allclose(rot, np.eye(3))
    # 90 degrees about z plus a translation
    a = np.eye(4)
    a[:3, :3] = [[0, -1, 0], [1, 0, 0], [0, 0, 1]]
    a[:3, 3] = [10, 0, 0]
    coord, rot = ref.align_pose([1.0, 0.0, 0.0, 0, 0, 0, 1.0], [0, 0, 0], a)
    assert np.allclose(coord, [10, 1, 0]) and np.allclose(rot, a[:3, :3])
    from scipy.spatial.transform import Rotation
    q = np.array([0.1, -0.3, 0.2, 0.9])
    assert np.allclose(ref.quat_to_matrix(q), Rotation.from_quat(q).as_matrix(), atol=1e-15)


def test_scene_prep_pose_matches_oracle_and_draw_order():
    """Host half of situation3d_amd.scene_prep (pose alignment + augmentation) against the matrix-form
    restatement; quaternions compared up to sign, 1e-12."""
    from situation3d_amd import scene_prep as sp
    rng = np.random.default_rng(3)
    a = np.eye(4)
    a[:3, :3] = sp.axis_rotation("z", 0.7)
    a[:3, 3] = [0.3, -1.2, 0.05]
    for trial in range(20):
        th = rng.uniform(-np.pi, np.pi)   # SQA3D situations are rotations about the up axis
        q = np.array([0.0, 0.0, np.sin(th / 2), np.cos(th / 2)])
        position = list(rng.normal(size=3)) + list(q)
        centre = rng.normal(size=3)
        coord, quat = sp.align_situation(position, centre, a)
        ocoord, orot = ref.align_pose(position, centre, a)
        assert np.allclose(coord, ocoord, atol=1e-12)
        oq = ref.matrix_to_quat(orot)
        assert min(np.abs(quat - oq).max(), np.abs(quat + oq).max()) < 1e-12
        aug = sp.SceneAugmentation(flips=trial % 4, rotations=[("x", 0.02), ("y", -0.04), ("z", 0.06)][:trial % 4])
        c2, q2 = sp.augment_situation(coord, quat, aug)
        oc2, om2 = ref.augment_pose(ocoord, orot, aug.flips, [sp.axis_rotation(ax, t) for ax, t in aug.rotations])
        assert np.allclose(c2, oc2, atol=1e-12)
        # (the reference's x flip keeps the matrix a rotation only for rotations about z, which is what the
        # aligned SQA3D poses are at that point)
        oq2 = ref.matrix_to_quat(om2)
        assert min(np.abs(q2 - oq2).max(), np.abs(q2 + oq2).max()) < 1e-12
    # draw order: mirror YZ, mirror XZ, rot x, rot y, rot z (sepdataset.py:243-286)
    np.random.seed(7)
    draws = [np.random.random() for _ in range(5)]
    np.random.seed(7)
    aug = sp.SceneAugmentation.sample()
    assert aug.flips == (1 if draws[0] > 0.5 else 0) | (2 if draws[1] > 0.5 else 0)
    assert [ax for ax, _ in aug.rotations] == ["x", "y", "z"]
    assert np.allclose([t for _, t in aug.rotations], [d * np.pi / 18 - np.pi / 36 for d in draws[2:]])

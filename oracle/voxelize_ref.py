"""CPU restatement (numpy, float64 like the reference) of the DataLoader-side scene preparation:
augmentation rotations, min shift, voxelisation, FNV keys, np.unique de-duplication, and the situation
pose's axis alignment.  TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and bench cpu_baseline
legs may import this; the product path (situation3d_amd/voxelizer.py) never does.

Follows
  lib/openscene/voxelization_utils.py:9-24     fnv_hash_vec
  lib/openscene/voxelization_utils.py:27-44    ravel_hash_vec
  lib/openscene/voxelization_utils.py:47-140   sparse_quantize
  lib/openscene/voxelizer_dev.py:15-56         Voxelizer
  lib/sepdataset.py:222-237                    pose axis alignment
  lib/sepdataset.py:241-300                    flips, per-axis rotations, min shift
Pinned against the reference itself: tests/golden/make_voxelize_golden.py imports the two openscene
files from /root/reference in the build container and stores inputs/outputs in
tests/golden/voxelize_golden.npz (the reference has no tests or vectors of its own for this path).
"""
import numpy as np

FNV_OFFSET = np.uint64(14695981039346656037)
FNV_PRIME = np.uint64(1099511628211)


def fnv_hash_vec(arr):
    """voxelization_utils.py:9-24: per column, multiply by the prime THEN xor (wrapping uint64)."""
    assert arr.ndim == 2
    cells = np.array(arr).astype(np.int64).astype(np.uint64)  # floor'd floats; negatives wrap like C
    h = np.full(cells.shape[0], FNV_OFFSET, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for j in range(cells.shape[1]):
            h = h * FNV_PRIME
            h = np.bitwise_xor(h, cells[:, j])
    return h


def ravel_hash_vec(arr):
    """voxelization_utils.py:27-44."""
    assert arr.ndim == 2
    a = np.array(arr, dtype=np.float64)
    a = a - a.min(0)
    a = a.astype(np.uint64)
    amax = a.max(0).astype(np.uint64) + np.uint64(1)
    keys = np.zeros(a.shape[0], dtype=np.uint64)
    for j in range(a.shape[1] - 1):
        keys += a[:, j]
        keys *= amax[j + 1]
    keys += a[:, -1]
    return keys


def _quantization_list(quantization_size, dimension):
    if isinstance(quantization_size, (list, tuple, np.ndarray)):
        assert len(quantization_size) == dimension, "Quantization size and coordinates size mismatch."
        return [q for q in quantization_size]
    if np.isscalar(quantization_size):
        return [quantization_size for _ in range(dimension)]
    raise ValueError("Not supported type for quantization_size.")


def sparse_quantize(coords, feats=None, labels=None, ignore_label=255,
                    set_ignore_label_when_collision=False, return_index=False, hash_type="fnv",
                    quantization_size=1):
    """voxelization_utils.py:47-140 (same argument meaning and return shapes)."""
    use_label = labels is not None
    use_feat = feats is not None
    if not use_label and not use_feat:
        return_index = True
    assert hash_type in ["ravel", "fnv"], \
        "Invalid hash_type. Either ravel, or fnv allowed. You put hash_type=" + hash_type
    assert coords.ndim == 2, \
        "The coordinates must be a 2D matrix. The shape of the input is " + str(coords.shape)
    if use_feat:
        assert feats.ndim == 2 and coords.shape[0] == feats.shape[0]
    if use_label:
        assert coords.shape[0] == len(labels)
    q = _quantization_list(quantization_size, coords.shape[1])
    discrete = np.floor(coords / np.array(q))
    key = ravel_hash_vec(discrete) if hash_type == "ravel" else fnv_hash_vec(discrete)
    if use_label:
        _, inds, counts = np.unique(key, return_index=True, return_counts=True)
        filtered = labels[inds]
        if set_ignore_label_when_collision:
            filtered[counts > 1] = ignore_label
        if return_index:
            return inds, filtered
        return discrete[inds], feats[inds], filtered
    _, inds, inverse = np.unique(key, return_index=True, return_inverse=True)
    if return_index:
        return inds, inverse
    if use_feat:
        return discrete[inds], feats[inds]
    return discrete[inds]


class Voxelizer:
    """voxelizer_dev.py:15-56."""

    def __init__(self, voxel_size=1, ignore_label=255):
        self.voxel_size = voxel_size
        self.ignore_label = ignore_label

    def get_transformation_matrix(self):
        m = np.eye(4)
        np.fill_diagonal(m[:3, :3], 1 / self.voxel_size)
        return m

    def voxelize(self, coords, feats, labels, center=None, link=None, return_ind=False):
        assert coords.shape[1] == 3 and coords.shape[0] == feats.shape[0] and coords.shape[0]
        rigid = self.get_transformation_matrix()
        homo = np.hstack((coords, np.ones((coords.shape[0], 1), dtype=coords.dtype)))
        aug = np.floor(homo @ rigid.T[:, :3])
        assert aug.min(0).sum() == 0, "Minimum of coordinates are not zeros!"
        inds, inverse = sparse_quantize(aug, return_index=True)
        aug, feats, labels = aug[inds], feats[inds], labels[inds]
        if return_ind:
            return aug, feats, labels, np.array(inverse), inds
        if link is not None:
            return aug, feats, labels, np.array(inverse), link[inds]
        return aug, feats, labels, np.array(inverse)


def prepare_scene(points, rotations=(), voxel_size=0.02, flips=0):
    """sepdataset.py:243-302 for one scene: mirror flips (bit 0: x, bit 1: y), p <- p.R^T for every
    rotation in order (float64 np.dot), p <- p - p.min(0), Voxelizer(voxel_size).voxelize(...).
    Returns (cells float64 (U,3), inds, inverse, min_coords)."""
    p = np.array(points)
    if flips & 1:
        p[:, 0] = -1 * p[:, 0]
    if flips & 2:
        p[:, 1] = -1 * p[:, 1]
    for r in rotations:
        p = np.dot(p[:, 0:3], np.transpose(np.asarray(r, dtype=np.float64)))
    mins = p.min(0)
    p = p - mins
    vox = Voxelizer(voxel_size)
    feats = np.zeros((p.shape[0], 1), dtype=np.float32)
    cells, _, _, inverse, inds = vox.voxelize(p, feats, np.zeros(p.shape[0], dtype=np.int64), return_ind=True)
    return cells, inds, inverse, np.asarray(mins, dtype=np.float64)


def quat_to_matrix(q):
    """scipy Rotation.from_quat(q).as_matrix() for xyzw quaternions (normalises first)."""
    x, y, z, w = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def align_pose(position, bs_center, axis_align_matrix):
    """sepdataset.py:222-237: undo the bounding-sphere centring, then the scene's axis alignment, for
    the situation position (homogeneous row vector times A^T) and orientation (A[:3,:3] @ R(q)).
    Returns (coord (3,), rotation matrix (3,3)); the reference converts the matrix back to a quaternion
    with scipy (sign convention of as_quat is scipy's)."""
    coord = np.array(position[:3], dtype=np.float64) + np.asarray(bs_center, dtype=np.float64)
    aug = np.ones((1, 4))
    aug[:, 0:3] = coord
    aug = np.dot(aug, np.asarray(axis_align_matrix, dtype=np.float64).transpose())
    rot = np.dot(np.asarray(axis_align_matrix, dtype=np.float64)[0:3, 0:3], quat_to_matrix(position[3:]))
    return aug[:, 0:3].reshape(-1), rot


def matrix_to_quat(m):
    """Unit quaternion xyzw of a rotation matrix (Shepperd's branch on the largest of trace / diagonal);
    defined up to sign, compare with +-."""
    m = np.asarray(m, dtype=np.float64)
    t = np.trace(m)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = [(m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s]
    else:
        i = int(np.argmax(np.diag(m)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + m[i, i] - m[j, j] - m[k, k]) * 2
        q = [0.0, 0.0, 0.0, (m[k, j] - m[j, k]) / s]
        q[i] = 0.25 * s
        q[j] = (m[j, i] + m[i, j]) / s
        q[k] = (m[k, i] + m[i, k]) / s
    return np.array(q)


def augment_pose(coord, rot, flips, rotations):
    """The pose half of sepdataset.py:243-295 on (position, rotation MATRIX): flips negate a coordinate
    and conjugate the matrix as written there (x flip: m[0,0], m[1,1] negated -- :249-250; y flip: rows
    and columns 0/1 swapped -- :258-259); every rotation r: coord <- coord.r^T, m <- r.m."""
    coord = np.array(coord, dtype=np.float64)
    m = np.array(rot, dtype=np.float64)
    if flips & 1:
        coord[0] = -coord[0]
        m[0, 0] *= -1
        m[1, 1] *= -1
    if flips & 2:
        coord[1] = -coord[1]
        m = m[[1, 0, 2], :][:, [1, 0, 2]]
    for r in rotations:
        r = np.asarray(r, dtype=np.float64)
        coord = np.dot(coord.reshape(1, -1), r.T).reshape(-1)
        m = np.dot(r, m)
    return coord, m

"""Per-batch scene preparation with the point work on the GPU.

The reference's `__getitem__` (lib/sepdataset.py:214-304) does, per sample and in numpy:
  1. situation pose -> scene frame: undo bounding-sphere centring, axis-align (222-237);
  2. augmentation draws, in this order: mirror YZ, mirror XZ, rot x, rot y, rot z, each of +-5 degrees
     (241-295), applied to the points AND to the pose;
  3. points and pose position shifted by the points' minimum (298-300);
  4. voxelisation + de-duplication of the points (302).
Here 1 and the pose half of 2 stay on the host (a handful of 3x3 products per sample, done with the same
scipy.spatial.transform.Rotation calls), while the point half of 2, 3 and 4 run for the whole batch in
`sig3d_voxelize`; the pose position is shifted on the device with the kernel's own minimum, so nothing
waits for the GPU.
"""
import numpy as np
import torch
from scipy.spatial.transform import Rotation

from . import voxelizer

_AXIS_MATS = {
    "x": lambda c, s: [[1, 0, 0], [0, c, -s], [0, s, c]],     # situation3d/utils/pc_utils.py:275-281
    "y": lambda c, s: [[c, 0, s], [0, 1, 0], [-s, 0, c]],     # :283-289
    "z": lambda c, s: [[c, -s, 0], [s, c, 0], [0, 0, 1]],     # :307-313
}


def axis_rotation(axis, angle):
    return np.array(_AXIS_MATS[axis](np.cos(angle), np.sin(angle)), dtype=np.float64)


class SceneAugmentation:
    """One sample's draws: `flips` bit 0 = mirror across YZ (x negated), bit 1 = across XZ (y negated);
    `rotations` = list of (axis, angle) in application order."""

    def __init__(self, flips=0, rotations=()):
        self.flips = int(flips)
        self.rotations = list(rotations)

    @classmethod
    def sample(cls, rng=np.random, no_mirror=False, no_rotx=False, no_roty=False, no_rotz=False):
        """Same draws, in the same order, as sepdataset.py:243-286 (`rng.random()` calls), so seeding
        numpy's global generator identically reproduces the reference's augmentation."""
        flips, rots = 0, []
        if not no_mirror:
            if rng.random() > 0.5:
                flips |= 1
            if rng.random() > 0.5:
                flips |= 2
        for axis, off in (("x", no_rotx), ("y", no_roty), ("z", no_rotz)):
            if not off:
                rots.append((axis, (rng.random() * np.pi / 18) - np.pi / 36))
        return cls(flips, rots)

    def matrices(self, k):
        """(k,3,3) float64, identity-padded so that a batch has one rotation count."""
        out = np.tile(np.eye(3), (k, 1, 1))
        for i, (axis, angle) in enumerate(self.rotations):
            out[i] = axis_rotation(axis, angle)
        return out


def align_situation(position, bs_center, axis_align_matrix):
    """sepdataset.py:222-237 -> (coord (3,) f64, quaternion xyzw (4,) f64) in the axis-aligned frame."""
    a = np.asarray(axis_align_matrix, dtype=np.float64)
    homo = np.ones((1, 4))
    homo[0, :3] = np.asarray(position[:3], dtype=np.float64) + np.asarray(bs_center, dtype=np.float64)
    coord = (homo @ a.T)[0, :3]
    rot = a[:3, :3] @ Rotation.from_quat(np.asarray(position[3:], dtype=np.float64)).as_matrix()
    return coord, Rotation.from_matrix(rot).as_quat()


def augment_situation(coord, quat, aug):
    """The pose half of sepdataset.py:243-295 for one sample."""
    coord = np.array(coord, dtype=np.float64)
    quat = np.array(quat, dtype=np.float64)
    if aug.flips & 1:   # :246-251
        coord[0] = -coord[0]
        m = Rotation.from_quat(quat).as_matrix()
        m[0, 0] *= -1
        m[1, 1] *= -1
        quat = Rotation.from_matrix(m).as_quat()
    if aug.flips & 2:   # :255-261
        coord[1] = -coord[1]
        m = Rotation.from_quat(quat).as_matrix()
        m = m[[1, 0, 2], :][:, [1, 0, 2]]
        quat = Rotation.from_matrix(m).as_quat()
    for axis, angle in aug.rotations:   # :264-295
        r = axis_rotation(axis, angle)
        coord = (coord.reshape(1, -1) @ r.T).reshape(-1)
        quat = Rotation.from_matrix(r @ Rotation.from_quat(quat).as_matrix()).as_quat()
    return coord, quat


def prepare_batch(coords, offsets, feats, labels, situations, augmentations=None, voxel_size=0.02):
    """coords/feats/labels: flat GPU tensors of the raw axis-aligned scenes; situations: list of
    (coord, quat) from align_situation; augmentations: list of SceneAugmentation or None (eval split).
    -> (VoxelBatch, auxiliary_task (B,7) f32 on the device: position - min_coords | quaternion),
    the `__quat__` layout of sepdataset.py:306-307."""
    b = len(situations)
    flips = rots = None
    poses = np.zeros((b, 7), dtype=np.float64)
    if augmentations is not None:
        k = max(1, max(len(a.rotations) for a in augmentations))
        rots = np.stack([a.matrices(k) for a in augmentations])
        flips = [a.flips for a in augmentations]
    for i, (coord, quat) in enumerate(situations):
        if augmentations is not None:
            coord, quat = augment_situation(coord, quat, augmentations[i])
        poses[i, :3], poses[i, 3:] = coord, quat
    vb = voxelizer.voxelize_batch(coords, offsets, feats, labels, rotations=rots, flips=flips,
                                  voxel_size=voxel_size)
    aux = torch.from_numpy(poses).to(coords.device, non_blocking=True)
    aux[:, :3] -= vb.mins
    return vb, aux.float()

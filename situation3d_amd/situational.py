"""Situational pose re-encode: host side of sig3d_situational_transform (+grad).

Reference math:
  * pose (B,7) = [t, q_xyzw] -> 4x4 and p' = [p,1] M^T  (situation3d/utils/temp.py:42-97);
  * Gaussian localisation target around the GT position, sigma = 0.16 m
    (situation3d/models/sqa_module.py:328-338);
  * unit-quaternion and rotation-vector helpers that the reference defines but never calls
    (sqa_module.py:12-64) are provided as thin torch functions for API completeness.
"""
import torch

from . import _lib


class _SituationalTransform(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pose, points, inverse):
        dev = _lib.require_device(pose, points)
        if pose.dtype != torch.float32 or points.dtype != torch.float32:
            raise RuntimeError("pose and points must be float tensors")
        pose, points = pose.contiguous(), points.contiguous()
        b, n, _ = points.shape
        out = torch.empty_like(points)
        with torch.cuda.device(dev):
            _lib.call("sig3d_situational_transform", b, n, _lib.ptr(pose), _lib.ptr(points),
                      _lib.ptr(out), int(inverse), _lib.stream_ptr(dev))
        ctx.save_for_backward(pose, points)
        ctx.inverse = int(inverse)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        pose, points = ctx.saved_tensors
        b, n, _ = points.shape
        grad_out = grad_out.contiguous()
        grad_points = torch.empty_like(points)
        grad_pose = torch.empty_like(pose)
        with torch.cuda.device(points.device):
            _lib.call("sig3d_situational_transform_grad", b, n, _lib.ptr(pose), _lib.ptr(points),
                      _lib.ptr(grad_out), _lib.ptr(grad_points), _lib.ptr(grad_pose), ctx.inverse,
                      _lib.stream_ptr(points.device))
        return grad_pose, grad_points, None


def situational_transform(pose, points, inverse=False):
    """points (B,N,3) -> R(q) p + t  (temp.py:86-97), or the agent-frame map R(q)^T (p - t) when
    `inverse`.  pose (B,7) = [tx,ty,tz,qx,qy,qz,qw]; R is temp.py:63-73's x2-y2-z2+w2 form."""
    return _SituationalTransform.apply(pose, points, bool(inverse))


def gaussian_localisation_target(scene_positions, gt_translation, sigma=0.16):
    """sqa_module.py:328-338: weights ~ exp(-|p_xy - t_xy|^2 / (2 sigma^2)), normalised per scene.
    On the GPU (no gradient wanted: it is a target) one launch of csrc/sqa_loss.hip instead of seven torch kernels."""
    if (scene_positions.is_cuda and scene_positions.dtype == torch.float32 and gt_translation.dtype == torch.float32
            and not (torch.is_grad_enabled() and (scene_positions.requires_grad or gt_translation.requires_grad))):
        import ctypes
        p, t = scene_positions.contiguous(), gt_translation.contiguous()
        out = torch.empty(p.shape[:2], dtype=torch.float32, device=p.device)
        with torch.cuda.device(p.device):
            _lib.call("sig3d_gaussian_target", p.shape[0], p.shape[1], p.shape[2], ctypes.c_float(sigma), _lib.ptr(p),
                      _lib.ptr(t), t.shape[1], _lib.ptr(out), _lib.stream_ptr(p.device))
        return out
    d = torch.norm(scene_positions[..., :2] - gt_translation[:, None, :2], dim=2)
    w = torch.exp(-d ** 2 / (2 * sigma ** 2))
    return w / w.sum(dim=1, keepdim=True)


def quaternions_to_rotation_matrices(q):
    """sqa_module.py:12-30 (scipy order x,y,z,w; equals temp.py's matrix for unit quaternions)."""
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rows = [
        torch.stack([1 - 2 * (y ** 2 + z ** 2), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
        torch.stack([2 * (x * y + z * w), 1 - 2 * (x ** 2 + z ** 2), 2 * (y * z - x * w)], -1),
        torch.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x ** 2 + y ** 2)], -1),
    ]
    return torch.stack(rows, 1)

"""oracle/pointnet2_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end of oracle/pointnet2_oracle.c with the same nine-function surface as the
reference's pybind module (lib/pointnet2/_ext_src/src/bindings.cpp:6-19), operating on CPU
torch tensors.  Host-side allocation/initialisation follows the reference wrappers
(torch::zeros outputs, FPS temp filled with 1e10: sampling.cpp:70-76, ball_query.cpp:19-21,
group_points.cpp:21-23,47-49, interpolate.cpp:24-29,55-57,84-86).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

_F = ctypes.POINTER(ctypes.c_float)
_I = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    """Compile the C restatement with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "pointnet2_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _fp(t):
    assert t.dtype == torch.float32 and t.is_contiguous() and t.device.type == "cpu"
    return ctypes.cast(t.data_ptr(), _F)


def _ip(t):
    assert t.dtype == torch.int32 and t.is_contiguous() and t.device.type == "cpu"
    return ctypes.cast(t.data_ptr(), _I)


def num_threads():
    return int(lib().oracle_num_threads())


def opt_n_threads(work_size):
    return int(lib().oracle_opt_n_threads(int(work_size)))


# ---- the nine functions of bindings.cpp:6-19 (positional, same order) -------------------

def gather_points(points, idx):
    b, c, n = points.shape
    m = idx.shape[1]
    out = torch.zeros(b, c, m, dtype=torch.float32)
    lib().oracle_gather_points(b, c, n, m, _fp(points), _ip(idx), _fp(out))
    return out


def gather_points_grad(grad_out, idx, n):
    b, c, m = grad_out.shape
    out = torch.zeros(b, c, n, dtype=torch.float32)
    lib().oracle_gather_points_grad(b, c, n, m, _fp(grad_out), _ip(idx), _fp(out))
    return out


def furthest_point_sampling(points, nsamples):
    b, n, _ = points.shape
    out = torch.zeros(b, nsamples, dtype=torch.int32)
    tmp = torch.full((b, n), 1e10, dtype=torch.float32)
    lib().oracle_furthest_point_sampling(b, n, int(nsamples), _fp(points), _fp(tmp), _ip(out))
    return out


def three_nn(unknowns, knows):
    b, n, _ = unknowns.shape
    m = knows.shape[1]
    idx = torch.zeros(b, n, 3, dtype=torch.int32)
    dist2 = torch.zeros(b, n, 3, dtype=torch.float32)
    lib().oracle_three_nn(b, n, m, _fp(unknowns), _fp(knows), _fp(dist2), _ip(idx))
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    b, c, m = points.shape
    n = idx.shape[1]
    out = torch.zeros(b, c, n, dtype=torch.float32)
    lib().oracle_three_interpolate(b, c, m, n, _fp(points), _ip(idx), _fp(weight), _fp(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    b, c, n = grad_out.shape
    out = torch.zeros(b, c, m, dtype=torch.float32)
    lib().oracle_three_interpolate_grad(b, c, n, m, _fp(grad_out), _ip(idx), _fp(weight),
                                        _fp(out))
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    idx = torch.zeros(b, m, nsample, dtype=torch.int32)
    lib().oracle_ball_query(b, n, m, ctypes.c_float(radius), int(nsample), _fp(new_xyz),
                            _fp(xyz), _ip(idx))
    return idx


def group_points(points, idx):
    b, c, n = points.shape
    _, npoints, nsample = idx.shape
    out = torch.zeros(b, c, npoints, nsample, dtype=torch.float32)
    lib().oracle_group_points(b, c, n, npoints, nsample, _fp(points), _ip(idx), _fp(out))
    return out


def group_points_grad(grad_out, idx, n):
    b, c, npoints, nsample = grad_out.shape
    out = torch.zeros(b, c, n, dtype=torch.float32)
    lib().oracle_group_points_grad(b, c, n, npoints, nsample, _fp(grad_out), _ip(idx),
                                   _fp(out))
    return out


# ---- situational transform (situation3d/utils/temp.py:42-97) ----------------------------

def pose_to_matrix(pose):
    b = pose.shape[0]
    mat = torch.zeros(b, 4, 4, dtype=torch.float32)
    lib().oracle_pose_to_matrix(b, _fp(pose), _fp(mat))
    return mat


def situational_transform(pose, points):
    b, n, _ = points.shape
    out = torch.zeros(b, n, 3, dtype=torch.float32)
    lib().oracle_situational_transform(b, n, _fp(pose), _fp(points), _fp(out))
    return out

"""`pointnet2._ext`: same nine functions as lib/pointnet2/_ext_src/src/bindings.cpp:6-19."""
from situation3d_amd.pointnet2._ext import (  # noqa: F401
    ball_query,
    furthest_point_sampling,
    gather_points,
    gather_points_grad,
    group_points,
    group_points_grad,
    three_interpolate,
    three_interpolate_grad,
    three_nn,
)

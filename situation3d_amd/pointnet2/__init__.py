"""Host-side mirror of the reference's lib/pointnet2 package over the gfx950 HIP kernels."""

"""Drop-in import name of the reference's native extension package.

The reference does `import pointnet2._ext as _ext` (lib/pointnet2/pointnet2_utils.py:25-33;
package built by lib/pointnet2/setup.py:22-28).  With /root/repo on sys.path that import
resolves here and lands on the gfx950 HIP kernels, so the reference's own
pointnet2_utils.py / pointnet2_modules.py run unchanged on an MI355X.
"""
__version__ = "3.0.0"  # lib/pointnet2/_version.py:1

import sys, runpy
sys.path.insert(0, '.')
import situation3d_amd.pointnet2.fused_mlp as f
f.MIN_POSITIONS = int(sys.argv.pop(1))
sys.argv[0] = 'bench.py'
runpy.run_path('bench.py', run_name='__main__')

import sys, torch, ctypes
sys.path.insert(0,'.')
from situation3d_amd.pointnet2 import _ext
from situation3d_amd import _lib as L
import bench
dev=torch.device('cuda',0)
b,n,m=8,40000,2048
batch=bench.synthetic_batch(b,n,3,dev)
xyz=batch['point_clouds'][...,:3].contiguous()
inds=_ext.furthest_point_sampling(xyz, m)
new_xyz=torch.gather(xyz,1,inds.long().unsqueeze(-1).expand(-1,-1,3)).contiguous()
def t(fn, it=10):
    for _ in range(2): fn()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/it*1e3
idx=torch.empty(b,m,64,dtype=torch.int32,device=dev)
work=torch.empty(_ext.ball_query_workspace_bytes(b,n),dtype=torch.uint8,device=dev)
tg=t(lambda: L.call("sig3d_ball_query_grid", b,n,m,ctypes.c_float(0.2),64,L.ptr(new_xyz),L.ptr(xyz),L.ptr(idx),L.ptr(work),work.numel(),L.stream_ptr()))
a=idx.clone()
tb=t(lambda: L.call("sig3d_ball_query", b,n,m,ctypes.c_float(0.2),64,L.ptr(new_xyz),L.ptr(xyz),L.ptr(idx),L.stream_ptr()))
print("SA1 ball query: grid %.1f us, brute force %.1f us, equal=%s, avg hits/centre=%.1f" % (tg,tb,torch.equal(a,idx),(idx!=idx[...,:1]).float().sum(-1).mean().item()+1))

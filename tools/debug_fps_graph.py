import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import pointnet2._ext as ext
from util import scene
dev = "cuda:0"
xs = [scene(8, 40000, seed=s).to(dev) for s in range(4)]
ref = [ext.furthest_point_sampling(x, 2048).clone() for x in xs]
torch.cuda.synchronize()
static = xs[0].clone()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): ext.furthest_point_sampling(static, 2048)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = ext.furthest_point_sampling(static, 2048)
    new = ext.gather_points(static.transpose(1, 2).contiguous(), out)
    idx = ext.ball_query(new.transpose(1, 2).contiguous(), static, 0.2, 64)
torch.cuda.synchronize()
for i in range(8):
    static.copy_(xs[i % 4]); g.replay(); torch.cuda.synchronize()
    r = ref[i % 4]
    print(i, "fps equal:", torch.equal(out, r), "min", int(out.min()), "max", int(out.max()), "bq max", int(idx.max()), "bq min", int(idx.min()))

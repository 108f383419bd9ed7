import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import bench
from situation3d_amd.geometry import GeometryPlan
from situation3d_amd.model import PointNet2Encoder
from util import scene
dev = torch.device("cuda:0")
for name, xyz in (("uniform", bench.synthetic_batch(8, 40000, 5, dev)["point_clouds"][..., :3].contiguous()),
                  ("surface", bench.synthetic_batch(8, 40000, 5, dev, surface=True)["point_clouds"][..., :3].contiguous()),
                  ("dups+zero tail", scene(8, 40000, seed=3, dup=4000, zero_tail=500).to(dev))):
    plan = GeometryPlan(8, 40000, PointNet2Encoder.LEVELS, dev).compute(xyz)
    torch.cuda.synchronize()
    for lvl in (1, 2, 3):
        m = plan.inds[lvl].shape[1]
        ar = torch.arange(m, device=dev, dtype=torch.int32)[None].expand(8, -1)
        eq = (plan.inds[lvl] == ar)
        print(name, "level", lvl + 1, "prefix property holds for", int(eq.all(1).sum()), "of 8 scenes; first mismatch at", [int((~e).nonzero()[0]) if not e.all() else -1 for e in eq])

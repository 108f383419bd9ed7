"""SA3 / SA4 alone (library-GEMM hybrid), forward + backward at the BASELINE config-3 shapes, for
`rocprofv3 --kernel-trace --stats`: which launches make up the 1.3 ms the two small levels cost per step?"""
import sys
import torch
sys.path.insert(0, '.')
from situation3d_amd import gemm_tuning
from situation3d_amd.pointnet2 import fused_mlp
from situation3d_amd.pointnet2.pointnet2_modules import PointnetSAModuleVotes
dev = torch.device("cuda", 0)
gemm_tuning.enable(tune_missing=True)
torch.manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else "sa3"
if which == "sa3":
    n, m, r, ns = 1024, 512, 0.8, 16
else:
    n, m, r, ns = 512, 256, 1.2, 16
b, c = 8, 256
sa = PointnetSAModuleVotes(npoint=m, radius=r, nsample=ns, mlp=[c, 128, 128, 256], use_xyz=True, normalize_xyz=True).to(dev).train()
sa.emit_point_major = True
xyz = torch.rand(b, n, 3, device=dev) * torch.tensor([8.0, 8.0, 3.0], device=dev)
feats = torch.randn(b, c, n, device=dev, requires_grad=True)
feats_pm = feats.detach().transpose(1, 2).contiguous().requires_grad_(True)
G = torch.randn(b, m, 256, device=dev)
for it in range(12):
    f = feats.detach().requires_grad_(True)
    f._pm = feats_pm
    _, out, _ = sa(xyz, f)
    (fused_mlp.point_major_of(out) * G).sum().backward()
torch.cuda.synchronize()
print("done", which)

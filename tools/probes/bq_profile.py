"""rocprofv3 --kernel-trace --stats -- python3 tools/probes/bq_profile.py [surface]: per-kernel times of the multi-level
ball query (eager launches, 10 repetitions) + how many centres overflow the 256-slot lists."""
import sys
import torch
sys.path.insert(0, '.')
import bench
from situation3d_amd import _lib as L
from situation3d_amd.pointnet2 import _ext
dev = torch.device('cuda', 0)
b, n = 8, 40000
surface = "surface" in sys.argv
batch = bench.synthetic_batch(b, n, 3, dev, surface=surface)
cur = batch['point_clouds'][..., :3].contiguous()
recs = []
for m, r, ns in [(2048, 0.2, 64), (1024, 0.4, 32), (512, 0.8, 16), (256, 1.2, 16)]:
    inds = _ext.furthest_point_sampling(cur, m)
    nxt = torch.gather(cur, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    recs.append((cur, nxt, r, ns, torch.empty(b, m, ns, dtype=torch.int32, device=dev)))
    cur = nxt
only = [int(a) for a in sys.argv[1:] if a.isdigit()]
if only:
    recs = [recs[i] for i in only]
arr = L.bq_levels(recs)
work = torch.zeros(L.bq_levels_workspace_bytes(b, arr), dtype=torch.uint8, device=dev)
for _ in range(10):
    L.call("sig3d_ball_query_levels", b, len(arr), arr, L.ptr(work), work.numel(), L.stream_ptr())
torch.cuda.synchronize()
slots = sum(b * r[1].shape[1] for r in recs)
cnt = work[:4 * slots].view(torch.int32)
off = 0
for r in recs:
    c = cnt[off:off + b * r[1].shape[1]]
    off += c.numel()
    print("n=%d m=%d r=%.1f: hits per centre mean %.1f max %d, centres over 256: %d"
          % (r[0].shape[1], r[1].shape[1], r[2], c.float().mean().item(), c.max().item(), int((c > 256).sum())))

"""Can RCCL all-reduces be captured into a hipGraph on this stack (torch 2.10 + ROCm 7)?  World size 1 probe,
stepwise closer to what the data-parallel step would capture: (1) one in-place SUM; (2) ReduceOp.AVG, async
handles and per-bucket waits on slices of a flat buffer; (3) the same with an autograd backward between launch
and wait.  Prints which stage works."""
import os
import sys
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 1
flat = torch.ones(64 << 20, device="cuda")
w = torch.randn(512, 512, device="cuda", requires_grad=True)
dist.all_reduce(flat[:1024], op=dist.ReduceOp.AVG)   # communicator set up outside the capture
torch.cuda.synchronize()
s = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    xin = torch.randn(256, 512, device="cuda")
    (xin @ w).sum().backward()
    w.grad = None
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        y = flat * 2.0
        if stage == 1:
            dist.all_reduce(y)
            z = y + 1.0
        else:
            handles = []
            chunk = 16 << 20
            for off in range(0, y.numel(), chunk):
                handles.append(dist.all_reduce(y[off:off + chunk], op=dist.ReduceOp.AVG, async_op=True))
            if stage >= 3:
                (xin @ w).sum().backward()
            for h in handles:
                h.wait()
            z = y + 1.0
    for _ in range(3):
        g.replay()
torch.cuda.synchronize()
print("stage %d captured and replayed: z[0] = %g" % (stage, float(z[0])))
dist.destroy_process_group()

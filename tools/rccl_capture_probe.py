"""Can an RCCL all-reduce be captured into a hipGraph on this stack (torch 2.10 + ROCm 7)?  World size 1 probe:
if yes, the whole data-parallel step (backward + bucketed all-reduce + AdamW) could be ONE graph."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.ones(1 << 20, device="cuda")
dist.all_reduce(x)            # communicator set up outside the capture
torch.cuda.synchronize()
s = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            y = x * 2.0
            dist.all_reduce(y)
            z = y + 1.0
        for _ in range(3):
            g.replay()
    torch.cuda.synchronize()
    print("captured and replayed: z[0] =", float(z[0]))
except Exception as e:  # noqa: BLE001
    print("capture failed:", type(e).__name__, str(e)[:300])
dist.destroy_process_group()

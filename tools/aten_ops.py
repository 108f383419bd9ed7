"""Which ATen (non-hand-written) GPU kernels does one eager training step launch, and from where?"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from situation3d_amd import gemm_tuning
from situation3d_amd.model import SIG3DQFormer
from situation3d_amd.trainer import build_optimizer, train_step
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
gemm_tuning.enable(tune_missing=False)
torch.manual_seed(0)
model = SIG3DQFormer(num_answers=bench.NUM_ANSWERS).to(dev).train()
opt = build_optimizer(model, name="flat_adamw")
batch = bench.synthetic_batch(bench.BATCH, bench.N_POINTS, 7, dev)
for _ in range(3):
    train_step(model, opt, dict(batch))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    train_step(model, opt, dict(batch))
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=6) if e.device_time_total > 0 and e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.count)
for e in rows[:60]:
    stack = [s for s in e.stack if "situation3d_amd" in s or "bench" in s][:2]
    print("%4d x %-28s %7.1f us  %s" % (e.count, e.key, e.device_time_total, " <- ".join(s.split("/")[-1] for s in stack)))

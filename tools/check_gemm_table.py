"""Does the committed GEMM solution table cover every GEMM of the bench step?  (Missing shapes are tuned during
warm-up, ~1 s each; rerun tools/tune_gemms.py when this prints new entries.)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from situation3d_amd import gemm_tuning
from situation3d_amd.model import SIG3DQFormer
from situation3d_amd.trainer import build_optimizer, train_step
import torch.cuda.tunable as tunable
dev = torch.device("cuda", 0)
n0 = gemm_tuning.enable(tune_missing=True)
torch.manual_seed(0)
model = SIG3DQFormer(num_answers=bench.NUM_ANSWERS).to(dev).train()
opt = build_optimizer(model, name="flat_adamw")
batch = bench.synthetic_batch(bench.BATCH, bench.N_POINTS, 7, dev)
with torch.cuda.stream(torch.cuda.Stream(dev)):
    for _ in range(2):
        train_step(model, opt, dict(batch))
torch.cuda.synchronize()
res = tunable.get_results()
print("entries loaded: %d, after two steps: %d" % (n0, len(res)))
for r in res[n0:]:
    print("  new:", r)

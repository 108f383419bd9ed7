"""Host-side enqueue time vs GPU time of the data-parallel step forms at world size 1 (no process group):
is the step bound by the GPU or by the Python / hipGraphLaunch thread?   python tools/dp_host_probe.py [cut]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) > 1:
    os.environ["SIG3D_QF_CUT"] = sys.argv[1]
import torch  # noqa: E402

import bench  # noqa: E402
from situation3d_amd import gemm_tuning  # noqa: E402
from situation3d_amd.ddp import GradBucketReducer, init_distributed  # noqa: E402
from situation3d_amd.graph_step import GraphedTrainStep  # noqa: E402
from situation3d_amd.model import SIG3DQFormer  # noqa: E402
from situation3d_amd.trainer import build_optimizer  # noqa: E402

init_distributed()   # SIG3D_SINGLE_RANK_PG=1: a real RCCL group of one
dev = torch.device("cuda:0")
gemm_tuning.enable(tune_missing=False)
torch.manual_seed(1234)
model = SIG3DQFormer(num_answers=bench.NUM_ANSWERS).to(dev).train()
opt = build_optimizer(model, name="flat_adamw")
red = GradBucketReducer.from_flat(opt.flat_grad_buffers())
batches = [bench.synthetic_batch(bench.BATCH, bench.N_POINTS, 1234 + i, dev) for i in range(4)]
work = torch.cuda.Stream(dev)
with torch.cuda.stream(work):
    g = GraphedTrainStep(model, opt, batches[0], prefetch_geometry=True, reducer=red)
    for i in range(5):
        g(batches[i % 4], batches[(i + 1) % 4])
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for i in range(30):
        a = time.perf_counter()
        g(batches[i % 4], batches[(i + 1) % 4])
        host.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print("cut=%s graphs=%d: host enqueue %.2f ms/step (min %.2f, max %.2f), wall %.2f ms/step, drain after the last enqueue %.2f ms"
      % (os.environ.get("SIG3D_QF_CUT", "default"), 2 + (g.graph_low is not None) if g._split else 1,
         1e3 * sum(host) / len(host), 1e3 * min(host), 1e3 * max(host), 1e3 * (t2 - t0) / 30, 1e3 * (t2 - t1)))

"""ms per graphed training step with a VARIANT build of the library (tools only: the product loads its own):
python tools/probes/step_time_lib.py [path/to/libvariant.so] -- e.g. tools/probes/libsig3d_fps_probe.so (sampling.hip built
with -DSIG3D_FPS_PROBE=1: the cooperative FPS sweeps half of its points per round; wrong indices, same launch structure)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from situation3d_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
import bench
from situation3d_amd import gemm_tuning
from situation3d_amd.graph_step import GraphedTrainStep
from situation3d_amd.model import SIG3DQFormer
from situation3d_amd.trainer import build_optimizer
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
gemm_tuning.enable(tune_missing=True)
torch.manual_seed(1234)
model = SIG3DQFormer(num_answers=bench.NUM_ANSWERS).to(dev).train()
opt = build_optimizer(model, name="flat_adamw")
batches = [bench.synthetic_batch(bench.BATCH, bench.N_POINTS, 1234 + i, dev) for i in range(4)]
depth = int(os.environ.get("SIG3D_GEO_DEPTH", "3"))
work = torch.cuda.Stream(dev)
with torch.cuda.stream(work):
    g = GraphedTrainStep(model, opt, batches[0], prefetch_geometry=True, prefetch_depth=depth)
    def run(n, k0):
        for i in range(n):
            k = k0 + i
            g(batches[k % 4], upcoming=[batches[(k + 1 + j) % 4] for j in range(depth)])
    run(20, 0)
    torch.cuda.synchronize()
    res = []
    for rep in range(5):
        t0 = time.perf_counter()
        run(40, 20 + 40 * rep)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 40 * 1e3)
print("%s: ms per step %s  median %.3f" % (os.path.basename(_lib.LIB_PATH), " ".join("%.3f" % r for r in res), sorted(res)[2]))

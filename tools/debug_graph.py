import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from situation3d_amd.model import SIG3DQFormer
from situation3d_amd.trainer import build_optimizer, train_step
from situation3d_amd.graph_step import GraphedTrainStep
dev = torch.device("cuda:0")
mode = sys.argv[1]
torch.manual_seed(1234)
model = SIG3DQFormer(num_answers=bench.NUM_ANSWERS).to(dev).train()
opt = build_optimizer(model)
batches = [bench.synthetic_batch(8, 40000, 1234 + i, dev) for i in range(4)]
if mode == "graph":
    g = GraphedTrainStep(model, opt, batches[0])
    step = lambda i: g(batches[i % 4])
else:
    step = lambda i: train_step(model, opt, dict(batches[i % 4]))
for i in range(14):
    l = step(i)
    if mode == "graph":
        torch.cuda.synchronize()
        sp = g.static_out["scene_positions"]
        print("   scene_positions finite:", bool(torch.isfinite(sp).all()), float(sp.abs().max()))
    bad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    print(i, float(l.item()), "nonfinite params:", bad[:3], flush=True)

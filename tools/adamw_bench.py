"""Time FlatAdamW.step() (sig3d_adamw_table) on the real model for several chunk sizes.
python tools/adamw_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from situation3d_amd import optim, _lib
if len(sys.argv) > 1:      # a variant build of the library (tools only)
    _lib.LIB_PATH = sys.argv[1]
from situation3d_amd.model import SIG3DQFormer

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = SIG3DQFormer(num_answers=706).to(dev)
n = sum(p.numel() for p in model.parameters())
for chunk in (65536, 262144):
    optim._CHUNK = chunk
    opt = optim.FlatAdamW([{"params": list(model.parameters()), "weight_decay": 0.05}], lr=2e-5)
    for p in model.parameters():
        p.grad = torch.randn_like(p)
    grads = [p.grad for p in model.parameters()]
    ts = []
    for it in range(12):
        for p, g in zip(model.parameters(), grads):
            p.grad = g
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); opt.step(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    t = sorted(ts[2:])[len(ts[2:]) // 2]
    print("chunk %8d: %.3f ms  %.2f TB/s (28 B/param, %d params)" % (chunk, t, 28 * n / t / 1e9, n))
# reference: plain device copy of the same byte volume
a = torch.empty(28 * n // 8, dtype=torch.float32, device=dev); b = torch.empty_like(a)
for _ in range(3): b.copy_(a)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): b.copy_(a)
e.record(); torch.cuda.synchronize()
print("copy of the same volume: %.3f ms" % (s.elapsed_time(e) / 10))

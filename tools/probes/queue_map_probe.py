"""Which torch streams share a hardware queue?  (situation3d_amd.streams.run_concurrently)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from situation3d_amd import streams
dev = torch.device("cuda", 0)
ss = [torch.cuda.Stream(dev) for _ in range(10)]
print("cuda_stream handles:", [hex(s.cuda_stream)[-6:] for s in ss])
for i in range(10):
    print(i, "".join("X" if i == j else ("." if streams.run_concurrently(ss[i], ss[j], dev, 500) else "S") for j in range(10)))
print("default stream vs each:", "".join("." if streams.run_concurrently(torch.cuda.default_stream(dev), s, dev, 500) else "S" for s in ss))
hp = [torch.cuda.Stream(dev, priority=-1) for _ in range(4)]
print("high-priority streams vs normal 0..9:")
for h in hp:
    print("  ", "".join("." if streams.run_concurrently(h, s, dev, 500) else "S" for s in ss))

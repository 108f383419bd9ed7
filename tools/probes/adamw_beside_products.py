"""Does the flat AdamW update (HBM bound) hide beside a layer-batched weight-gradient product (matrix-core bound)?
Two streams, no graph, no fork inside a graph: the product on one, the update of 56.6 M parameters (one weight kind of
the Q-Former) on the other, against the same two launches on one stream."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from situation3d_amd import optim

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
n = 24 * 768 * 3072
p = torch.nn.Parameter(torch.randn(n, device=dev))
g = torch.randn(n, device=dev)
A = torch.randn(24, 256, 768, device=dev)
B = torch.randn(24, 256, 3072, device=dev)
out = torch.empty(24, 768, 3072, device=dev)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def gemm():
    torch.bmm(A.transpose(1, 2), B, out=out)


for wgs in (None, 64, 128, 192, 256, 512):
    opt = optim.FlatAdamW([{"params": [p], "weight_decay": 0.05}], lr=1e-5, max_workgroups=wgs)

    def adamw():
        p.grad = g
        opt.step()

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s1)
        for _ in range(reps):
            fn()
        s1.wait_stream(s2)
        b.record(s1)
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3

    def only_gemm():
        with torch.cuda.stream(s1):
            gemm()

    def only_adamw():
        with torch.cuda.stream(s1):
            adamw()

    def sequential():
        with torch.cuda.stream(s1):
            gemm(); adamw()

    def beside():
        s2.wait_stream(s1)
        with torch.cuda.stream(s1):
            gemm()
        with torch.cuda.stream(s2):
            adamw()
        s1.wait_stream(s2)

    print("update on %s workgroups: product %.0f us, update %.0f us, one stream %.0f us, two streams %.0f us"
          % (wgs or "all", timed(only_gemm), timed(only_adamw), timed(sequential), timed(beside)))

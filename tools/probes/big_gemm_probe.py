"""Which library GEMM forms break when an operand or the result passes 2 GiB?  (config-5 backward, Nk = 80 000:
the stacked K/V projection of the six cross layers is 80 000 x 9216 floats = 2.95 GB.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from situation3d_amd import gemm_tuning
if "--tuned" in sys.argv:
    gemm_tuning.enable(tune_missing=True)
dev = "cuda:0"
torch.manual_seed(0)
n, c, w = 80000, 1408, 9216
enc = torch.randn(n, c, device=dev)
W = torch.randn(w, c, device=dev) * 0.02
bias = torch.randn(w, device=dev)
out = torch.empty(n, w, device=dev)
torch.addmm(bias, enc, W.t(), out=out)
ref = torch.empty_like(out)
for r0 in range(0, n, 16384):
    for c0 in range(0, w, 1536):
        ref[r0:r0 + 16384, c0:c0 + 1536] = torch.addmm(bias[c0:c0 + 1536], enc[r0:r0 + 16384], W[c0:c0 + 1536].t())
print("addmm out= 2.95 GB: nan", bool(torch.isnan(out).any()), "max diff", float((out - ref).abs().max()))
bad = ((out - ref).abs() > 1e-2).nonzero()
if bad.numel():
    print("  first bad", bad[0].tolist(), "last bad", bad[-1].tolist(), "count", bad.shape[0])
# strided A operand (column block of the 2.95 GB matrix) in dX = dkv[:, cols] @ W[cols]
dkv = torch.randn(n, w, device=dev)
cols = slice(3 * 1536, 6 * 1536)
g1 = dkv[:, cols].mm(W[cols])
g2 = dkv[:, cols].contiguous().mm(W[cols])
print("strided-A mm: nan", bool(torch.isnan(g1).any()), "max diff", float((g1 - g2).abs().max()))
# dW = dkv[:, cols]^T @ enc with out= into a slice
gw = torch.empty(w, c, device=dev)
torch.mm(dkv[:, cols].t(), enc, out=gw[cols])
gw2 = dkv[:, cols].contiguous().t().mm(enc)
print("strided-A^T mm out=: nan", bool(torch.isnan(gw[cols]).any()), "rel diff",
      float((gw[cols] - gw2).abs().max() / gw2.abs().max()))
cs = torch.sum(dkv[:, cols], dim=0)
print("colsum ok", float((cs - dkv[:, cols].contiguous().sum(0)).abs().max()))

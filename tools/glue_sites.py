"""Which lines of this package issue the small torch copy / fill / cat / elementwise launches of one eager
training step?  Python-level tracing (torch's profiler does not resolve stacks on this stack): every call of a
handful of torch entry points is attributed to the innermost frame inside situation3d_amd/ or bench.py.

python tools/glue_sites.py
"""
import collections, os, sys, traceback
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench
from situation3d_amd import gemm_tuning
from situation3d_amd.model import SIG3DQFormer
from situation3d_amd.trainer import build_optimizer, train_step

dev = torch.device("cuda", 0)
gemm_tuning.enable(tune_missing=False)
torch.manual_seed(0)
model = SIG3DQFormer(num_answers=bench.NUM_ANSWERS).to(dev).train()
opt = build_optimizer(model, name="flat_adamw")
batch = bench.synthetic_batch(bench.BATCH, bench.N_POINTS, 7, dev)
for _ in range(2):
    train_step(model, opt, dict(batch))
torch.cuda.synchronize()

counts = collections.Counter()
active = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if ("situation3d_amd" in fr.filename or fr.filename.endswith("bench.py")) and "glue_sites" not in fr.filename:
            return "%s:%d %s" % (os.path.relpath(fr.filename, ROOT), fr.lineno, fr.line.strip()[:70])
    return "?"


def wrap(owner, name, label=None, only_if=None):
    orig = getattr(owner, name)

    def f(*a, **k):
        if active[0] and (only_if is None or only_if(*a, **k)):
            counts[(label or name, site())] += 1
        return orig(*a, **k)
    setattr(owner, name, f)


T = torch.Tensor
wrap(T, "copy_")
wrap(T, "clone")
wrap(T, "contiguous", only_if=lambda t, *a, **k: not t.is_contiguous())
wrap(T, "zero_")
wrap(T, "fill_")
wrap(T, "to", only_if=lambda t, *a, **k: True)
wrap(T, "float")
wrap(T, "expand")
for fn in ("cat", "zeros", "ones", "zeros_like", "ones_like", "empty_like", "split", "stack", "full"):
    wrap(torch, fn)
wrap(T, "new_zeros")
wrap(T, "mean")
wrap(T, "sum")
active[0] = True
train_step(model, opt, dict(batch))
torch.cuda.synchronize()
active[0] = False
for (op, where), n in sorted(counts.items(), key=lambda kv: (-kv[1], kv[0])):
    if op in ("expand", "split", "empty_like"):
        continue
    print("%3d x %-12s %s" % (n, op, where))

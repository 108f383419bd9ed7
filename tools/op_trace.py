"""The ORDERED list of launches of one eager training step: every library entry point (`_lib.call`) and every ATen op
that is not a view, with the shapes / strides of its tensor arguments -- to find which line of the package issues a
given torch copy / fill / reduction seen in a kernel trace (tools/step_timeline.py --list).

python tools/op_trace.py [--all]        (--all: library calls too; default: ATen ops with one library call of context)
"""
import os, sys
import torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from situation3d_amd import _lib, gemm_tuning
from situation3d_amd.model import SIG3DQFormer
from situation3d_amd.trainer import build_optimizer, train_step

VIEWS = {"view", "_unsafe_view", "reshape", "t", "transpose", "expand", "slice", "select", "as_strided", "detach", "alias",
         "unsqueeze", "squeeze", "permute", "split", "split_with_sizes", "unbind", "_reshape_alias", "empty", "empty_like",
         "empty_strided", "new_empty", "new_empty_strided", "is_pinned", "_local_scalar_dense", "lift_fresh", "view_as",
         "numpy_T", "chunk", "narrow", "unfold", "_to_copy_noop", "is_same_size", "sym_size", "sym_stride", "stride", "size"}
log = []
on = [False]


def desc(a):
    if isinstance(a, torch.Tensor):
        return "%s%s%s" % (tuple(a.shape), "" if a.is_contiguous() else "s%s" % (tuple(a.stride()),), str(a.dtype)[6:])
    if isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor):
        return "[" + ",".join(desc(x) for x in a) + "]"
    return None


class Trace(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if on[0] and name not in VIEWS:
            cuda = any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + [out])
            if cuda or name in ("zeros", "ones", "full", "zeros_like"):
                log.append(("aten", name + " " + " ".join(d for d in (desc(a) for a in args) if d)))
        return out


orig_call = _lib.call


def call(name, *a):
    if on[0]:
        log.append(("lib", name))
    return orig_call(name, *a)


_lib.call = call
dev = torch.device("cuda", 0)
gemm_tuning.enable(tune_missing=False)
torch.manual_seed(0)
model = SIG3DQFormer(num_answers=bench.NUM_ANSWERS).to(dev).train()
opt = build_optimizer(model, name="flat_adamw")
batch = bench.synthetic_batch(bench.BATCH, bench.N_POINTS, 7, dev)
for _ in range(2):
    train_step(model, opt, dict(batch))
torch.cuda.synchronize()
with Trace():
    on[0] = True
    train_step(model, opt, dict(batch))
    torch.cuda.synchronize()
    on[0] = False
show_all = "--all" in sys.argv
prev_lib = None
n_aten = 0
for i, (kind, what) in enumerate(log):
    if kind == "lib":
        if show_all:
            print("%4d      %s" % (i, what))
        prev_lib = what
    else:
        n_aten += 1
        nxt = next((w for k, w in log[i + 1:] if k == "lib"), "")
        print("%4d ATEN %-100s | after %s | before %s" % (i, what[:100], prev_lib, nxt))
print("%d aten ops, %d library calls" % (n_aten, len(log) - n_aten))

"""Every ATen op of one eager training step that launches a kernel, attributed to the innermost frame of this
package (forward code and the Python bodies of custom autograd Functions alike) -- the list VERDICT r01 item 7
asked for.  View / metadata ops are skipped.

python tools/dispatch_trace.py [--surface]
"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from situation3d_amd import gemm_tuning  # noqa: E402
from situation3d_amd.model import SIG3DQFormer  # noqa: E402
from situation3d_amd.trainer import build_optimizer, train_step  # noqa: E402

VIEWS = {"view", "_unsafe_view", "reshape", "t", "transpose", "permute", "expand", "slice", "select", "unsqueeze",
         "squeeze", "as_strided", "detach", "alias", "narrow", "unbind", "split", "split_with_sizes", "chunk",
         "empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "size", "stride", "sym_size",
         "is_same_size", "_local_scalar_dense", "unflatten", "flatten", "view_as", "expand_as", "lift_fresh",
         "set_", "resize_", "_reshape_alias", "record_stream", "is_pinned", "_to_copy.meta", "item", "dim",
         "numel", "sym_numel", "sym_stride", "sym_storage_offset", "prim_device", "device", "_version"}


def site():
    for fr in reversed(traceback.extract_stack()[:-3]):
        if ("situation3d_amd" in fr.filename or fr.filename.endswith("bench.py")) and "tools" not in fr.filename:
            return "%s:%d %s" % (os.path.relpath(fr.filename, ROOT), fr.lineno, (fr.line or "").strip()[:64])
    return "(autograd engine / torch internals)"


class Trace(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.counts = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func._schema.name.split("::")[-1]
        on_gpu = any(torch.is_tensor(a) and a.is_cuda for a in list(args) + list((kwargs or {}).values())) or (
            torch.is_tensor(out) and out.is_cuda)
        if name not in VIEWS and on_gpu:
            shape = tuple(out.shape) if torch.is_tensor(out) else ""
            self.counts[(name, site(), str(shape))] += 1
        return out


if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    gemm_tuning.enable(tune_missing=False)
    torch.manual_seed(0)
    model = SIG3DQFormer(num_answers=bench.NUM_ANSWERS).to(dev).train()
    opt = build_optimizer(model, name="flat_adamw")
    batch = bench.synthetic_batch(bench.BATCH, bench.N_POINTS, 7, dev, surface="--surface" in sys.argv)
    for _ in range(2):
        train_step(model, opt, dict(batch))
    torch.cuda.synchronize()
    with Trace() as tr:
        train_step(model, opt, dict(batch))
    torch.cuda.synchronize()
    by_op = collections.Counter()
    for (name, where, shape), n in tr.counts.items():
        by_op[name] += n
    print("ATen ops with GPU tensors in one step: %d" % sum(by_op.values()))
    print("  " + ", ".join("%s x%d" % kv for kv in by_op.most_common()))
    by_site = collections.defaultdict(list)
    for (name, where, shape), n in tr.counts.items():
        by_site[where].append((n, name, shape))
    for where, ops in sorted(by_site.items(), key=lambda kv: -sum(o[0] for o in kv[1])):
        print("%4d  %s" % (sum(o[0] for o in ops), where))
        for n, name, shape in sorted(ops, reverse=True)[:6]:
            print("        %3d x %-22s %s" % (n, name, shape))

"""hipGraph capture of the training step.

One step of the composed hot path is ~2000 kernel launches (12 Q-Former layers x ~40 small
kernels forward, twice that backward, 4 SA levels, the optimizer).  Issued eagerly from Python
that is ~20 us of host work per launch -- the step becomes host-bound at ~45 ms while the GPU
needs far less.  The whole step (forward, losses, backward, value clip, AdamW) is therefore
captured ONCE into a hipGraph and replayed: one host call per step, launch gaps of ~1-2 us set
by the hardware queue instead of the interpreter.  Everything on the path is capture-safe by
construction: the C-ABI kernels launch on the capturing stream, their zero-fills are memset
nodes, no entry point synchronises or allocates, and the optimizer runs in `capturable` mode.

Inputs live in static device buffers that are refreshed (device-to-device copy) before each
replay; outputs (loss) are read from static buffers after it.
"""
import torch
import torch.nn as nn

from .trainer import get_loss


def _copy_into(dst, src):
    for k, v in src.items():
        if isinstance(v, dict):
            _copy_into(dst[k], v)
        else:
            dst[k].copy_(v, non_blocking=True)


def _clone(d):
    return {k: (_clone(v) if isinstance(v, dict) else v.clone()) for k, v in d.items()}


class GraphedTrainStep:
    """Captures `zero_grad -> forward -> get_loss -> backward -> clip_grad_value_ -> step`
    (lib/solver.py:374-402, 618-627) for one fixed batch shape."""

    def __init__(self, model, optimizer, example_batch, max_grad_value=1.0, warmup=3):
        self.model, self.optimizer = model, optimizer
        self.max_grad_value = max_grad_value
        self.static_batch = _clone(example_batch)
        self.static_loss = None
        self.graph = torch.cuda.CUDAGraph()
        params = [p for p in model.parameters() if p.requires_grad]

        def one_step():
            out = model(dict(self.static_batch))
            loss, _ = get_loss(out)
            loss.backward()
            if max_grad_value is not None and max_grad_value > 0:
                nn.utils.clip_grad_value_(params, clip_value=max_grad_value)
            optimizer.step()
            return loss

        # warm-up on a side stream (library workspaces, autotuning, allocator pools), as required
        # before capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                optimizer.zero_grad(set_to_none=True)
                one_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.static_loss = one_step()
        torch.cuda.synchronize()

    def __call__(self, batch):
        _copy_into(self.static_batch, batch)
        self.graph.replay()
        return self.static_loss

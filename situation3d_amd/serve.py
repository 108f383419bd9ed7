"""Forward-only serving step as one hipGraph with the geometry chains of the NEXT batches beside it
(BASELINE config 2: SQA3D forward, 40k points, B = 4).

Alone, a forward pass is bound by the furthest-point-sampling chain (~3800 dependent rounds, 4.8 ms of
the 8.0 ms at B = 4) while the rest of the chip idles.  FPS / centre gather / ball query depend on the
coordinates only (geometry.GeometryPlan), so the chains of the next `depth` batches run as graphs of their own
on streams of their own (geometry.GeometryPipeline, the machinery of the training step:
graph_step.GraphedTrainStep) while the forward of batch i -- ONE linear graph -- reads `plan_cur`.  Outputs are
identical to the inline path.
"""
import os

import torch

from . import gemm_tuning
from .geometry import GeometryPipeline
from .graph_step import _clone, _copy_into


class PipelinedForward:
    """Forward-only serving with the geometry chains of the next `depth` batches in flight at once.

    One chain ahead, the step is bound by the furthest-point-sampling chain of ONE batch (B = 4: 5.3 ms per batch
    with a 2.3 ms forward): the chain is 2048 strictly dependent rounds, but it only occupies 8 workgroups per scene,
    so the chains of SEVERAL batches run side by side at nearly full speed each.  Every in-flight batch has its own
    geometry graph, stream, plan and coordinate buffer (geometry.GeometryPipeline); call i
      1. waits for the chain of batch i (launched `depth` calls ago), hands its plan over to the forward's plan
         (one table copy),
      2. launches the chain of batch i + depth on the stream that has just become free,
      3. replays the forward graph of batch i.
    Outputs are those of the inline forward, bit for bit; a caller that breaks the announced order pays one inline
    chain for that batch (geometry.Announced: tokens, or tensor object + version).
    Measured (tools/infer_bench.py, 40 000 points): B = 4: 5.1 ms per batch one chain ahead -> 2.92 ms with two chains in
    flight (1369 samples/s), 2.88 with three; B = 8: 5.8 -> 3.75 ms with two or three (2135 samples/s).  (Round 2,
    with streams as HIP handed them out and chains behind stream waits, was erratic in the depth -- B = 8 / two chains
    5.65 ms, B = 4 / three 4.5 ms: chains that shared a hardware queue; GeometryPipeline now picks streams that do not.)
    High-priority geometry streams stay pathological (12-13 ms per batch with two chains): leave `high_priority` off."""

    def __init__(self, model, example_batch, depth=2, warmup=2, geometry_levels=None, high_priority=False):
        stream = torch.cuda.current_stream()
        if stream == torch.cuda.default_stream():
            raise RuntimeError("%s must be built (and used) inside `with torch.cuda.stream(s):`" % type(self).__name__)
        assert depth >= 1
        self.model, self.depth, self.stream = model.eval(), int(depth), stream
        self.static_batch = _clone(example_batch)
        pc = self.static_batch["point_clouds"]
        b, n = pc.shape[0], pc.shape[1]
        self._pipe = GeometryPipeline(b, n, geometry_levels or model.encoder.LEVELS, pc.device, stream, depth=self.depth,
                                      handshake=os.environ.get("SIG3D_GEO_HANDSHAKE", "1") != "0",
                                      stream_priority=-1 if high_priority else 0,
                                      fps_waves=16)    # forward-only: the sampling's latency bounds a chain, not its footprint
        self.plan_cur = self._pipe.plan_cur
        self.plan_cur.compute(pc[..., :3].contiguous())

        def fwd():
            batch = dict(self.static_batch)
            batch["geometry_plan"] = self.plan_cur
            with torch.no_grad():
                return model(batch)

        for _ in range(warmup):
            fwd()
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with gemm_tuning.no_tuning(), torch.cuda.graph(self.graph, stream=stream):
            self.static_out = fwd()
        self._pipe.capture(pool=self.graph.pool())
        torch.cuda.synchronize()

    @property
    def inline_chains(self):
        return self._pipe.inline_chains

    def handshake_timed_out(self):
        """True when a geometry chain ever gave up waiting for its start ticket (synchronises)."""
        return self._pipe.timed_out()

    def __call__(self, batch, upcoming, token=None, upcoming_tokens=None):
        """Outputs of `batch`.  `upcoming`: the batches of the next `depth` calls, in order (upcoming[-1] is the one
        whose geometry chain starts now; the others were announced by earlier calls)."""
        self._pipe.advance(batch["point_clouds"], [u["point_clouds"] for u in upcoming], token, upcoming_tokens)
        _copy_into(self.static_batch, batch)
        self.graph.replay()
        return self.static_out


class GraphedForward(PipelinedForward):
    """One chain ahead: the geometry of batch i+1 under the forward of batch i (rounds 1-2 forked that branch INSIDE
    the forward's graph; a graph of its own beside a linear forward graph is what the training step measured as
    faster, see graph_step.GraphedTrainStep)."""

    def __init__(self, model, example_batch, warmup=2, geometry_levels=None):
        super().__init__(model, example_batch, depth=1, warmup=warmup, geometry_levels=geometry_levels)

    def __call__(self, batch, next_batch, token=None, next_token=None):
        """Outputs of `batch`; `next_batch` announces the batch of the following call (its geometry is computed under
        this call's forward).  Breaking the announced order costs one inline geometry chain."""
        return super().__call__(batch, [next_batch], token, [next_token])

"""Forward-only serving step as one hipGraph with the geometry of the NEXT batch on a forked branch
(BASELINE config 2: SQA3D forward, 40k points, B = 4).

Alone, a forward pass is bound by the furthest-point-sampling chain (~3800 dependent rounds, 4.8 ms of
the 8.0 ms at B = 4) while the rest of the chip idles.  FPS / centre gather / ball query depend on the
coordinates only (geometry.GeometryPlan), so the graph has two branches, exactly like the training step
(graph_step.GraphedTrainStep): the forward of batch i reading `plan_cur`, and the geometry chain of batch
i+1 filling `plan_next`; the plans are handed over at the join.  Outputs are identical to the inline path.
"""
import torch

from . import gemm_tuning
from .geometry import GeometryPlan
from .graph_step import Announced, _clone, _copy_into


class GraphedForward:
    def __init__(self, model, example_batch, warmup=2, geometry_levels=None):
        stream = torch.cuda.current_stream()
        if stream == torch.cuda.default_stream():
            raise RuntimeError("GraphedForward must be built (and used) inside `with torch.cuda.stream(s):`")
        self.model = model.eval()
        self.static_batch = _clone(example_batch)
        pc = self.static_batch["point_clouds"]
        b, n = pc.shape[0], pc.shape[1]
        levels = geometry_levels or model.encoder.LEVELS
        self.plan_cur = GeometryPlan(b, n, levels, pc.device)
        self.plan_next = GeometryPlan(b, n, levels, pc.device)
        self.static_next_xyz = pc[..., :3].contiguous()
        self.side = torch.cuda.Stream(pc.device)
        self.plan_cur.copy_from(self.plan_next)          # builds the hand-over's copy table OUTSIDE any capture
        self.plan_cur.compute(self.static_next_xyz)
        self._announced = Announced()

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
            self.side.wait_stream(stream)                    # fork
            with torch.cuda.stream(self.side):
                self.plan_next.compute(self.static_next_xyz)
            self.static_out = fwd()
            stream.wait_stream(self.side)                    # join
            self.plan_cur.copy_from(self.plan_next)          # hand over for the next replay
        torch.cuda.synchronize()

    def __call__(self, batch, next_batch, token=None, next_token=None):
        """Outputs of `batch`; `next_batch` announces the batch of the following call (its geometry is
        computed under this call's forward).  Breaking the announced order costs one inline geometry chain.
        The hand-over is keyed on explicit tokens or on tensor identity + version (graph_step.Announced),
        never on a device address."""
        if not self._announced.matches(batch["point_clouds"], token):
            self.plan_cur.compute(batch["point_clouds"][..., :3].contiguous())
        self.static_next_xyz.copy_(next_batch["point_clouds"][..., :3], non_blocking=True)
        self._announced.set(next_batch["point_clouds"], next_token)
        _copy_into(self.static_batch, batch)
        self.graph.replay()
        return self.static_out


class PipelinedForward:
    """Forward-only serving with the geometry chains of the next `depth` batches in flight at once.

    GraphedForward is bound by the furthest-point-sampling chain of ONE batch (B = 4: 5.3 ms per batch with a 2.3 ms
    forward): the chain is 2048 strictly dependent rounds, but it only occupies 8 workgroups per scene, so the
    chains of SEVERAL batches run side by side at nearly full speed each.  Here every in-flight batch has its own
    geometry graph, stream, plan and coordinate buffer; call i
      1. waits for the chain of batch i (launched `depth` calls ago), hands its plan over to the forward's plan
         (one table copy),
      2. launches the chain of batch i + depth on the stream that has just become free,
      3. replays the forward graph of batch i.
    Outputs are those of the inline forward, bit for bit; a caller that breaks the announced order pays one inline
    chain for that batch (same identity rule as GraphedForward: tokens, or tensor object + version).
    Measured (tools/infer_bench.py, 40 000 points): B = 4: 5.2 ms per batch one chain ahead -> 2.97 ms with two
    chains in flight (1345 samples/s); B = 8: 5.9 -> 4.0 ms with three (2000 samples/s).  The best depth depends on
    how HIP maps the streams onto its four hardware queues per priority (B = 4 / three chains: 4.5 ms; B = 8 / two:
    5.65 ms), and high-priority geometry streams are pathological with two chains (13 ms) -- measure before
    changing `depth` / `high_priority`."""

    def __init__(self, model, example_batch, depth=2, warmup=2, geometry_levels=None, high_priority=False):
        stream = torch.cuda.current_stream()
        if stream == torch.cuda.default_stream():
            raise RuntimeError("PipelinedForward must be built (and used) inside `with torch.cuda.stream(s):`")
        assert depth >= 1
        self.model, self.depth, self.stream = model.eval(), int(depth), stream
        self.static_batch = _clone(example_batch)
        pc = self.static_batch["point_clouds"]
        b, n = pc.shape[0], pc.shape[1]
        levels = geometry_levels or model.encoder.LEVELS
        self.plan_cur = GeometryPlan(b, n, levels, pc.device)
        self.slots = []
        for _ in range(self.depth):
            slot = dict(plan=GeometryPlan(b, n, levels, pc.device), xyz=pc[..., :3].contiguous(),
                        stream=torch.cuda.Stream(pc.device, priority=-1 if high_priority else 0),
                        graph=torch.cuda.CUDAGraph(), announced=Announced())
            self.slots.append(slot)
        self.plan_cur.copy_from(self.slots[0]["plan"])      # builds the copy table outside any capture
        self.plan_cur.compute(self.slots[0]["xyz"])
        for slot in self.slots:
            slot["plan"].compute(slot["xyz"])                # scratch allocations of the chain, outside capture
        self._tables = [None] * self.depth
        self.calls = 0

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
        for slot in self.slots:
            slot["stream"].wait_stream(stream)
            with torch.cuda.graph(slot["graph"], stream=slot["stream"], pool=self.graph.pool()):
                slot["plan"].compute(slot["xyz"])
            stream.wait_stream(slot["stream"])
        torch.cuda.synchronize()

    def __call__(self, batch, upcoming, token=None, upcoming_tokens=None):
        """Outputs of `batch`.  `upcoming`: the batches of the next `depth` calls, in order (upcoming[-1] is the one
        whose geometry chain starts now; the others were announced by earlier calls)."""
        assert len(upcoming) == self.depth
        toks = list(upcoming_tokens) if upcoming_tokens is not None else [None] * self.depth
        slot = self.slots[self.calls % self.depth]
        if slot["announced"].matches(batch["point_clouds"], token):
            self.stream.wait_stream(slot["stream"])          # the chain launched `depth` calls ago
            self.plan_cur.copy_from(slot["plan"])
        else:                                                # prologue, or the caller broke the announced order
            self.plan_cur.compute(batch["point_clouds"][..., :3].contiguous())
        far = upcoming[-1]
        slot["xyz"].copy_(far["point_clouds"][..., :3], non_blocking=True)
        slot["announced"].set(far["point_clouds"], toks[-1])
        slot["stream"].wait_stream(self.stream)              # coordinates staged, plan handed over
        with torch.cuda.stream(slot["stream"]):
            slot["graph"].replay()
        _copy_into(self.static_batch, batch)
        self.graph.replay()
        self.calls += 1
        return self.static_out

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

"""hipGraph capture of the training step (+ the geometry chain of the next batch beside it).

One step of the composed hot path is ~2000 kernel launches (12 Q-Former layers x ~40 small
kernels forward, twice that backward, 4 SA levels, the optimizer).  Issued eagerly from Python
that is ~20 us of host work per launch -- the step becomes host-bound at ~45 ms while the GPU
needs far less.  The whole step (forward, losses, backward, value clip, AdamW) is therefore
captured ONCE into a hipGraph and replayed: one host call per step, launch gaps of ~1-2 us set
by the hardware queue instead of the interpreter.  Everything on the path is capture-safe by
construction: the C-ABI kernels launch on the capturing stream, their zero-fills are memset
nodes, no entry point synchronises or allocates, and the optimizer runs in `capturable` mode.

Geometry prefetch.  FPS / centre gather / ball query depend on xyz only (geometry.py) and FPS is
a latency-bound chain that keeps a handful of CUs busy for milliseconds.  With
`prefetch_geometry=True` the geometry plan of batch i+1 is computed while the training step of
batch i (reading the plan computed during the previous call) runs: the chain is a hipGraph of its
own on a stream of its own (geometry.GeometryPipeline) beside the step's graph, which stays ONE
linear chain, and the fresh plan is handed over at the start of the next call.  (Rounds 1-2 forked
the chain inside the step's graph and joined it at the end: slower; removed
in round 6.)  Every step still performs the full work of one batch (one geometry chain +
one training pass); only the order is pipelined, like a data loader running one batch ahead.  The
plan is bit-identical to computing it inline.

Stream discipline: warm-up, capture, replay and any eager steps of the same model must all run
on ONE non-default stream (`with torch.cuda.stream(work): ...`).  Autograd's AccumulateGrad
nodes remember the stream they were first used on; if that differs from the capture stream,
backward forks onto the other stream inside the capture and the allocator's single-stream
assumption for graph pools breaks (observed on MI355X as corrupted activations / faults).
The prefetch branch is safe because it allocates nothing: it only launches C-ABI kernels into
preallocated buffers.

Inputs live in static device buffers that are refreshed (device-to-device copy) before each
replay; outputs (loss, answer_scores, ...) are read from static buffers after it.
"""
import contextlib
import os

import torch
import torch.nn as nn

from . import gemm_tuning, timeline
from .geometry import Announced, GeometryPipeline, GeometryPlan  # noqa: F401  (Announced: re-exported)
from .scratch import STEP_ZEROS
from .trainer import get_loss


def _pairs(dst, src, out):
    for k, v in src.items():
        if isinstance(v, dict):
            _pairs(dst[k], v, out)
        else:
            out.append((dst[k], v))
    return out


_table_copies = {}


def _copy_into(dst, src, extra=()):
    """Refresh the static input buffers of a captured step from the caller's batch (+ `extra` (dst, src)
    pairs): one table-driven launch when every tensor is a contiguous device tensor of matching size
    (table_copy.TableCopy), plain copies otherwise."""
    pairs = _pairs(dst, src, []) + list(extra)
    ok = all(d.is_cuda and s.is_cuda and d.is_contiguous() and s.is_contiguous() and d.dtype == s.dtype
             and d.shape == s.shape and (d.numel() * d.element_size()) % 4 == 0 for d, s in pairs)
    if ok and pairs:
        dev = pairs[0][0].device
        tc = _table_copies.get(dev)
        if tc is None:
            from .table_copy import TableCopy
            tc = _table_copies[dev] = TableCopy(dev)
        tc(pairs)
        return
    for d, s in pairs:
        d.copy_(s, non_blocking=True)


def _clone(d):
    return {k: (_clone(v) if isinstance(v, dict) else v.clone()) for k, v in d.items()}


def _flush_deferred(model):
    """Deferred, layer-batched weight gradients of the Q-Former (qformer._WeightGradArena): fill whatever the
    backward pass so far has made ready (a no-op when nothing is pending or the mode is off)."""
    qf = getattr(model, "Qformer", None)
    enc = getattr(getattr(qf, "bert", None), "encoder", None)
    if enc is not None and hasattr(enc, "flush_weight_grads"):
        enc.flush_weight_grads()


def _bn_momenta(model):
    return [m.momentum for m in model.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)]


class GraphedTrainStep:
    """Captures `zero_grad -> forward -> get_loss -> backward -> clip_grad_value_ -> step`
    (lib/solver.py:374-402, 618-627) for one fixed batch shape, on the CURRENT stream."""

    def __init__(self, model, optimizer, example_batch, max_grad_value=1.0, warmup=3,
                 prefetch_geometry=False, geometry_levels=None, reducer=None, split_backward=True,
                 prefetch_depth=None, qf_cut=None):
        """`reducer` (ddp.GradBucketReducer, data parallel): the step becomes graph A (forward +
        backward, gradients accumulated into the reducer's flat buckets) -> eager bucketed RCCL
        all-reduce -> graph B (value clip + AdamW).  No collective is ever captured."""
        stream = torch.cuda.current_stream()
        if stream == torch.cuda.default_stream():
            raise RuntimeError("GraphedTrainStep must be built (and used) inside "
                               "`with torch.cuda.stream(work_stream):` -- not on the default stream")
        self.model, self.optimizer, self.stream = model, optimizer, stream
        self.static_batch = _clone(example_batch)
        self.static_loss = None
        self.static_out = None
        self.graph = torch.cuda.CUDAGraph()
        self.reducer = reducer
        self.graph_opt = torch.cuda.CUDAGraph() if reducer is not None else None
        if reducer is not None:
            reducer.hooks_enabled = False
            if getattr(optimizer, "process_group", None) is None:
                optimizer.process_group = reducer.group   # liveness agreed on the reducer's group from the first gather on
        self.prefetch = bool(prefetch_geometry)
        # The geometry chains of the next `prefetch_depth` batches (default 1; bench.py runs 3) are graphs of their own
        # on streams of their own (geometry.GeometryPipeline) beside the step's graph, which stays ONE linear chain:
        # every kernel node enqueued while another queue holds a blocked barrier costs ~1.7 us extra on this runtime
        # (tools/probes/fork_penalty.py), which is what a branch forked inside the step's graph paid.
        self.prefetch_depth = 1 if not self.prefetch else max(1, int(prefetch_depth or 1))
        self._pipe = None
        params =[p for p in model.parameters() if p.requires_grad]

        if self.prefetch:
            pc = self.static_batch["point_clouds"]
            b, n = pc.shape[0], pc.shape[1]
            levels = geometry_levels or model.encoder.LEVELS
            self._pipe = GeometryPipeline(b, n, levels, pc.device, stream, depth=self.prefetch_depth,
                                          handshake=os.environ.get("SIG3D_GEO_HANDSHAKE", "1") != "0",
                                          example_xyz=pc[..., :3])
            self.plan_cur = self._pipe.plan_cur
            if reducer is not None:
                from . import ddp
                ddp._OWN_BESIDE.extend(sl["stream"] for sl in self._pipe.slots)   # own-stream collectives: other queues
            self.plan_cur.compute(pc[..., :3].contiguous())  # geometry of the example batch

        def fwd_bwd():
            batch = dict(self.static_batch)
            if self.prefetch:
                batch["geometry_plan"] = self.plan_cur
            # every zero-initialised accumulator of the step: slices of one region, one fill node (scratch.py)
            STEP_ZEROS.begin_step(batch["point_clouds"].device, grads_ok=fused_opt)
            try:
                out = model(batch)
                loss, out = get_loss(out)
                self.static_out = out  # answer_scores, aux_scores, ... of the last replay
                loss.backward()
                _flush_deferred(model)
                if reducer is not None and fused_opt:
                    optimizer.gather_grads()  # scattered grads -> the flat buffers the reducer owns
            finally:
                STEP_ZEROS.end_step()
            return loss

        def fwd_bwd_head():
            """Split mode, part 1: forward, then backward of the heads and the UPPER Q-Former layers (down to
            the cut inside the Q-Former and to the visual tokens), gradients gathered into the flat buffers."""
            batch = dict(self.static_batch)
            batch["_split_backward"] = True
            batch["_qf_cut"] = self._qf_cut
            if self.prefetch:
                batch["geometry_plan"] = self.plan_cur
            STEP_ZEROS.begin_step(batch["point_clouds"].device, grads_ok=True)   # closed by bwd_encoder()
            try:
                out = model(batch)
                self._boundary = out.pop("_boundary")
                self._qf_boundary = out.pop("_qf_boundary", None)
                out.pop("_split_backward", None)
                out.pop("_qf_cut", None)
                loss, out = get_loss(out)
                self.static_out = out
                leaves = [self._boundary[1]] + ([self._qf_boundary[1]] if self._qf_boundary is not None else [])
                loss.backward(inputs=self._upper_params + leaves)
                _flush_deferred(model)   # weight gradients of the layers above the cut (qformer._WeightGradArena)
                optimizer.gather_grads(zero=True)
            except BaseException:
                STEP_ZEROS.end_step()    # a step that died must not leave the zero region open for whoever runs next
                raise
            return loss

        def bwd_lower():
            """Split mode, part 2: the lower Q-Former layers and its embeddings from the cut's gradient; the
            gradient of the visual tokens keeps accumulating in their boundary leaf."""
            hidden, leaf = self._qf_boundary
            hidden.backward(leaf.grad, inputs=self._lower_params + [self._boundary[1]])
            _flush_deferred(model)
            optimizer.gather_grads(zero=False)
            self._qf_boundary = None

        def bwd_encoder():
            """Split mode, last part: position MLP and point encoder from the tokens' gradient, own gather."""
            try:
                tokens, leaf = self._boundary
                tokens.backward(leaf.grad)
                optimizer.gather_grads(zero=False)
                self._boundary = None
            finally:
                STEP_ZEROS.end_step()

        fused_opt = getattr(optimizer, "flat_grad_buffers", None) is not None  # optim.FlatAdamW
        # data parallel + flat storage: AdamW runs bucket by bucket behind that bucket's all-reduce
        self._bucketed_update = bool(reducer is not None and fused_opt and getattr(reducer, "flat_mode", False))
        # ... and the backward pass is cut at the point encoder's output: the all-reduce of everything
        # downstream (Q-Former, heads: 99 % of the gradient bytes) runs while the encoder's backward
        # (a third of the step) is still computing.  Two graphs + two bucket sets; no collective captured.
        self._split = False
        self._qf_cut = None
        self._emb_sink = None
        if (self._bucketed_update and split_backward and os.environ.get("SIG3D_NO_SPLIT") is None
                and hasattr(model, "encoder") and hasattr(model, "Qformer")):
            from .ddp import GradBucketReducer
            layers = list(model.Qformer.bert.encoder.layer)
            # qf_cut = k: also cut the backward after Q-Former layer k (third graph / gather / bucket set), so that the
            # upper layers' gradients travel under the lower layers' backward.  The deferred weight gradients are
            # written straight into the flat gradient buffers, so the cut is only cheap when the optimizer stores the
            # layers below and above it as two arenas (trainer.build_optimizer(qf_cut=k) -> encoder.storage_cut, the
            # default here): over ONE kind-major arena the two pieces interleave kind by kind and fall into dozens of
            # small collectives and AdamW launches (+0.95 ms per step at world size 1).
            enc_mod = model.Qformer.bert.encoder
            cut = qf_cut if qf_cut is not None else (getattr(enc_mod, "storage_cut", None) or 0)
            if not getattr(model.Qformer.bert, "segmented_layout", False) or len(layers) < 2:
                cut = 0
            cut = max(0, min(int(cut), len(layers) - 1))
            up_mods = layers[cut:] + [model.position_head, model.rotation_head, model.aux_reg, model.answer_cls]
            upper = [p for mod in up_mods for p in mod.parameters() if p.requires_grad]
            up_ids = {id(p) for p in upper}
            lower = [p for p in model.Qformer.parameters() if p.requires_grad and id(p) not in up_ids]
            lower += [model.query_tokens] if model.query_tokens.requires_grad else []
            if cut == 0:           # no cut inside the Q-Former: everything downstream of the tokens is one part
                upper, lower = upper + lower, []
            # the word-embedding table's gradient (94 MB dense, <= B x T live rows) travels as rows
            # (ddp.SparseRowExchange) unless SIG3D_DENSE_EMBED_GRAD is set
            emb = getattr(model.Qformer.bert, "embeddings", None)
            q = self.static_batch.get("q_feat")
            table = emb.word_embeddings.weight if emb is not None and q is not None else None
            if table is not None and (not table.requires_grad or os.environ.get("SIG3D_DENSE_EMBED_GRAD")):
                table = None
            parts = optimizer.flat_grad_parts([upper, lower], exclude=[table] if table is not None else ())
            if parts[0] and parts[2]:
                if table is not None:
                    from .ddp import SparseRowExchange
                    self._emb_sink = SparseRowExchange(q["input_ids"].numel(), table.shape[1], table.device,
                                                       process_group=reducer.group,
                                                       padding_idx=emb.word_embeddings.padding_idx)
                    emb.row_grad_sink = self._emb_sink
                    gi, lo, hi, self._emb_grad = optimizer.mark_externally_reduced(table)
                    self._emb_range = (gi, lo, hi)
                self._qf_cut = cut or None
                self._upper_params, self._lower_params = upper, lower
                self._red_head = GradBucketReducer.from_flat(parts[0], process_group=reducer.group)
                self._red_low = GradBucketReducer.from_flat(parts[1], process_group=reducer.group) if lower else None
                self._red_enc = GradBucketReducer.from_flat(parts[2], process_group=reducer.group)
                self.graph_low = torch.cuda.CUDAGraph() if lower else None
                self.graph_enc = torch.cuda.CUDAGraph()
                self._split = True

        if reducer is not None and fused_opt and hasattr(model, "Qformer"):
            # deferred weight gradients straight into the flat gradient buffers the reducer owns
            model.Qformer.bert.encoder.grad_store = optimizer.flat_grad_run

        def update():
            if not fused_opt and max_grad_value is not None and max_grad_value > 0:
                nn.utils.clip_grad_value_(params, clip_value=max_grad_value)
            optimizer.step()  # FlatAdamW: clip + AdamW + zero_grad in one kernel per group

        def clear_grads():
            if fused_opt:
                return  # flat gradients are zeroed by every step() and start at zero
            if reducer is not None:
                reducer.zero_grad()   # gradients live in the flat buckets: zero those in place
            else:
                optimizer.zero_grad(set_to_none=True)

        for _ in range(warmup):  # library workspaces, autotuning, allocator pools
            clear_grads()
            fwd_bwd()
            if reducer is not None:
                reducer.reduce_all()
            if self._emb_sink is not None:
                self._emb_sink.launch()
                self._emb_sink.finish_into(self._emb_grad)
            update()
        torch.cuda.synchronize()
        if reducer is None:
            optimizer.zero_grad(set_to_none=True)
        # with a live process group its watchdog thread issues HIP calls of its own (event queries);
        # they are harmless to this capture, so only police the capturing thread
        import torch.distributed as dist
        cap_mode = "thread_local" if dist.is_available() and dist.is_initialized() else "global"
        # the gradient-table uploads of THIS step's captures read pinned staging buffers of their own, dropped with it
        self._opt_tables = optimizer.new_capture_tables(4) if fused_opt else None
        scope = optimizer.captured_with(self._opt_tables) if fused_opt else contextlib.nullcontext()
        with scope:
            self._capture_all(stream, cap_mode, reducer, fwd_bwd, fwd_bwd_head, bwd_lower, bwd_encoder, update)
        torch.cuda.synchronize()
        # hyper-parameters that are baked into captured launches by VALUE: BatchNorm momentum
        # (sig3d_bn_finalize; the reference's BNMomentumScheduler changes it per epoch, lib/solver.py:248-257)
        # and, for optimizers other than FlatAdamW with fused kernels, nothing else.  The learning rate is a
        # device scalar (optim.FlatAdamW.sync_lr) and follows a scheduler across replays.
        self._captured_bn_momenta = _bn_momenta(model)

    def _capture_all(self, stream, cap_mode, reducer, fwd_bwd, fwd_bwd_head, bwd_lower, bwd_encoder, update):
        with gemm_tuning.no_tuning(), torch.cuda.graph(self.graph, stream=stream, capture_error_mode=cap_mode):
            if reducer is not None:
                reducer.zero_grad()
            timeline.mark("main:start")
            self.static_loss = fwd_bwd_head() if self._split else fwd_bwd()
            timeline.mark("main:backward done")
            if reducer is None:
                update()
            timeline.mark("main:update done")
            timeline.mark("main:end")
        if self._split:
            if self.graph_low is not None:
                with gemm_tuning.no_tuning(), torch.cuda.graph(self.graph_low, stream=stream, pool=self.graph.pool(),
                                                               capture_error_mode=cap_mode):
                    bwd_lower()
            with gemm_tuning.no_tuning(), torch.cuda.graph(self.graph_enc, stream=stream, pool=self.graph.pool(), capture_error_mode=cap_mode):
                bwd_encoder()
        if self._pipe is not None:
            # The geometry chains as graphs of their own, replayed on their streams under ALL the graphs and
            # collectives of a step.  (A branch forked inside the first graph has to finish with that graph:
            # +1.1 ms per step in the split form, and with RCCL's stream on a high-priority queue (ddp._pg_options)
            # a graph with an internal fork replays pathologically slowly: 26 ms per step with a group of one.)
            self._pipe.capture(pool=self.graph.pool(), capture_error_mode=cap_mode)
        if reducer is not None:
            with gemm_tuning.no_tuning(), torch.cuda.graph(self.graph_opt, stream=stream,
                                                           pool=self.graph.pool(),
                                                           capture_error_mode=cap_mode):
                update()

    def handshake_timed_out(self):
        """True when a geometry chain ever gave up waiting for its ticket (geometry.GeometryPipeline.timed_out)."""
        return self._pipe is not None and self._pipe.timed_out()

    def __call__(self, batch, next_batch=None, token=None, next_token=None, upcoming=None, upcoming_tokens=None):
        """`upcoming`: the batches of the next `prefetch_depth` calls, in order (the chain of upcoming[-1] starts
        under this step; `next_batch` alone is accepted at depth 1).  `token` / `next_token` / `upcoming_tokens`
        (optional): the caller's ids of those batches (step or sequence numbers).  Without them the hand-over is
        keyed on tensor identity + version (geometry.Announced); a caller that refills ONE buffer in place must pass
        tokens to keep the prefetch, or pays an inline geometry chain per step -- never a wrong plan."""
        if _bn_momenta(self.model) != self._captured_bn_momenta:
            raise RuntimeError("BatchNorm momentum changed since capture (BNMomentumScheduler): the captured "
                               "sig3d_bn_finalize launches hold the old value -- build a new GraphedTrainStep")
        sync_lr = getattr(self.optimizer, "sync_lr", None)
        if sync_lr is not None:
            sync_lr()   # a scheduler's new learning rate -> the device scalar the captured AdamW reads
        if self.prefetch:
            if upcoming is None:
                if next_batch is None:
                    raise ValueError("prefetch_geometry=True needs the batch(es) of the NEXT step(s)")
                upcoming, upcoming_tokens = [next_batch], [next_token]
            if len(upcoming) != self.prefetch_depth:
                raise ValueError("prefetch depth %d: pass upcoming=[the next %d batches]"
                                 % (self.prefetch_depth, self.prefetch_depth))
        if self._pipe is not None:
            # hand over the plan of `batch`, start the chain of upcoming[-1] -- under everything below
            self._pipe.advance(batch["point_clouds"], [u["point_clouds"] for u in upcoming], token, upcoming_tokens)
        _copy_into(self.static_batch, batch)
        self.graph.replay()
        if self.reducer is not None:
            if self._split:
                self._red_head.launch_all()      # heads + upper Q-Former layers: on the wire ...
                if self.graph_low is not None:
                    self.graph_low.replay()      # ... while the lower layers' backward runs,
                    self._red_low.launch_all()   # whose gradients then travel ...
                if self._emb_sink is not None:
                    self._emb_sink.launch()      # word-embedding rows (the embeddings' backward has just run)
                self.graph_enc.replay()          # ... under the point encoder's backward
                self._red_enc.launch_all()
                self.optimizer.mark_gathered()
                self.optimizer.begin_bucketed_step()
                self.optimizer.update_buckets(self._red_head)
                if self._red_low is not None:
                    self.optimizer.update_buckets(self._red_low)
                if self._emb_sink is not None:
                    self._emb_sink.finish_into(self._emb_grad)     # all ranks' rows -> the table's dense slot
                    self.optimizer.update_range(*self._emb_range)
                self.optimizer.update_buckets(self._red_enc)
                self.optimizer.end_bucketed_step()
            elif self._bucketed_update:
                # all-reduce per bucket, AdamW per bucket right behind it (eager launches, ~12 per step)
                self.optimizer.mark_gathered()   # the replayed graph filled the flat gradient buffers
                self.optimizer.step_after(self.reducer)
            else:
                self.reducer.reduce_all()
                self.graph_opt.replay()
        return self.static_loss

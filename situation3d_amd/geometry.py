"""Geometry plan of the point encoder: everything in the SA stack that depends on xyz ONLY.

FPS indices, centre coordinates and ball-query neighbour lists of all set-abstraction levels are
functions of the input coordinates alone (pointnet2_modules.py:233-240 + pointnet2_utils.py:334)
-- no learned parameter and no feature enters them.  FPS is a chain of ~3800 strictly dependent
rounds per step (latency-bound, a few workgroups), so it is the worst possible citizen of the
critical path and the best possible candidate for overlap: a `GeometryPlan` computes the whole
chain into PREALLOCATED buffers through the C ABI (no allocator traffic, safe on a side stream
and inside hipGraph capture), which lets the training step of batch i run concurrently with the
geometry of batch i+1 (graph_step.GraphedTrainStep, `prefetch_geometry=True`) -- the same role
the reference gives its DataLoader workers for the CPU-side voxelisation (lib/sepdataset.py).

The numbers produced are bit-identical to calling the ops inline.
"""
import ctypes
import os

import torch

from . import _lib, timeline
from .pointnet2 import _ext, fused_mlp


# SIG3D_NESTED_FPS=0 runs the dependent rounds on every level (A/B timing; results are identical)
NESTED_FPS = os.environ.get("SIG3D_NESTED_FPS", "1") != "0"
NESTED_CHAIN = True   # the nested levels prove their prefix in one launch pair (False: one after the other, round 2)


class Announced:
    """Identity of the batch whose geometry the forked branch computed during the previous replay.
    A device address is NOT an identity (the caching allocator recycles addresses; a loader that refills one
    staging buffer in place passes the same address every step), so the hand-over is keyed on
      * an explicit `token` (step / sequence id) when the caller passes one, else
      * the announced tensor OBJECT (kept referenced here, so its storage cannot be recycled) together with
        its autograd version counter, which every in-place write through any view of it advances.
    Anything else -- a different object, a refilled buffer, a missing token -- recomputes the geometry inline."""

    def __init__(self):
        self.tensor = self.version = self.token = None

    def set(self, tensor, token=None):
        self.tensor, self.version, self.token = tensor, tensor._version, token

    def matches(self, tensor, token=None):
        if token is not None or self.token is not None:
            return token is not None and token == self.token
        return tensor is self.tensor and tensor._version == self.version


class GeometryPlan:
    """levels: list of (npoint, radius, nsample) from the first SA layer down."""

    def __init__(self, batch, n_points, levels, device):
        self.levels = list(levels)
        self.batch, self.n_points = batch, n_points
        self.inds, self.new_xyz, self.ball_idx, self._temp, self.compact = [], [], [], [], []
        self.fps_proven = []  # per level: (B,) int32, 1 where the nested-FPS proof held (levels >= 1)
        self._fps_work = None   # scratch of the block-list FPS (level 0, scenes above 8192 points)
        # waves per scene of that kernel (only with SIG3D_FPS_BLOCKS=1): 4 for a chain that runs beside a training step
        # (-0.07 ms per step in the steady state; 16 waves: +0.27), 16 where the chain's own latency bounds the
        # throughput (serve.py)
        self.fps_waves = 4
        self._bq_work = None    # scratch of the multi-level ball query (allocated once: static under hipGraph)
        self._bq_clean = False  # True once the workspace has been through a call (counters zero again)
        self._bq_levels = None  # its level table (ctypes array of sig3d_bq_level: raw pointers of the plan's buffers)
        n = n_points
        for npoint, radius, nsample in self.levels:
            self.inds.append(torch.zeros(batch, npoint, dtype=torch.int32, device=device))
            self.new_xyz.append(torch.zeros(batch, npoint, 3, dtype=torch.float32, device=device))
            self.ball_idx.append(torch.zeros(batch, npoint, nsample, dtype=torch.int32, device=device))
            self._temp.append(torch.zeros(batch, max(n, 128), dtype=torch.float32, device=device))
            self.fps_proven.append(torch.zeros(batch, dtype=torch.int32, device=device))
            # distinct neighbours of the padded lists (csrc/compact.hip), for the levels that run the MFMA path
            big = fused_mlp.COMPACT and batch * npoint * nsample >= fused_mlp.COMPACT_MIN_POSITIONS
            self.compact.append(fused_mlp.CompactLists(batch, npoint, nsample, device) if big else None)
            n = npoint

    def compute(self, xyz):
        """Launch the chain for xyz (B,N,3) on the current stream; fills the plan's buffers."""
        dev = _lib.require_device(xyz)
        if not xyz.is_contiguous() or xyz.dtype != torch.float32:
            raise RuntimeError("xyz must be a contiguous float tensor")
        b, n, _ = xyz.shape
        assert (b, n) == (self.batch, self.n_points)
        s = _lib.stream_ptr(dev)
        cur = xyz
        with torch.cuda.device(dev):
            timeline.mark("geo:start")
            srcs = []
            chain = self._nested_chain(b) if NESTED_FPS and NESTED_CHAIN else None
            for lvl, (npoint, radius, nsample) in enumerate(self.levels):
                if chain is not None and lvl >= 1:
                    if lvl == 1:   # levels 1.. sample the FPS-ordered centres of level 0: one proof for all of them
                        _lib.call("sig3d_fps_nested_chain", b, n, len(self.levels) - 1, chain[0], _lib.ptr(cur),
                                  _lib.ptr(self._temp[1]), chain[1], chain[2], _lib.ptr(self._chain_flags), s)
                    timeline.mark("geo:L%d fps" % (lvl + 1))
                    srcs.append(cur)
                    cur, n = self.new_xyz[lvl], npoint
                    continue
                if (lvl == 0 or not NESTED_FPS) and _ext.FPS_BLOCKS and n > 8192:
                    if self._fps_work is None:
                        self._fps_work = _lib.fps_workspace(b, n, dev)
                    _lib.call("sig3d_furthest_point_sampling_blocks", b, n, npoint, _lib.ptr(cur),
                              _lib.ptr(self._fps_work), self._fps_work.numel(), int(self.fps_waves),
                              _lib.ptr(self.inds[lvl]), s)
                elif lvl == 0 or not NESTED_FPS:
                    _lib.call("sig3d_furthest_point_sampling", b, n, npoint, _lib.ptr(cur),
                              _lib.ptr(self._temp[lvl]), _lib.ptr(self.inds[lvl]), s)
                else:
                    # `cur` is the FPS-ordered output of the level above: prove inds == 0..m-1 instead
                    # of running the dependent rounds (csrc/sampling.hip, same results for any input)
                    _lib.call("sig3d_furthest_point_sampling_nested", b, n, npoint, _lib.ptr(cur),
                              _lib.ptr(self._temp[lvl]), _lib.ptr(self.inds[lvl]),
                              _lib.ptr(self.fps_proven[lvl]), s)
                timeline.mark("geo:L%d fps" % (lvl + 1))
                if lvl == 0 and self._stop_after("fps0"):
                    return self
                _lib.call("sig3d_gather_xyz", b, n, npoint, _lib.ptr(cur), _lib.ptr(self.inds[lvl]),
                          _lib.ptr(self.new_xyz[lvl]), s)
                srcs.append(cur)
                cur, n = self.new_xyz[lvl], npoint
            if self._stop_after("sampling"):
                return self
            # the neighbour lists of ALL levels depend on coordinates only: one scatter + one rank launch
            # (csrc/ball_query.hip: centres binned into cells, points streamed once) instead of a chain per level
            ptrs = [t.data_ptr() for t in srcs]
            if self._bq_levels is None or self._bq_srcs != ptrs:   # the level table holds raw pointers
                probs = [(srcs[l], self.new_xyz[l], lv[1], lv[2], self.ball_idx[l]) for l, lv in enumerate(self.levels)]
                self._bq_levels = _lib.bq_levels(probs)
                need = _lib.bq_levels_workspace_bytes(b, self._bq_levels)
                assert need >= 0, "too many centres for one multi-level ball query"
                if self._bq_work is None or self._bq_work.numel() < need:
                    self._bq_work = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
                self._bq_srcs = ptrs
                self._bq_clean = False   # a new workspace / problem list: the next call zeroes the counters itself
            # after one completed call the workspace's counters are zero again (the rank kernel cleans up): no memset
            _lib.call("sig3d_ball_query_levels_ex", b, len(self._bq_levels), self._bq_levels, _lib.ptr(self._bq_work),
                      self._bq_work.numel(), _lib.BQ_CLEAN if self._bq_clean else 0, s)
            # the flag describes the workspace as the DEVICE will find it at the next call: only a call that was
            # really issued (not one recorded into a graph that may never be replayed) leaves the counters zero
            self._bq_clean = not torch.cuda.is_current_stream_capturing()
            if self._stop_after("ballquery"):
                return self
            for lvl in range(len(self.levels)):
                if self.compact[lvl] is not None:
                    self.compact[lvl].compute(self.ball_idx[lvl])
            timeline.mark("geo:lists")
        return self

    def _stop_after(self, stage):
        """Measurement hook: tools/probes/geo_probes.py replaces this to cut a captured chain short after `stage`
        ("fps0", "sampling", "ballquery").  The product never stops early and reads no environment variable here."""
        return False

    def _nested_chain(self, b):
        """ctypes tables of sig3d_fps_nested_chain for levels 1.. (None when the chain's size limits do not hold)."""
        if getattr(self, "_chain", False) is not False:
            return self._chain
        self._chain = None
        deeper = self.levels[1:]
        ns = [lv[0] for lv in self.levels[:-1]]
        if 1 <= len(deeper) <= 4 and all(lv[0] <= n_in <= 8192 and lv[0] <= 4096 for lv, n_in in zip(deeper, ns)):
            m = (ctypes.c_int * len(deeper))(*[lv[0] for lv in deeper])
            idxs = (ctypes.c_void_p * len(deeper))(*[t.data_ptr() for t in self.inds[1:]])
            cent = (ctypes.c_void_p * len(deeper))(*[t.data_ptr() for t in self.new_xyz[1:]])
            self._chain_flags = torch.zeros(len(deeper), b, dtype=torch.int32, device=self.inds[0].device)
            for k in range(len(deeper)):                        # fps_proven[lvl]: views of the chain's flags
                self.fps_proven[k + 1] = self._chain_flags[k]
            self._chain = (m, idxs, cent)
        return self._chain

    def level(self, i):
        return self.inds[i], self.new_xyz[i], self.ball_idx[i], self.compact[i]

    def copy_from(self, other):
        """Hand-over at the end of a step: every tensor of `other` into this plan, ONE launch
        (table_copy.TableCopy) instead of one copy kernel per tensor."""
        pairs = list(zip(self.inds + self.new_xyz + self.ball_idx, other.inds + other.new_xyz + other.ball_idx))
        for mine, theirs in zip(self.compact, other.compact):
            if mine is not None:
                pairs += list(zip(mine.tensors(), theirs.tensors()))
        if getattr(self, "_table_copy", None) is None:
            from .table_copy import TableCopy
            self._table_copy = TableCopy(self.inds[0].device)
        self._table_copy(pairs)


# sig3d_ticket_wait gives up (and raises its error word) after this long.  A chain that gave up runs on coordinates
# that may not be staged yet, so the timeout is NOT a recovery path: it only keeps a crashed producer from hanging
# the device, and GeometryPipeline.advance() raises as soon as it sees the error word (a captured device-to-host copy
# at the end of every chain keeps a pinned host mirror of it current: no synchronisation).  The default is longer than
# any stall a healthy run produces (a straggler rank inside an all-reduce, a checkpoint written by rank 0) and longer
# than the process-group watchdog's 10 minutes.
_HANDSHAKE_TIMEOUT_US = int(float(os.environ.get("SIG3D_HANDSHAKE_TIMEOUT_S", "1200")) * 1e6)


class GeometryPipeline:
    """The geometry chains of the next `depth` batches in flight, each a hipGraph of its own on a stream of its own,
    beside a consumer (training step / forward) that stays ONE linear graph on the caller's stream.

    Every in-flight batch has a slot {plan, coordinate buffer, stream, graph}.  `advance(batch i)`:
      1. waits for the chain of batch i (launched `depth` calls ago) and hands its plan over to `plan_cur`, the plan
         the consumer's graph reads (one table copy) -- or computes it inline when the caller broke the announced
         order (graph_step.Announced: tokens, or tensor object + version; never a device address);
      2. stages the coordinates of batch i + depth and launches its chain on the slot that has just become free.
    depth 1 is round 2's "geometry of batch i+1 under step i".  With the step at ~7.4 ms and ONE chain taking ~7.5 ms
    beside it (3.6 ms alone: the FPS rounds wait on a loaded memory system), the chain had become the critical path
    again; two chains in flight give each of them two steps.

    The chains start behind a device-side handshake (sig3d_ticket_signal on the consumer's stream once the
    coordinates are staged, sig3d_ticket_wait as the slot graph's first node) instead of slot.stream.wait_stream():
    the host runs ahead, so that barrier packet would sit blocked at the head of the slot's hardware queue, and the
    command processor's polling of a blocked barrier costs every kernel the consumer dispatches meanwhile ~1.7 us
    (tools/probes/fork_penalty.py).  `handshake=False` restores the stream wait."""

    def __init__(self, batch, n_points, levels, device, stream, depth=1, handshake=True, stream_priority=0,
                 example_xyz=None, fps_waves=4):
        """fps_waves: waves per scene of the block-list FPS in the chains (GeometryPlan.fps_waves): 4 beside a training
        step (the chain has steps of slack and must not take CU slots from the step), 16 where the chains' own latency
        bounds the throughput (forward-only serving)."""
        assert depth >= 1
        self.depth, self.stream, self.device, self.handshake = int(depth), stream, device, bool(handshake)
        self.plan_cur = GeometryPlan(batch, n_points, levels, device)
        self.plan_cur.fps_waves = fps_waves
        self.slots = []
        self._words = torch.zeros(4 * self.depth, dtype=torch.int32, device=device)   # per slot: ticket, consumed, error
        # the slots' buffers start from real coordinates when the caller has some (a plan of all-zero points is valid
        # but degenerate: every ball holds the same 64 points)
        example = (example_xyz.detach().to(device=device, dtype=torch.float32).contiguous() if example_xyz is not None
                   else torch.zeros(batch, n_points, 3, dtype=torch.float32, device=device))
        # host mirror of the slots' error words (pinned: the copy at the end of a chain is an asynchronous graph node)
        self._host_err = torch.zeros(self.depth, dtype=torch.int32).pin_memory()
        # HIP multiplexes streams onto 4 hardware queues per priority, and which queue a new stream gets is not under
        # the caller's control (tools/probes/queue_map_probe.py: two fresh streams share one in ~1 of 4 cases).  A
        # chain on the consumer's queue is simply served in order with it (step 8.1 -> 11.8 ms): every slot stream is
        # checked to run concurrently with the consumer's stream and with the other slots' (streams.stream_beside).
        from . import streams
        beside = [stream]
        self.shared_queues = 0      # slots that found no free hardware queue (depth > 3): they serialise with another
        for k in range(self.depth):
            st, ok = streams.stream_beside(beside, device, priority=stream_priority)
            beside.append(st)
            self.shared_queues += 0 if ok else 1
            self.slots.append(dict(plan=GeometryPlan(batch, n_points, levels, device), xyz=example.clone(),
                                   stream=st, graph=torch.cuda.CUDAGraph(),
                                   announced=Announced(), words=self._words[4 * k:4 * k + 4]))
            self.slots[-1]["plan"].fps_waves = fps_waves
        for slot in self.slots:                       # scratch allocations and copy tables, outside any capture
            slot["plan"].compute(slot["xyz"])
            self.plan_cur.copy_from(slot["plan"])
        self.calls = 0
        self.inline_chains = 0    # batches whose geometry had to be computed on the consumer's stream

    def capture(self, pool=None, capture_error_mode="global"):
        """Capture the slot graphs (after the consumer's graph, whose memory pool they may share: a chain allocates
        nothing)."""
        for slot in self.slots:
            w = slot["words"]
            slot["stream"].wait_stream(self.stream)
            kw = dict(pool=pool) if pool is not None else {}
            with torch.cuda.graph(slot["graph"], stream=slot["stream"], capture_error_mode=capture_error_mode, **kw):
                if self.handshake:
                    _lib.call("sig3d_ticket_wait", _lib.ptr(w[0:1]), _lib.ptr(w[1:2]), _HANDSHAKE_TIMEOUT_US,
                              _lib.ptr(w[2:3]), _lib.stream_ptr(self.device))
                self._slot_body(slot)
                if self.handshake:
                    k = self.slots.index(slot)
                    self._host_err[k:k + 1].copy_(w[2:3], non_blocking=True)
            self.stream.wait_stream(slot["stream"])

    def _slot_body(self, slot):
        """What a slot's graph runs behind its ticket wait: the chain.  (tools/probes/geo_probes.py swaps in idle
        kernels or nothing to measure what the chain costs the step beside it; the product never does.)"""
        slot["plan"].compute(slot["xyz"])

    def _skip_chain(self, point_clouds):
        """Measurement hook (tools/probes/geo_probes.py); the product always runs its chains."""
        return False

    def advance(self, point_clouds, upcoming, token=None, upcoming_tokens=None):
        """`point_clouds`: (B,N,3+C) of the batch the consumer is about to run; `upcoming`: those of the next `depth`
        calls, in order (upcoming[-1] is the one whose chain starts now).  Call on the consumer's stream, BEFORE
        replaying its graph."""
        if len(upcoming) != self.depth:
            raise ValueError("a geometry pipeline of depth %d needs the next %d batches" % (self.depth, self.depth))
        toks = list(upcoming_tokens) if upcoming_tokens is not None else [None] * self.depth
        slot = self.slots[self.calls % self.depth]
        if self._host_err.any():
            raise RuntimeError(
                "geometry pipeline: a chain gave up waiting for its ticket after %.0f s (slot error words %s): the "
                "consumer's stream was stalled for that long, and the plan of at least one batch since then was "
                "computed from coordinates that were not staged yet.  Steps taken since are not trustworthy; restart "
                "from the last checkpoint (SIG3D_HANDSHAKE_TIMEOUT_S raises the limit, SIG3D_GEO_HANDSHAKE=0 orders "
                "the chains with stream waits instead)." % (_HANDSHAKE_TIMEOUT_US / 1e6, self._host_err.tolist()))
        if self._skip_chain(point_clouds):
            self.calls += 1
            return
        self.stream.wait_stream(slot["stream"])              # the chain launched `depth` calls ago (any chain: its
        if slot["announced"].matches(point_clouds, token):   # coordinate buffer is about to be overwritten)
            self.plan_cur.copy_from(slot["plan"])
        else:                                                # prologue, or the caller broke the announced order
            self.plan_cur.compute(point_clouds[..., :3].contiguous())
            self.inline_chains += 1
        far = upcoming[-1]
        slot["xyz"].copy_(far[..., :3], non_blocking=True)
        slot["announced"].set(far, toks[-1])
        if self.handshake:                                   # coordinates staged, plan handed over: the slot may start
            with torch.cuda.device(self.device):
                _lib.call("sig3d_ticket_signal", _lib.ptr(slot["words"][0:1]), self.stream.cuda_stream)
        else:
            slot["stream"].wait_stream(self.stream)
        with torch.cuda.stream(slot["stream"]):
            slot["graph"].replay()
        self.calls += 1

    def timed_out(self):
        """True when a chain ever gave up waiting for its ticket (it then ran anyway: a plan may be stale).
        Synchronises; advance() checks the pinned host mirror of the same words on every call without synchronising."""
        return bool(self._words.view(self.depth, 4)[:, 2].any().item())

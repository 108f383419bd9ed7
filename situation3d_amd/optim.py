"""FlatAdamW: value clip + AdamW as ONE streaming HIP kernel (csrc/optim.hip), for the update of
lib/solver.py:618-627 / situation3d/train/train.py:226-238.

Storage: parameters and both moments of a parameter group live in flat float32 buffers (every
`p.data` is a view; module state_dict keys and shapes are untouched).  Gradients are NOT flat:
autograd hands each parameter a fresh tensor when `.grad is None` (no kernel), whereas a
persistent flat `.grad` view costs one accumulate-add launch per parameter (+315 launches per
step for this model, measured).  So `step()` walks a device-resident table of 64 Ki-element
chunks {p, g, m, v, n, weight_decay} -- one launch for the whole model -- and then drops the
gradients (`zero_grad(set_to_none=True)` semantics of the reference's Solver loop).

Data parallel: `gather_grads()` copies the scattered gradients into flat buffers (one launch),
`flat_grad_buffers()` exposes them for an in-place bucketed all-reduce
(ddp.GradBucketReducer.from_flat) and the following `step()` consumes the flat, reduced gradients.

Numerically this is torch.optim.AdamW (amsgrad=False) preceded by clip_grad_value_.  A parameter
that received no gradient in a step is left untouched on EVERY path (table, gathered, bucketed), like
torch.optim.AdamW and the reference's DDP with find_unused_parameters=True (runner_base.py:91-93).
The learning rate lives in a device scalar the kernels read when they EXECUTE: a scheduler that
mutates `param_groups[*]["lr"]` (lib/solver.py:239-247) is honoured by captured hipGraph replays too --
`sync_lr()` (called by step() and by graph_step.GraphedTrainStep before every replay) refreshes it.
`state_dict()` / `load_state_dict()` speak torch.optim.AdamW's layout (per-parameter `step`, `exp_avg`,
`exp_avg_sq`, indexed by position in `param_groups[*]["params"]` = the caller's `named_parameters()` order;
the flat STORAGE order is a separate argument), so the reference's checkpoints (lib/solver.py:652-660,
train.py:256-262) round-trip in both directions.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib

_CHUNK = 65536
_SPARE_TABLES = 2        # staging tables kept ready for a capture nobody reserved tables for (see CaptureTables)
_REC = np.dtype([("p", "u8"), ("g", "u8"), ("m", "u8"), ("v", "u8"), ("n", "i8"), ("wd", "f4"),
                 ("pad", "f4")])


# workgroups that walk the chunk table of one update launch when the optimizer is not given max_workgroups
# (0: one workgroup per 64 Ki-element chunk)
DEFAULT_MAX_WORKGROUPS = 0


def flat_offsets(params):
    """Element offsets of `params` laid back to back with 16-byte alignment (the flat buffers' rule) -> (offsets, total)."""
    offs, total = [], 0
    for p in params:
        offs.append(total)
        total += (p.numel() + 3) // 4 * 4
    return offs, total


def part_runs(mine, total, parts, exclude=()):
    """mine: [(parameter, element offset)] of one flat buffer in storage order, `total` its length.  parts: lists of
    parameters.  -> one list of (lo, hi) element ranges per part -- every maximal run of the part's parameters is a
    range -- plus a last list for the parameters of no part; parameters in `exclude` belong to no range."""
    owner = {}
    for k, ps in enumerate(parts):
        for p in ps:
            owner[id(p)] = k
    for p in exclude:
        owner[id(p)] = -1
    out = [[] for _ in range(len(parts) + 1)]
    i = 0
    while i < len(mine):
        k = owner.get(id(mine[i][0]), len(parts))
        j = i
        while j < len(mine) and owner.get(id(mine[j][0]), len(parts)) == k:
            j += 1
        lo = mine[i][1]
        hi = mine[j][1] if j < len(mine) else total
        if k >= 0:
            out[k].append((lo, hi))
        i = j
    return out


class CaptureTables:
    """The pinned staging buffers + device tables of the gradient-table uploads of ONE captured step: a captured upload
    is a memcpy node that re-reads its pinned staging buffer at every replay, so every upload call site of every live
    capture needs a buffer of its own -- two graphs over one optimizer (a step rebuilt after a BatchNorm momentum
    change while the old one is still replayed, the arms of tools/ab_step.py) must never share one.  Built OUTSIDE a
    capture (pinned memory cannot be allocated inside one) by FlatAdamW.new_capture_tables(), owned by whoever owns
    the graphs (graph_step.GraphedTrainStep): dropped with them, so there is no cap on the number of captures."""

    def __init__(self, sets):
        self.sets, self.used = list(sets), 0

    def take(self):
        if self.used >= len(self.sets):
            raise RuntimeError("FlatAdamW: this capture uploads more gradient tables than were reserved for it "
                               "(new_capture_tables(sites)): %d" % len(self.sets))
        self.used += 1
        return self.sets[self.used - 1]


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 clip_value=0.0, storage_order=None, max_workgroups=None):
        """storage_order: optional, one list per parameter group -- the SAME trainable parameters in the order
        they are to lie in the flat buffers (trainer.build_optimizer keeps operands the Q-Former stacks into one
        GEMM adjacent).  `param_groups[*]["params"]` -- the order state_dict() / load_state_dict() index by, like
        torch.optim -- stays the caller's order, so checkpoints are independent of the storage layout."""
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.clip_value = float(clip_value)
        # None: one workgroup per 64 Ki-element chunk.  k: at most k workgroups walk the chunk table
        # (sig3d_adamw_table_bounded) -- for an update that shares the chip with another stream's kernels; 256 (one
        # per CU) streams as fast as the full grid (tools/adamw_overlap.py: 0.91 vs 0.93 ms)
        self.max_workgroups = None if max_workgroups is None else max(1, int(max_workgroups))
        self._groups = []
        dev = None
        recs, owners = [], []   # static part of the chunk table; owners[i] = index into self._params
        self._params = []
        if storage_order is not None and len(storage_order) != len(self.param_groups):
            raise ValueError("storage_order needs one list per parameter group")
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.requires_grad]
            if storage_order is not None:
                layout = [p for p in storage_order[gi] if p.requires_grad]
                if len(layout) != len(ps) or {id(p) for p in layout} != {id(p) for p in ps}:
                    raise ValueError("storage_order[%d] is not a permutation of the group's parameters" % gi)
                ps = layout
            if gi > 0 and (group["lr"], group["betas"], group["eps"]) != \
                    (self.param_groups[0]["lr"], self.param_groups[0]["betas"], self.param_groups[0]["eps"]):
                raise RuntimeError("FlatAdamW: groups may differ in weight_decay only")
            for p in ps:
                if not p.is_cuda or p.dtype != torch.float32:
                    raise RuntimeError("FlatAdamW needs float32 parameters on the GPU")
            if not ps:
                self._groups.append(None)
                continue
            dev = ps[0].device
            offs, total = flat_offsets(ps)   # 16-byte alignment of every parameter inside the flat buffers
            flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
            m, v = torch.zeros_like(flat_p), torch.zeros_like(flat_p)
            for p, off in zip(ps, offs):
                flat_p[off:off + p.numel()].copy_(p.data.reshape(-1))
                p.data = flat_p[off:off + p.numel()].view_as(p)
                p.grad = None
                pi = len(self._params)
                self._params.append((p, gi, off))
                for c0 in range(0, p.numel(), _CHUNK):
                    n = min(_CHUNK, p.numel() - c0)
                    byte = 4 * (off + c0)
                    recs.append((flat_p.data_ptr() + byte, 4 * c0, m.data_ptr() + byte,
                                 v.data_ptr() + byte, n, group["weight_decay"], 0.0))
                    owners.append(pi)
            self._groups.append(dict(p=flat_p, m=m, v=v, g=None, total=total))
        self._dev = dev
        self._step = torch.zeros((), dtype=torch.float32, device=dev)
        self._lr_host = float(self.param_groups[0]["lr"])
        self._lr_dev = torch.full((1,), self._lr_host, dtype=torch.float32, device=dev)
        self._live = np.zeros(len(self._params), dtype=bool)   # which parameters the last gather saw a gradient for
        self._external = []   # parameters whose flat gradient slot is filled by someone else (see below)
        self._aliased = {}    # group index -> [(lo, hi)] element ranges of the flat gradients that producers write in place
        self._flat_tables = {}
        self._static = np.array(recs, dtype=_REC)          # 'g' holds the byte offset inside the grad
        self._owners = np.array(owners, dtype=np.int64)
        self._host = torch.empty(len(recs) * _REC.itemsize, dtype=torch.uint8).pin_memory()
        self._host_np = self._host.numpy().view(_REC)
        self._table = torch.empty(len(recs) * _REC.itemsize, dtype=torch.uint8, device=dev)
        self.capture_tables = None     # CaptureTables of the capture in progress (captured_with)
        self._spare = [self._new_table_set() for _ in range(_SPARE_TABLES)]
        self._anonymous = []           # spares that a capture without reserved tables took: alive as long as the optimizer
        self._gathered = False

    # ---- chunk table ------------------------------------------------------------------------
    def _new_table_set(self):
        host = torch.empty(len(self._static) * _REC.itemsize, dtype=torch.uint8).pin_memory()
        return host, host.numpy().view(_REC), torch.empty_like(self._table)

    def new_capture_tables(self, sites=4):
        """Staging tables for the uploads of one captured step (`sites` upload call sites: step() or every
        gather_grads() of the capture).  Call outside any capture, keep the result alive with the graphs, capture
        inside `with optimizer.captured_with(tables):`."""
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("FlatAdamW.new_capture_tables: pinned memory cannot be allocated during a capture")
        return CaptureTables([self._new_table_set() for _ in range(sites)])

    def captured_with(self, tables):
        import contextlib

        @contextlib.contextmanager
        def scope():
            prev, self.capture_tables = self.capture_tables, tables
            try:
                yield tables
            finally:
                self.capture_tables = prev
        return scope()

    def _table_set(self):
        """(pinned host staging, its numpy view, device table) for the upload being issued now.  Eager: the shared set
        (the copy blocks, the buffer is free again when it returns).  Inside a capture: a set of its own -- from the
        capture's CaptureTables, else one of the spares (refilled by the next eager upload)."""
        if not torch.cuda.is_current_stream_capturing():
            while len(self._spare) < _SPARE_TABLES:
                self._spare.append(self._new_table_set())
            return self._host, self._host_np, self._table
        if self.capture_tables is not None:
            return self.capture_tables.take()
        if not self._spare:
            raise RuntimeError("FlatAdamW: a capture needs gradient tables of its own and the spares are used up -- "
                               "reserve them before capturing: tables = optimizer.new_capture_tables(sites); "
                               "with optimizer.captured_with(tables): <capture>")
        self._anonymous.append(self._spare.pop())
        return self._anonymous[-1]

    def _upload(self, dst_field_from_flat_g=False):
        """Fill the per-step columns of the table (gradient pointers) and push it to the device."""
        gptr = np.zeros(len(self._params), dtype=np.uint64)
        live = np.zeros(len(self._params), dtype=bool)
        for i, (p, gi, off) in enumerate(self._params):
            g = p.grad
            if g is None:
                continue
            if not g.is_contiguous():
                g = g.contiguous()
                p.grad = g
            gptr[i] = g.data_ptr()
            live[i] = True
        if dst_field_from_flat_g:
            self._live |= self._agree_on_liveness(live)
            self._live[self._external] = True
        host, t, table = self._table_set()
        t[:] = self._static
        t["g"] = gptr[self._owners] + self._static["g"]
        t["n"] = np.where(live[self._owners], self._static["n"], 0)  # params without grad: skipped
        if dst_field_from_flat_g:  # gather: destination = flat gradient storage (passed in 'm')
            base = np.zeros(len(self._params), dtype=np.uint64)
            for i, (p, gi, off) in enumerate(self._params):
                base[i] = self._groups[gi]["g"].data_ptr() + 4 * off
            t["m"] = base[self._owners] + self._static["g"]
            t["n"] = np.where(t["g"] == t["m"], 0, t["n"])   # the gradient already lives in its flat slot
        # eager: blocking copy (the pinned staging buffer is rewritten next step); inside a hipGraph
        # capture the copy becomes a memcpy node reading the (then static) staging buffer
        table.copy_(host, non_blocking=torch.cuda.is_current_stream_capturing())
        return table

    def _agree_on_liveness(self, live):
        """Data parallel: the gradients are averaged over ranks, so WHICH parameters get an AdamW record must be
        the same everywhere -- a parameter is live when ANY rank produced a gradient for it (torch DDP with
        find_unused_parameters treats one as unused only when it is unused on all ranks; runner_base.py:91-93).
        One small MAX all-reduce in the eager data-parallel step (where the set can depend on the data); a captured
        step has one static launch structure on every rank and skips it."""
        import torch.distributed as dist
        # the ranks that AVERAGE these gradients: the reducer's group when one was named (`process_group`
        # attribute, set by whoever pairs this optimizer with a ddp.GradBucketReducer), else the default group
        group = getattr(self, "process_group", None)
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1) \
                or torch.cuda.is_current_stream_capturing():
            return live
        # the set only changes when the data changes which parameters are used: a rank that sees the mask it agreed
        # on last time skips the exchange -- if EVERY rank does (one 1-byte MAX all-reduce of "mine changed")
        changed = self.__dict__.get("_live_agreed_for") is None or not np.array_equal(self._live_agreed_for, live)
        dev = self._dev if dist.get_backend(group) == "nccl" else "cpu"
        flag = torch.tensor([1 if changed else 0], dtype=torch.uint8, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if int(flag.item()) == 0:
            return self._live_agreed
        mask = torch.from_numpy(live.astype(np.uint8)).to(dev)
        dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=group)
        self._live_agreed_for, self._live_agreed = live.copy(), mask.cpu().numpy().astype(bool)
        return self._live_agreed

    # ---- public ----------------------------------------------------------------------------------
    def flat_grad_buffers(self):
        for g in self._groups:
            if g is not None and g["g"] is None:
                g["g"] = torch.zeros(g["total"], dtype=torch.float32, device=self._dev)
        return [g["g"] for g in self._groups if g is not None]

    def gather_grads(self, zero=True):
        """Scattered .grad tensors -> flat gradient buffers (one launch); the next step() then reads
        the flat (e.g. all-reduced) gradients.  Parameters whose .grad is None are skipped, so a
        backward pass run in two parts gathers in two calls: the first with zero=True (whole buffers
        cleared), the second with zero=False.  Inside a capture every call gets a staging table of its
        own (_table_set)."""
        self.flat_grad_buffers()
        if zero:
            self._live[:] = False
            self._zero_unaliased()  # parameters without a gradient contribute zeros to the all-reduce
        table = self._upload(dst_field_from_flat_g=True)
        with torch.cuda.device(self._dev):
            _lib.call("sig3d_gather_table", len(self._static), _lib.ptr(table),
                      _lib.stream_ptr(self._dev))
        self._gathered = True
        for p, _, _ in self._params:
            p.grad = None

    def flat_grad_run(self, params):
        """The slice of a flat gradient buffer that covers exactly `params`, in this order, or None when they do
        not lie back to back in one group's storage.  A producer that writes the gradients of these parameters
        straight into the slice (qformer._WeightGradArena) hands autograd views of it as `.grad`; gather_grads()
        then neither copies nor zeroes the range."""
        self.flat_grad_buffers()
        index = {id(p): (gi, off) for p, gi, off in self._params}
        if not params or any(id(p) not in index for p in params):
            return None
        gi, lo = index[id(params[0])]
        pos = lo
        for p in params:
            g2, off = index[id(p)]
            if g2 != gi or off != pos or p.numel() % 4 != 0:
                return None
            pos += p.numel()
        runs = self._aliased.setdefault(gi, [])
        if (lo, pos) not in runs:
            runs.append((lo, pos))
            runs.sort()
        return self._groups[gi]["g"][lo:pos]

    def _zero_unaliased(self):
        """Zero the flat gradient buffers except the ranges their producers overwrite completely every step."""
        for gi, g in enumerate(self._groups):
            if g is None:
                continue
            pos = 0
            for lo, hi in self._aliased.get(gi, ()):
                if lo > pos:
                    g["g"][pos:lo].zero_()
                pos = max(pos, hi)
            if pos < g["total"]:
                g["g"][pos:].zero_()

    def mark_externally_reduced(self, param):
        """`param` gets no .grad from autograd: its slot of the flat gradient buffer is filled from outside
        between gather_grads() and the update (ddp.SparseRowExchange for an embedding table).  It stays live
        for the bucketed update.  -> (group index, lo, hi, view of the slot shaped like the parameter)."""
        self.flat_grad_buffers()
        for i, (p, gi, off) in enumerate(self._params):
            if p is param:
                if i not in self._external:
                    self._external.append(i)
                return gi, off, off + p.numel(), self._groups[gi]["g"][off:off + p.numel()].view_as(p)
        raise KeyError("not a parameter of this optimizer")

    def mark_gathered(self):
        """A replayed hipGraph ran gather_grads(): the flat gradient buffers are current."""
        self._gathered = True

    def flat_grad_split(self, part_params):
        """Slices of the flat gradient buffers covering exactly `part_params` and exactly the rest, as
        ([part slices], [rest slices]) -- or None when `part_params` do not form ONE contiguous run in
        a group's storage order (they do for a submodule such as the point encoder)."""
        self.flat_grad_buffers()
        ids = {id(p) for p in part_params}
        part, rest = [], []
        for gi, f in enumerate(self._groups):
            if f is None:
                continue
            mine = [(p, off) for p, g2, off in self._params if g2 == gi]
            flags = [id(p) in ids for p, _ in mine]
            if not any(flags):
                rest.append(f["g"])
                continue
            k0 = flags.index(True)
            k1 = len(flags) - flags[::-1].index(True)
            if not all(flags[k0:k1]):
                return None
            lo = mine[k0][1]
            hi = mine[k1][1] if k1 < len(mine) else f["total"]
            part.append(f["g"][lo:hi])
            if lo > 0:
                rest.append(f["g"][:lo])
            if hi < f["total"]:
                rest.append(f["g"][hi:])
        return part, rest

    def flat_grad_parts(self, parts, exclude=()):
        """parts: list of parameter lists.  -> one list of flat-gradient slices per part (every maximal run of
        the part's parameters in storage order is a slice) plus a last list for all other parameters.
        Parameters in `exclude` belong to no list (their slot is exchanged some other way)."""
        self.flat_grad_buffers()
        out = [[] for _ in range(len(parts) + 1)]
        for gi, f in enumerate(self._groups):
            if f is None:
                continue
            mine = [(p, off) for p, g2, off in self._params if g2 == gi]
            for k, runs in enumerate(part_runs(mine, f["total"], parts, exclude)):
                out[k] += [f["g"][lo:hi] for lo, hi in runs]
        return out

    @torch.no_grad()
    def step_after(self, reducer):
        """Data-parallel update overlapped with the gradient exchange: every bucket's all-reduce is
        launched at once (ddp.GradBucketReducer over flat_grad_buffers(), after gather_grads()), then
        the AdamW kernel runs bucket by bucket as soon as that bucket's collective has completed --
        the 1 ms HBM-bound update hides under the remaining collectives instead of following them."""
        self.process_group = reducer.group     # liveness is agreed among the ranks that share these gradients
        reducer.launch_all()
        self.begin_bucketed_step()
        self.update_buckets(reducer)
        self.end_bucketed_step()

    def begin_bucketed_step(self):
        assert self._gathered, "call gather_grads() first"
        with torch.cuda.device(self._dev):
            _lib.call("sig3d_step_increment", _lib.ptr(self._step), _lib.stream_ptr(self._dev))

    def end_bucketed_step(self):
        self._gathered = False

    # ---- learning rate / liveness ---------------------------------------------------------------
    def sync_lr(self):
        """Push a changed `param_groups[*]["lr"]` to the device scalar the kernels read.  Never inside a
        capture (the fill would be baked in): graph_step.GraphedTrainStep calls this before each replay."""
        lr = float(self.param_groups[0]["lr"])
        if any(float(g["lr"]) != lr for g in self.param_groups):
            raise RuntimeError("FlatAdamW: groups may differ in weight_decay only")
        if lr != self._lr_host and not torch.cuda.is_current_stream_capturing():
            self._lr_dev.fill_(lr)
            self._lr_host = lr

    def _flat_table(self, ranges):
        """Device chunk table for AdamW over slices [(group, lo, hi), ...] of the FLAT storage (gradients
        included), restricted to the parameters the last gather saw a gradient for: dead parameters get
        no record, so they are neither decayed nor stepped.  Cached per (ranges, liveness)."""
        key = (tuple(ranges), self._live.tobytes())
        hit = self._flat_tables.get(key)
        if hit is not None:
            return hit[0], hit[1]
        recs = []
        for gi, lo, hi in ranges:
            f = self._groups[gi]
            wd = self.param_groups[gi]["weight_decay"]
            for pi, (p, g2, off) in enumerate(self._params):
                if g2 != gi or not self._live[pi]:
                    continue
                a, b = max(off, lo), min(off + p.numel(), hi)
                for c0 in range(a, b, _CHUNK):
                    n = min(_CHUNK, b - c0)
                    recs.append((f["p"].data_ptr() + 4 * c0, f["g"].data_ptr() + 4 * c0,
                                 f["m"].data_ptr() + 4 * c0, f["v"].data_ptr() + 4 * c0, n, wd, 0.0))
        arr = np.array(recs, dtype=_REC)
        host = torch.empty(max(len(recs), 1) * _REC.itemsize, dtype=torch.uint8).pin_memory()
        host.numpy().view(_REC)[:len(recs)] = arr
        table = torch.empty(host.numel(), dtype=torch.uint8, device=self._dev)
        table.copy_(host, non_blocking=torch.cuda.is_current_stream_capturing())
        if len(self._flat_tables) >= 64 and not torch.cuda.is_current_stream_capturing():
            # liveness that varies from step to step (eager, data-dependent graphs) must not grow the cache without
            # bound; tables referenced by a captured graph are only ever created during a capture and are kept
            for k in [k for k, v in self._flat_tables.items() if not v[3]][:32]:
                del self._flat_tables[k]
        # the staging buffer stays alive (memcpy node); [3]: created inside a capture
        self._flat_tables[key] = (table, len(recs), host, torch.cuda.is_current_stream_capturing())
        return table, len(recs)

    def _adamw_ranges(self, ranges, stream):
        table, n = self._flat_table(ranges)
        g0 = self.param_groups[0]
        b1, b2 = g0["betas"]
        self._launch_table(n, table, stream)

    def _launch_table(self, n, table, stream, max_workgroups=None):
        g0 = self.param_groups[0]
        b1, b2 = g0["betas"]
        args = [n, _lib.ptr(table), _lib.ptr(self._step), ctypes.c_float(g0["lr"]), _lib.ptr(self._lr_dev),
                ctypes.c_float(b1), ctypes.c_float(b2), ctypes.c_float(g0["eps"]), ctypes.c_float(self.clip_value)]
        bound = max_workgroups if max_workgroups is not None else \
            (self.max_workgroups if self.max_workgroups is not None else DEFAULT_MAX_WORKGROUPS)
        if not bound:
            _lib.call("sig3d_adamw_table", *args, stream)
        else:
            _lib.call("sig3d_adamw_table_bounded", *args, int(bound), stream)

    @torch.no_grad()
    def update_buckets(self, reducer):
        self.process_group = reducer.group
        return self._update_buckets(reducer)

    @torch.no_grad()
    def _update_buckets(self, reducer):
        """AdamW on every bucket of `reducer` (slices of flat_grad_buffers()), each right behind its own
        all-reduce; the collectives must have been launched (reducer.launch_all()).  The gradients stay in
        place: the next gather_grads(zero=True) clears the buffers."""
        dev = self._dev
        stream = _lib.stream_ptr(dev)
        self.sync_lr()
        with torch.cuda.device(dev):
            for bucket in reducer.buckets:
                reducer.wait(bucket)   # the current stream waits for this bucket's collective only
                flat = bucket["flat"]
                gi = None
                for f_i, f in enumerate(self._groups):   # which group's flat buffer is this a slice of?
                    if f is None:
                        continue
                    lo = f["g"].data_ptr()
                    if lo <= flat.data_ptr() < lo + 4 * f["total"]:
                        gi, off = f_i, (flat.data_ptr() - lo) // 4
                        break
                assert gi is not None, "bucket is not a slice of this optimizer's flat gradients"
                self._adamw_ranges([(gi, off, off + flat.numel())], stream)

    @torch.no_grad()
    def update_range(self, gi, lo, hi):
        """AdamW on flat elements [lo, hi) of group `gi` (inside a bucketed step, like one bucket)."""
        with torch.cuda.device(self._dev):
            self._adamw_ranges([(gi, lo, hi)], _lib.stream_ptr(self._dev))

    def zero_grad(self, set_to_none=True):
        for p, _, _ in self._params:
            p.grad = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        dev = self._dev
        stream = _lib.stream_ptr(dev)
        g0 = self.param_groups[0]
        b1, b2 = g0["betas"]
        self.sync_lr()
        with torch.cuda.device(dev):
            _lib.call("sig3d_step_increment", _lib.ptr(self._step), stream)
            if self._gathered:
                self._adamw_ranges([(gi, 0, f["total"]) for gi, f in enumerate(self._groups) if f is not None],
                                   stream)
                self._gathered = False
            else:
                table = self._upload()
                self._launch_table(len(self._static), table, stream)
                for p, _, _ in self._params:
                    p.grad = None
        return loss

    # ---- checkpointing (torch.optim.AdamW layout) ----------------------------------------------------
    def _indexed_params(self):
        """[(state index, parameter)] in torch's convention: positions in the concatenated group lists."""
        out, i = [], 0
        for group in self.param_groups:
            for p in group["params"]:
                out.append((i, p))
                i += 1
        return out

    def _moment_views(self):
        views = {}
        for p, gi, off in self._params:
            f = self._groups[gi]
            views[id(p)] = (f["m"][off:off + p.numel()].view_as(p), f["v"][off:off + p.numel()].view_as(p))
        return views

    def state_dict(self):
        """What torch.optim.AdamW(...).state_dict() would hold after the same steps (lib/solver.py:657
        saves it, train.py:262 loads it): state[i] = {step, exp_avg, exp_avg_sq} per parameter index."""
        views = self._moment_views()
        step = float(self._step.item())
        state = {}
        for i, p in self._indexed_params():
            if id(p) in views and step > 0:
                m, v = views[id(p)]
                state[i] = {"step": torch.tensor(step, dtype=torch.float32), "exp_avg": m.clone(),
                            "exp_avg_sq": v.clone()}
        groups, i = [], 0
        for group in self.param_groups:
            g = {k: v for k, v in group.items() if k != "params"}
            g.update(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False,
                     fused=None, decoupled_weight_decay=True)
            g["params"] = list(range(i, i + len(group["params"])))
            i += len(group["params"])
            groups.append(g)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, state_dict):
        """Accepts this class's and torch.optim.AdamW's / Adam's state_dict (same groups and parameter
        order): moments are copied into the flat buffers, the step counter is the (common) per-parameter
        `step`; parameters without an entry start from zero moments like in torch."""
        groups = state_dict["param_groups"]
        if len(groups) != len(self.param_groups) or any(
                len(a["params"]) != len(b["params"]) for a, b in zip(groups, self.param_groups)):
            raise ValueError("loaded state dict has a different number of parameter groups / parameters")
        for mine, theirs in zip(self.param_groups, groups):
            for k in ("lr", "eps", "weight_decay"):
                if k in theirs:
                    mine[k] = theirs[k]
            if "betas" in theirs:
                mine["betas"] = tuple(theirs["betas"])
            if theirs.get("amsgrad"):
                raise ValueError("FlatAdamW has no amsgrad state")
        views = self._moment_views()
        steps = set()
        for f in self._groups:
            if f is not None:
                f["m"].zero_()
                f["v"].zero_()
        for i, p in self._indexed_params():
            st = state_dict["state"].get(i)
            if st is None or id(p) not in views:
                continue
            m, v = views[id(p)]
            for key in ("exp_avg", "exp_avg_sq"):   # copy_ would broadcast a (768,) state into a (768, 768) slot
                if tuple(st[key].shape) != tuple(p.shape):
                    raise ValueError("state %d: %s has shape %s, the parameter at that position %s -- the loaded "
                                     "state dict orders its parameters differently"
                                     % (i, key, tuple(st[key].shape), tuple(p.shape)))
            m.copy_(st["exp_avg"])
            v.copy_(st["exp_avg_sq"])
            steps.add(float(st["step"]))
        if len(steps) > 1:
            raise ValueError("FlatAdamW keeps ONE step counter; the loaded per-parameter steps differ: %s"
                             % sorted(steps))
        self._step.fill_(steps.pop() if steps else 0.0)
        # weight decay sits in the static chunk table; learning rate in the device scalar
        for k, pi in enumerate(self._owners):
            self._static["wd"][k] = self.param_groups[self._params[pi][1]]["weight_decay"]
        # tables a captured graph replays (memcpy nodes re-read their pinned staging buffers) stay allocated; a graph
        # captured before this call keeps the weight decay it was captured with -- rebuild the step after loading
        self._retired = self.__dict__.get("_retired", []) + [v for v in self._flat_tables.values() if v[3]]
        self._flat_tables.clear()
        self.sync_lr()

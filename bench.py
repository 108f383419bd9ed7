"""bench.py -- headline metric of BASELINE.json on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: spawns its own N ranks, below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one training pass of the hot path over one synthetic batch:
point encoder (SA1-4 on 40 000 points / scene) -> situational pose re-encode -> Q-Former fusion
(32 queries + 20 question tokens) -> losses -> backward -> value clip -> AdamW.  B = 8 scenes per
GPU (BASELINE config "SQA3D train step fwd+bwd+Adam, 40k pts, B=8"), weak scaling over ranks
(one process per GPU, bucketed RCCL all-reduce overlapped with backward).  Inputs are generated
before the timed region and are resident in HBM.  Prints ONE JSON line on rank 0.

Launching (reference: `python -m torch.distributed.run --nproc_per_node=4 train.py`,
3DLLM_BLIP2-base/scripts/slurm_3dllm_run.slurm:30): under an external launcher (WORLD_SIZE set) this
process IS one rank.  Without one, `--gpus N` with N > 1 makes this process a PARENT that starts the N ranks
as child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment),
relays rank 0's JSON line as its own last line of stdout and exits non-zero when any rank fails.  The
parent decides that before torch or the HIP library is imported: it never touches a GPU and never
replaces itself with another program.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the dense-mode and surface-data lines printed beside the headline (N = 1 only)")
    ap.add_argument("--no-graph", action="store_true", help="issue every launch eagerly")
    ap.add_argument("--torch-adamw", action="store_true",
                    help="torch.optim.AdamW + clip_grad_value_ instead of the fused flat optimizer")
    ap.add_argument("--force-reducer", action="store_true",
                    help="use the data-parallel code path (flat gradient buckets, two graphs) at N=1")
    ap.add_argument("--no-gemm-tuning", action="store_true",
                    help="library GEMMs with the default heuristic instead of the tuned solutions")
    ap.add_argument("--no-ops-roofline", action="store_true",
                    help="skip the stand-alone dense op-level pair (roofline_ops); for --pmc passes over the step's own kernels")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="compute FPS/ball-query geometry inline instead of one batch ahead")
    ap.add_argument("--surface", action="store_true",
                    help="time the step on surface-shaped scenes (the 'synthetic-surface data' variant) INSTEAD of the "
                         "SURVEY 8d headline distribution -- for profiles; the line says so in `data`")
    # round 5: THREE chains in flight.  Since the main chain lost another ~0.5 ms in round 4 the step waited for its
    # geometry in every step (tools/probes/chain_slack.py: +0.05 ... +0.18 ms); with the chain two more steps ahead it
    # does not: -0.065 ms at depth 2, -0.094 at depth 3, +0.01 from 2 to 4 (tools/ab_step.py env:SIG3D_GEO_DEPTH)
    # round 6: measured at world size 1 behind a real RCCL group of one (tools/cut_ab.sh, three alternating pairs):
    # 8.93 -> 8.52 ms per step, exposed joins 0.39 -> 0.18 ms -- the second cut pays even before there is a wire
    ap.add_argument("--qf-cut", type=int, default=int(os.environ.get("SIG3D_QF_CUT", "6")),
                    help="data-parallel forms only: cut the backward pass after this Q-Former layer as well (three graphs, three "
                         "bucket sets; the optimizer stores the layers below / above it as two arenas) so that the upper "
                         "layers' gradients are on the wire while the lower layers compute; 0 = one cut, at the scene tokens")
    ap.add_argument("--geo-depth", type=int, default=int(os.environ.get("SIG3D_GEO_DEPTH", "3")),
                    help="geometry chains in flight beside the step (geometry.GeometryPipeline; 1 = round 2's one-ahead)")
    return ap.parse_args(argv)


def spawn_ranks(n, argv, script=None):
    """Parent of a self-launched N-rank run: one child per GPU, same command line (`script`: this file).  Rank 0's stdout is
    relayed (its JSON line stays the last line), the other ranks' output goes to stderr.
    Returns the exit code: 0 only when every rank returned 0; the first failure ends the others."""
    import threading
    with socket.socket() as sk:       # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it here
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr,
                                      text=True, bufsize=1, start_new_session=True))
    lines = []
    relay = threading.Thread(target=lambda: lines.extend(ln.rstrip("\n") for ln in procs[0].stdout), daemon=True)
    relay.start()
    code, live = 0, set(range(n))

    def stop(which, sig):             # exactly the process groups started above
        for o in which:
            try:
                os.killpg(procs[o].pid, sig)
            except OSError:
                pass

    def on_signal(signum, _frame):    # a harness timeout / Ctrl-C must not leave ranks behind holding the GPUs
        raise KeyboardInterrupt("signal %d" % signum)

    import signal
    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    try:
        while live:
            for r in sorted(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.discard(r)
                if rc != 0 and code == 0:
                    code = rc if rc > 0 else 1
                    sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (r, rc))
                    stop(live, signal.SIGTERM)
            time.sleep(0.05)
    except KeyboardInterrupt as why:
        sys.stderr.write("bench.py: interrupted (%s); stopping the ranks\n" % why)
        code = code or 130
    finally:
        still = [r for r in range(n) if procs[r].poll() is None]
        if still:                     # SIGTERM, a grace period, then SIGKILL
            stop(still, signal.SIGTERM)
            t_end = time.time() + 10
            while time.time() < t_end and any(procs[r].poll() is None for r in still):
                time.sleep(0.1)
            stop([r for r in still if procs[r].poll() is None], signal.SIGKILL)
        for sg, h in old.items():
            signal.signal(sg, h)
    relay.join(timeout=10)
    for ln in lines:
        print(ln)
    sys.stdout.flush()
    return code


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _probes = sorted(k for k in os.environ if k.startswith("SIG3D_PROBE_"))
    if _probes:
        sys.exit("bench.py: refusing to run with %s set (probe switches belong to tools/ab_step.py)" % ", ".join(_probes))
    _n = parse_args().gpus
    if _n > 1:
        sys.exit(spawn_ranks(_n, sys.argv[1:]))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from situation3d_amd import _lib, gemm_tuning  # noqa: E402
from situation3d_amd.ddp import GradBucketReducer, init_distributed  # noqa: E402
from situation3d_amd.graph_step import GraphedTrainStep  # noqa: E402
from situation3d_amd.model import SIG3DQFormer  # noqa: E402
from situation3d_amd.trainer import build_optimizer, get_loss, train_step  # noqa: E402

N_POINTS, BATCH, N_QUERY, N_TEXT, NUM_ANSWERS = 40000, 8, 32, 20, 706
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_F32_PEAK_TFLOPS = 157.3   # same guide: peak FP32 (vector)
MFMA_F32_PEAK_TFLOPS = 157.3   # same guide: f32-input MFMA runs at the f32 vector rate
MFMA_BF16_PEAK_TFLOPS = 2500.0 # same guide: dense bf16 matrix peak
LDS_PEAK_GBS = 150000.0        # same guide: ~150 TB/s aggregate for ds_read_b64 / b128 with every CU streaming
SA_LEVELS = [(40000, 2048, 64, 3), (2048, 1024, 32, 128), (1024, 512, 16, 256), (512, 256, 16, 256)]


def synthetic_batch(batch, n_points, seed, device, surface=False):
    """xyz ~ U([0,8]x[0,8]x[0,3]) m, colours U(0,1), pose = U(room) +
    z-rotation quaternion, 20 question token ids U(1000,30000), soft multi-hot answers.
    surface=True: the points lie on room surfaces instead (surface_points)."""
    g = torch.Generator().manual_seed(seed)
    if surface:
        xyz = surface_points(batch, n_points, g)
    else:
        xyz = torch.rand(batch, n_points, 3, generator=g) * torch.tensor([8.0, 8.0, 3.0])
    rgb = torch.rand(batch, n_points, 3, generator=g)
    t = torch.rand(batch, 3, generator=g) * torch.tensor([8.0, 8.0, 3.0])
    ang = (torch.rand(batch, generator=g) * 2 - 1) * 3.14159265
    quat = torch.stack([torch.zeros(batch), torch.zeros(batch), torch.sin(ang / 2), torch.cos(ang / 2)], 1)
    ids = torch.randint(1000, 30000, (batch, N_TEXT), generator=g)
    ans = torch.zeros(batch, NUM_ANSWERS)
    ans[torch.arange(batch), torch.randint(0, NUM_ANSWERS, (batch,), generator=g)] = 1.0
    d = {
        "point_clouds": torch.cat([xyz, rgb], -1),
        "auxiliary_task": torch.cat([t, quat], 1),
        "q_feat": {"input_ids": ids, "attention_mask": torch.ones(batch, N_TEXT, dtype=torch.long)},
        "answer_cat_scores": ans,
    }
    return to_device(d, device)


def surface_points(batch, n_points, g):
    """ScanNet-shaped SURFACE scenes: points on the floor, the four walls and the faces of a dozen furniture
    boxes of an 8 x 8 x 3 m room (+ 5 mm sensor noise), area-proportional -- real scans are surfaces, so
    neighbour lists are fuller than in the volume-uniform distribution SURVEY.md 8d prescribes for the
    headline (where 89-96 % of the SA1 / SA2 list entries are padding)."""
    out = torch.empty(batch, n_points, 3)
    for b in range(batch):
        faces = [((0.0, 0.0, 0.0), (8.0, 0.0, 0.0), (0.0, 8.0, 0.0))]                      # floor: origin, edge u, edge v
        for o, u in (((0, 0, 0), (8, 0, 0)), ((0, 8, 0), (8, 0, 0)), ((0, 0, 0), (0, 8, 0)), ((8, 0, 0), (0, 8, 0))):
            faces.append((tuple(map(float, o)), tuple(map(float, u)), (0.0, 0.0, 3.0)))    # walls
        nbox = 12
        lo = torch.rand(nbox, 2, generator=g) * 6.5 + 0.25
        sz = torch.rand(nbox, 3, generator=g) * torch.tensor([1.6, 1.6, 1.4]) + 0.3
        for i in range(nbox):
            x0, y0 = lo[i].tolist()
            sx, sy, sz_ = sz[i].tolist()
            faces.append(((x0, y0, sz_), (sx, 0.0, 0.0), (0.0, sy, 0.0)))                   # top
            faces.append(((x0, y0, 0.0), (sx, 0.0, 0.0), (0.0, 0.0, sz_)))                  # four sides
            faces.append(((x0, y0 + sy, 0.0), (sx, 0.0, 0.0), (0.0, 0.0, sz_)))
            faces.append(((x0, y0, 0.0), (0.0, sy, 0.0), (0.0, 0.0, sz_)))
            faces.append(((x0 + sx, y0, 0.0), (0.0, sy, 0.0), (0.0, 0.0, sz_)))
        o = torch.tensor([f[0] for f in faces])
        u = torch.tensor([f[1] for f in faces])
        v = torch.tensor([f[2] for f in faces])
        area = torch.linalg.cross(u, v).norm(dim=1)
        which = torch.multinomial(area / area.sum(), n_points, replacement=True, generator=g)
        st = torch.rand(n_points, 2, generator=g)
        pts = o[which] + st[:, :1] * u[which] + st[:, 1:] * v[which]
        out[b] = pts + torch.randn(n_points, 3, generator=g) * 0.005
    return out


def to_device(d, device):
    out = {}
    for k, v in d.items():
        out[k] = to_device(v, device) if isinstance(v, dict) else v.to(device).contiguous()
    return out


CPU_SCENES = 8   # the whole B = 8 batch (BatchNorm statistics are batch-wide: the metric's own configuration)


def group_algorithmic_bytes(b, n, m, ns, c):
    """group_points(C) = B*(4CN + 4*M*ns + 4C*M*ns); the fused kernel moves the xyz
    group (C=3) and the feature group (C=c) in one launch and reads idx once."""
    return b * (4 * 3 * n + 4 * c * n + 4 * m * ns + 4 * (3 + c) * m * ns)


def oracle_forward(cpu_model, batch, backward=False, ext=None):
    """The composed path of `batch` on the HOST through the oracle: the nine native ops from
    oracle/pointnet2_oracle.c (bound as `pointnet2._ext` under this build's module stack), SharedMLP on
    torch CPU, the Q-Former through oracle/qformer_ref.py (eval-mode dropout) -> data_dict with the model's
    output keys and the loss.  Checker code only -- never on the product path.
    ext: another binding of the nine ops + pose_to_matrix (tests: the float64 adjudication run keeps the oracle's
    indices and moves data in float64)."""
    from oracle import pointnet2_ref, qformer_ref
    from situation3d_amd.pointnet2 import pointnet2_utils
    saved_ext = pointnet2_utils._ext
    if ext is not None:
        pointnet2_ref = ext
    pointnet2_utils._ext = pointnet2_ref  # CPU restatement of the nine ops
    try:
        n = batch["point_clouds"].shape[0]
        pc = batch["point_clouds"]
        xyz = pc[..., :3].contiguous()
        feats = pc[..., 3:].transpose(1, 2).contiguous()
        tok_xyz, tok_feat = cpu_model.encoder(xyz, feats)
        tok_feat = tok_feat.transpose(1, 2)
        pose = batch["auxiliary_task"]
        M = pointnet2_ref.pose_to_matrix(pose)
        sit = torch.einsum("bcr,bnc->bnr", M[:, :3, :3], tok_xyz - pose[:, None, :3])
        tokens = tok_feat + cpu_model.pos_embed(sit if getattr(cpu_model, "pos_embed_dim", 3) == 3
                                                else tok_xyz[..., :2])
        sd = dict(cpu_model.Qformer.bert.state_dict())
        sd.update({k: v for k, v in cpu_model.Qformer.bert.named_parameters()})
        c = cpu_model.Qformer.config
        cfg = dict(num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
                   layer_norm_eps=c.layer_norm_eps, add_cross_attention=True,
                   cross_attention_freq=c.cross_attention_freq)
        q = batch["q_feat"]
        nq = cpu_model.query_tokens.shape[1]
        att = torch.cat([torch.ones(n, nq, dtype=torch.long), q["attention_mask"]], 1)
        hidden = qformer_ref.bert_model(sd, cfg, query_embeds=cpu_model.query_tokens.expand(n, -1, -1),
                                        input_ids=q["input_ids"], attention_mask=att,
                                        encoder_hidden_states=tokens)
        fused = hidden[:, :nq]
        pooled = fused.mean(1)
        dd = dict(batch)
        dd["scene_positions"], dd["att_feat_pre"], dd["att_feat_ori"] = tok_xyz, tok_feat, fused
        dd["aux_scores"] = cpu_model.aux_reg(pooled)
        dd["answer_scores"] = cpu_model.answer_cls(pooled)
        get_loss(dd)
        if backward:   # the grouping / gather backward ops dispatch through `_ext` too: still bound to the oracle
            dd["loss"].backward()
    finally:
        pointnet2_utils._ext = saved_ext
    return dd


def _without_dropout(model):
    """The oracle's Q-Former is eval-mode (dropout = identity) while BatchNorm keeps batch statistics:
    the comparable product run is train mode with every dropout probability at zero."""
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return model


def cpu_baseline(model, seed):
    """The same step on the host cores through the ORACLE (kind "port"): CPU_SCENES scenes of the
    B=8 workload as one batch, forward + backward (~10 s) -- and, for free, an end-to-end parity check at
    full size: the same scenes with the same (trained) weights through the HIP path give `loss_gpu`."""
    from oracle import pointnet2_ref
    import copy
    model.Qformer.bert.encoder._arena = None   # per-forward scratch of the last step (non-leaf tensors): not part of the model
    cpu_model = _without_dropout(copy.deepcopy(model).cpu().train())
    batch = synthetic_batch(CPU_SCENES, N_POINTS, seed, "cpu")
    t0 = time.perf_counter()
    dd = oracle_forward(cpu_model, batch, backward=True)
    dt = time.perf_counter() - t0
    dev = next(model.parameters()).device
    gpu_model = _without_dropout(copy.deepcopy(model).train())
    with torch.no_grad():
        out = gpu_model(to_device(batch, dev))
        loss_gpu, _ = get_loss(out)
    loss_cpu, loss_gpu = float(dd["loss"].detach()), float(loss_gpu)
    scores_diff = float((out["answer_scores"].cpu() - dd["answer_scores"].detach()).abs().max())
    threads = max(pointnet2_ref.num_threads(), torch.get_num_threads())
    return {"value": round(CPU_SCENES / dt, 4), "unit": "samples/s", "cores": threads,
            "kind": "port",
            "sample": "%d scenes (B=%d of the B=8 step), 40k pts, fwd+bwd, oracle C ops (OpenMP, %d threads) + "
                      "torch CPU fp32 MLP/Q-Former (%d threads), %.1f s of wall time"
                      % (CPU_SCENES, CPU_SCENES, pointnet2_ref.num_threads(), torch.get_num_threads(), dt),
            # full-size end-to-end parity of the timed model (trained weights, dropout off, same scenes)
            "loss_cpu": round(loss_cpu, 5), "loss_gpu": round(loss_gpu, 5),
            "loss_rel_diff": float("%.3g" % (abs(loss_cpu - loss_gpu) / max(abs(loss_cpu), 1e-12))),
            "answer_scores_max_abs_diff": float("%.3g" % scores_diff)}


# The sources that decide which launches the ball-query / grouping pair consists of and what they move: its two kernel
# files, the distinct-neighbour lists the compact grouping reads, the helpers both include, the set-abstraction modules'
# Python (which levels group at all) and the geometry plan.  (Until late round 6 every file under csrc/ counted: a comment in
# a GEMM header marked the pair's counters stale.)
PAIR_SOURCES = ("situation3d_amd/csrc/ball_query.hip", "situation3d_amd/csrc/group_points.hip",
                "situation3d_amd/csrc/compact.hip", "situation3d_amd/csrc/sig3d_common.h",
                "situation3d_amd/pointnet2/", "situation3d_amd/geometry.py")


def pair_traffic_is_stale(source_hashes, root=None):
    """source_hashes: {path: sha256[:16]} stored with the PMC counters (tools/pmc_traffic.py).  True when one of the pair's
    sources is missing or differs from the tree the counters were read in (by content: there is no git on the GPU box)."""
    import hashlib
    root = ROOT if root is None else root
    mine = {f: h for f, h in source_hashes.items() if f.startswith(PAIR_SOURCES)}
    if not mine:
        return True
    return any(not os.path.exists(os.path.join(root, f)) or
               hashlib.sha256(open(os.path.join(root, f), "rb").read()).hexdigest()[:16] != h for f, h in mine.items())


TIMED_ENTRY_POINTS = ["sig3d_query_group_fused", "sig3d_query_group_fused_pm", "sig3d_query_group_compact",
                      "sig3d_transpose_cn", "sig3d_adamw_table", "sig3d_adamw_flat", "sig3d_ball_query",
                      "sig3d_ball_query_grid", "sig3d_ball_query_levels", "sig3d_ball_query_levels_ex",
                      "sig3d_furthest_point_sampling", "sig3d_furthest_point_sampling_blocks",
                      "sig3d_sa_first_layer_fwd", "sig3d_sa_first_layer_dw"]
KSTEPS = 3   # eager steps bracketed with HIP events after the timed region


def measure(args, rank, world, device, steps, warmup, surface=False, compact=True, kernels=True):
    """Build model + optimizer + (graphed) step for one data / mode variant, time `steps` steps between
    barriers + synchronize, optionally bracket the hand-written kernels of KSTEPS eager steps with HIP events.
    -> dict(dt, final_loss, recs, model, compact_info, use_graph, n_act)."""
    from situation3d_amd.pointnet2 import fused_mlp
    saved_compact = fused_mlp.COMPACT
    fused_mlp.COMPACT = bool(compact) and saved_compact
    try:
        torch.manual_seed(1234)  # identical initial weights on every rank (and in every variant)
        model = SIG3DQFormer(num_answers=NUM_ANSWERS).to(device).train()
        # clip_grad_value_(1.0) + AdamW (lr 2e-5, wd 0.05: scripts/train.sh:7) + zero_grad fused over
        # flat storage; under data parallelism the same flat gradient buffers are all-reduced in place
        data_parallel = world > 1 or args.force_reducer
        optimizer = build_optimizer(model, name="adamw" if args.torch_adamw else "flat_adamw",
                                    qf_cut=args.qf_cut if data_parallel and not args.torch_adamw else None)
        reducer = None
        if data_parallel:
            reducer = (GradBucketReducer(model.parameters()) if args.torch_adamw
                       else GradBucketReducer.from_flat(optimizer.flat_grad_buffers()))
        n_batches = min(4, steps + warmup)
        batches = [synthetic_batch(BATCH, N_POINTS, 1234 + 1000 * rank + i, device, surface=surface)
                   for i in range(n_batches)]
        # every forward/backward of this process runs on ONE non-default stream (see graph_step.py)
        work = torch.cuda.Stream(device)
        recs, graphed = None, None
        with torch.cuda.stream(work):
            use_graph = not args.no_graph
            if use_graph:
                graphed = GraphedTrainStep(model, optimizer, batches[0], prefetch_geometry=not args.no_prefetch,
                                           reducer=reducer, prefetch_depth=args.geo_depth)
                depth = graphed.prefetch_depth

                def step(i):
                    return graphed(batches[i % n_batches], upcoming=[batches[(i + 1 + k) % n_batches] for k in range(depth)])
            else:
                def step(i):
                    return train_step(model, optimizer, dict(batches[i % n_batches]), reducer=reducer)

            for i in range(warmup):
                step(i)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                loss = step(warmup + i)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            final_loss = float(loss.item())
            if graphed is not None and graphed.handshake_timed_out():
                # a geometry chain gave up waiting for its ticket (20 minutes: geometry.py) and ran on coordinates that may not have been
                # staged: the numbers of this run would be those of a broken pipeline
                raise RuntimeError("a geometry chain timed out waiting for its start ticket (sig3d_ticket_wait)")
            comm, bq_tests = None, None
            if reducer is not None:
                # the gradient exchange of a few more steps, bracketed by events on the compute stream (ddp.CommStats):
                # bytes and collectives per step, time spent waiting for them, share of their window covered by compute
                from situation3d_amd import ddp
                ddp.COMM_STATS = ddp.CommStats()
                try:
                    for i in range(3):
                        ddp.COMM_STATS.begin_step()
                        step(warmup + steps + i)
                    comm = ddp.COMM_STATS.summary()
                finally:
                    ddp.COMM_STATS = None
                if world > 1:   # the slowest rank's exposure is the step's
                    t = torch.tensor([comm["exposed_ms"], comm["comm_window_ms"]], dtype=torch.float64, device=device)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    comm["exposed_ms"], comm["comm_window_ms"] = round(float(t[0]), 4), round(float(t[1]), 4)
                    comm["overlap_frac"] = round(1.0 - float(t[0]) / float(t[1]), 4) if float(t[1]) > 0 else None
            if kernels:
                # Per-launch durations of the hand-written kernels: HIP events on the launch stream around
                # every call.  Under hipGraph replay there are no per-kernel events, so the SAME launches are
                # issued once more eagerly right after the timed region and bracketed there.
                _lib.enable_timing(TIMED_ENTRY_POINTS)
                if reducer is not None:
                    reducer.hooks_enabled = True
                # the same launches as the replayed graph: geometry through a GeometryPlan (all ball queries of the
                # stack in one launch pair), then the step over that plan
                from situation3d_amd.geometry import GeometryPlan
                eplan = GeometryPlan(BATCH, N_POINTS, model.encoder.LEVELS, device)
                for i in range(KSTEPS):
                    bt = dict(batches[i % n_batches])
                    bt["geometry_plan"] = eplan.compute(bt["point_clouds"][..., :3].contiguous())
                    train_step(model, optimizer, bt, reducer=reducer)
                torch.cuda.synchronize()
                recs = _lib.timing_records()
                _lib.enable_timing(None)
                # the neighbour search's own work: the distance tests of one more call over the last plan
                # (sig3d_ball_query_levels_stats; the kernel is bound by LDS round trips, not by HBM)
                bq_tests = None
                if getattr(eplan, "_bq_levels", None) is not None:
                    words = torch.zeros(1, dtype=torch.int64, device=device)
                    _lib.call("sig3d_ball_query_levels_stats", BATCH, len(eplan._bq_levels), eplan._bq_levels,
                              _lib.ptr(eplan._bq_work), eplan._bq_work.numel(), 0, _lib.ptr(words), _lib.stream_ptr(device))
                    bq_tests = int(words.item())
        # set-abstraction levels that ran over the distinct neighbours only, with the fraction of their
        # (centre, sample) positions that are distinct on the last batch
        compact_info, distinct = {}, {}
        plan = getattr(graphed, "plan_cur", None)
        if plan is not None:
            for li, cl in enumerate(plan.compact):
                if cl is not None:
                    bsz, mpt, nsm = cl.shape
                    frac = float(cl.n_act.float().mean().item()) / (mpt * nsm)
                    distinct["SA%d" % (li + 1)] = round(frac, 4)
                    layer = getattr(model.encoder, "sa%d" % (li + 1))
                    if getattr(layer, "_compact_decision", False):
                        compact_info["SA%d" % (li + 1)] = round(frac, 4)
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return dict(dt=float(t.item()), final_loss=final_loss, recs=recs, model=model, compact_info=compact_info,
                    distinct=distinct, use_graph=use_graph, comm=comm, bq_tests=bq_tests)
    finally:
        fused_mlp.COMPACT = saved_compact


def ball_query_algorithmic_bytes(b, n, m, ns):
    """B*(12N + 12M + 4*M*ns)"""
    return b * (12 * n + 12 * m + 4 * m * ns)


def pair_roofline(recs, distinct):
    """The pair the north star names -- ball_query + group_points -- over ALL FOUR levels of a step, from the
    event-bracketed eager launches: achieved = algorithmic bytes / summed duration.  Dense grouping launches
    count SURVEY.md 8d's bytes; a compact launch (distinct neighbours only) counts its OWN algorithmic bytes:
    the sources it gathers from once, its index list and the columns it writes (distinct fraction x dense)."""
    t_ms, nbytes, launches = 0.0, 0, 0
    parts = {}

    def add(name, ms, nb):
        nonlocal t_ms, nbytes, launches
        t_ms, nbytes, launches = t_ms + ms, nbytes + nb, launches + 1
        p = parts.setdefault(name, {"launches": 0, "ms": 0.0, "algorithmic_bytes": 0})
        p["launches"] += 1
        p["ms"] += ms
        p["algorithmic_bytes"] += nb

    for name in ("sig3d_ball_query", "sig3d_ball_query_grid"):
        for s_ev, e_ev, ints in recs[name]:
            add(name, s_ev.elapsed_time(e_ev), ball_query_algorithmic_bytes(ints[0], ints[1], ints[2], ints[3]))
    for s_ev, e_ev, ints in recs["sig3d_ball_query_levels"] + recs["sig3d_ball_query_levels_ex"]:   # SA1-4 in one launch pair
        add("sig3d_ball_query_levels", s_ev.elapsed_time(e_ev),
            sum(ball_query_algorithmic_bytes(ints[0], n, m, ns) for n, m, ns, _ in SA_LEVELS))
    for name, pm in (("sig3d_query_group_fused", False), ("sig3d_query_group_fused_pm", True)):
        for s_ev, e_ev, ints in recs[name]:
            add(name, s_ev.elapsed_time(e_ev),
                group_algorithmic_bytes(ints[0], ints[1], ints[2], ints[5] if pm else ints[4], ints[3]))
    for s_ev, e_ev, ints in recs["sig3d_query_group_compact"]:
        bb, nn, mm, cc, ns_ = ints[0], ints[1], ints[2], ints[3], ints[5]
        level = [k for k, (ln, lm, lns, lc) in zip(("SA1", "SA2", "SA3", "SA4"), SA_LEVELS) if (ln, lm, lns) == (nn, mm, ns_)]
        frac = distinct.get(level[0], 1.0) if level else 1.0
        add("sig3d_query_group_compact", s_ev.elapsed_time(e_ev),
            int(bb * (12 * nn + 4 * cc * nn) + frac * bb * mm * ns_ * (4 + 4 * (3 + cc))))
    for s_ev, e_ev, ints in recs["sig3d_transpose_cn"]:   # point-major copies feeding the wide levels: pure overhead
        add("sig3d_transpose_cn", s_ev.elapsed_time(e_ev), 0)
    return t_ms, nbytes, launches, parts


def ops_roofline(device, seed=1234):
    """The pair as the reference's API exposes it -- ball_query + QueryAndGroup's grouping (pointnet2_utils.py:
    260-376) -- DENSE, at the four SA shapes of BASELINE config 3 (B = 8; SURVEY.md 8d: 314.8 MB algorithmic per
    pass), through the C ABI, every launch bracketed by HIP events on the launch stream.  The training step itself
    has fused most of this away (compact lists at SA1 / SA2, the first SharedMLP layer gathering on load), so its
    own pair (`roofline`) is a handful of latency-sized launches; this is the bandwidth the op kernels reach when
    they are used the way the reference uses them."""
    import ctypes
    from situation3d_amd.pointnet2 import _ext
    pc = synthetic_batch(BATCH, N_POINTS, seed, device)["point_clouds"]
    cur = pc[..., :3].contiguous()
    from situation3d_amd.model import PointNet2Encoder
    radii = [lv[1] for lv in PointNet2Encoder.LEVELS]
    probs, feats = [], []
    g = torch.Generator(device="cpu").manual_seed(seed)
    for (n, m, ns, c), radius in zip(SA_LEVELS, radii):
        inds = _ext.furthest_point_sampling(cur, m)
        nxt = torch.gather(cur, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        probs.append((cur, nxt, radius, ns, torch.empty(BATCH, m, ns, dtype=torch.int32, device=device)))
        f = torch.rand(BATCH, c, n, generator=g).to(device)
        # wide levels read the point-major twin their producer writes (sig3d_bn_relu_maxpool_pm)
        feats.append(f.transpose(1, 2).contiguous() if c >= 32 else f)
        cur = nxt
    arr = _lib.bq_levels(probs)
    work = torch.empty(max(_lib.bq_levels_workspace_bytes(BATCH, arr), 16), dtype=torch.uint8, device=device)
    outs = [torch.empty(BATCH, 3 + c, m, ns, device=device) for n, m, ns, c in SA_LEVELS]
    s = _lib.stream_ptr(device)
    garr = _lib.group_levels([(xyz, nxt, radius, idx, feats[li], c >= 32, outs[li])
                              for li, ((n, m, ns, c), (xyz, nxt, radius, _, idx)) in enumerate(zip(SA_LEVELS, probs))])
    reps, t_bq, t_gr, t_one = 5, [], [], []
    for it in range(reps + 2):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
        ev[0].record()
        _lib.call("sig3d_ball_query_levels", BATCH, len(arr), arr, _lib.ptr(work), work.numel(), s)
        ev[1].record()
        for li, ((n, m, ns, c), (xyz, nxt, radius, _, idx)) in enumerate(zip(SA_LEVELS, probs)):
            if c >= 32:
                _lib.call("sig3d_query_group_fused_pm", BATCH, n, m, c, c, ns, 1, 1, ctypes.c_float(radius), _lib.ptr(xyz),
                          _lib.ptr(nxt), _lib.ptr(feats[li]), _lib.ptr(idx), _lib.ptr(outs[li]), s)
            else:
                _lib.call("sig3d_query_group_fused", BATCH, n, m, c, ns, 1, 1, ctypes.c_float(radius), _lib.ptr(xyz),
                          _lib.ptr(nxt), _lib.ptr(feats[li]), _lib.ptr(idx), _lib.ptr(outs[li]), s)
            ev[2 + li].record()
        # the same four groupings as ONE launch (sig3d_query_group_levels: every list exists once the ball query has run)
        _lib.call("sig3d_query_group_levels", BATCH, len(garr), garr, s)
        ev[6].record()
        torch.cuda.synchronize()
        if it >= 2:
            t_bq.append(ev[0].elapsed_time(ev[1]))
            t_gr.append([ev[1 + li].elapsed_time(ev[2 + li]) for li in range(4)])
            t_one.append(ev[5].elapsed_time(ev[6]))
    bq_ms = sum(t_bq) / reps
    gr_ms = [sum(t[li] for t in t_gr) / reps for li in range(4)]
    bq_bytes = sum(ball_query_algorithmic_bytes(BATCH, n, m, ns) for n, m, ns, _ in SA_LEVELS)
    gr_bytes = [group_algorithmic_bytes(BATCH, n, m, ns, c) for n, m, ns, c in SA_LEVELS]
    one_ms = sum(t_one) / reps
    per_level_gbs = sum(gr_bytes) / (sum(gr_ms) * 1e-3) / 1e9
    total_ms, total_b = bq_ms + one_ms, bq_bytes + sum(gr_bytes)
    gbs = total_b / (total_ms * 1e-3) / 1e9
    grp_gbs = sum(gr_bytes) / (one_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "ball_query (SA1-4, one launch pair) + QueryAndGroup's grouping (SA1-4, one launch: "
                                      "sig3d_query_group_levels), dense lists, BASELINE config 3 shapes, through the C ABI",
            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "algorithmic_bytes": total_b, "ms": round(total_ms, 4),
            "ball_query": {"ms": round(bq_ms, 4), "algorithmic_bytes": bq_bytes},
            "group_points": {"ms": round(one_ms, 4), "algorithmic_bytes": gr_bytes,
                             "achieved": round(grp_gbs, 1), "frac": round(grp_gbs / HBM_PEAK_GBS, 4),
                             "per_level_launches": {"ms": [round(t, 4) for t in gr_ms], "achieved": round(per_level_gbs, 1),
                                                    "frac": round(per_level_gbs / HBM_PEAK_GBS, 4)}}}


def forward_only_variant(device, bsz=4, reps=20):
    from situation3d_amd.serve import PipelinedForward
    torch.manual_seed(1234)
    model = SIG3DQFormer(num_answers=NUM_ANSWERS).to(device).eval()
    bs = [synthetic_batch(bsz, N_POINTS, 5000 + i, device) for i in range(4)]
    work = torch.cuda.Stream(device)
    with torch.cuda.stream(work), torch.no_grad():
        for _ in range(3):
            model(dict(bs[0]))
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with gemm_tuning.no_tuning(), torch.cuda.graph(g, stream=work):
            model(dict(bs[0]))
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        single = (time.perf_counter() - t0) / reps
        pipe = PipelinedForward(model, bs[0], depth=2, high_priority=False)
        for i in range(6):
            pipe(bs[i % 4], [bs[(i + 1 + k) % 4] for k in range(2)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(6, 6 + 2 * reps):
            pipe(bs[i % 4], [bs[(i + 1 + k) % 4] for k in range(2)])
        torch.cuda.synchronize()
        piped = (time.perf_counter() - t0) / (2 * reps)
    del pipe, g, model
    torch.cuda.empty_cache()
    return {"single_batch_latency_ms": round(single * 1e3, 3), "single_batch_samples_per_s": round(bsz / single, 1),
            "pipelined_ms_per_batch": round(piped * 1e3, 3), "value": round(bsz / piped, 1), "unit": "samples/s",
            "batch": bsz, "note": "hipGraph replay of one forward; and two geometry chains in flight (serve.PipelinedForward)"}


def config5_variant(device, bsz=4, reps=5):
    """BASELINE config 5 at its single-GPU shape (3D-LLM BLIP-2 point branch, blip2_t5.py:102-129): B = 4 scenes of
    Nk point tokens of width 1408 -> position embedding -> Q-Former (32 queries, six cross-attention layers) -> t5_proj.
    The key / value projections of the six cross layers are 94 % of the FLOPs: they run on sig3d_gemmp (six
    bf16 matrix-core products per f32 product over operands split once) -- timed launch by launch with HIP events --
    and, for the comparison, on the library (SIG3D_QF_BIG_ROWS=0: `library_projections_ms`, with the solutions
    TunableOp picks for these shapes when tuning is on, as in a full bench.py run)."""
    from situation3d_amd import qformer as qf
    from situation3d_amd.blip2 import Blip2PointQFormer
    torch.manual_seed(55)
    model = Blip2PointQFormer().to(device).eval()
    out = {}
    big_rows = qf.BIG_ROWS

    def run(nk, backward, own):
        qf.BIG_ROWS = big_rows if own else 0
        g = torch.Generator(device="cpu").manual_seed(77)
        feat = torch.randn(bsz, nk, 1408, device=device)
        pc = torch.randint(0, 256, (bsz, nk, 3), generator=g).float().to(device)

        def once():
            if backward:
                f = feat.detach().requires_grad_(True)
                model.zero_grad(set_to_none=True)
                model({"pc_feat": f, "pc": pc})["inputs_t5"].sum().backward()
            else:
                with torch.no_grad():
                    model({"pc_feat": feat, "pc": pc})
        n_rep = reps if nk >= 40000 else 4 * reps
        for _ in range(3):          # (TunableOp tunes a library shape at its first call: inside the warm-up)
            once()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_rep):
            once()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_rep
        proj_ms = None
        if own:     # the projections' launches, bracketed by events on their stream
            _lib.enable_timing(["sig3d_gemmp"])
            once()
            torch.cuda.synchronize()
            recs = _lib.timing_records()["sig3d_gemmp"]
            _lib.enable_timing(None)
            proj_ms = [s_.elapsed_time(e_) for s_, e_, _ in recs]
        del feat, pc
        torch.cuda.empty_cache()
        return dt, proj_ms

    for key, nk, backward in (("forward, Nk=80000", 80000, False), ("forward + backward, Nk=5000", 5000, True)):
        dt, proj_ms = run(nk, backward, True)
        dt_lib, _ = run(nk, backward, False)
        flops = 2.0 * bsz * nk * 1408 * 1536 * 6 * (3 if backward else 1)
        ach = flops / (sum(proj_ms) * 1e-3) / 1e12
        out[key] = {
            "ms": round(dt * 1e3, 3), "value": round(bsz / dt, 2), "unit": "samples/s", "batch": bsz,
            "library_projections_ms": round(dt_lib * 1e3, 3),
            "roofline_kv_projection": {
                # the kernel issues bf16 MFMAs -- six per f32-equivalent product -- so its roofline is the dense bf16
                # matrix peak over the ISSUED work; the f32-equivalent rate is reported beside it, without a fraction
                "bound": "mfma", "launches": len(proj_ms), "ms": round(sum(proj_ms), 3), "flops": flops,
                "issued_flops": 6 * flops,
                "achieved": round(6 * ach, 1), "unit": "TFLOP/s", "peak": MFMA_BF16_PEAK_TFLOPS,
                "frac": round(6 * ach / MFMA_BF16_PEAK_TFLOPS, 4),
                "f32_equivalent_tflops": round(ach, 1),
                "dtype": "bf16 MFMA: f32 results from six bf16 products per f32 product (three-term split, f32 accumulation)"}}
    qf.BIG_ROWS = big_rows
    del model
    torch.cuda.empty_cache()
    return out


def comm_model(world, bytes_per_step, single_gpu_ms=7.5, qf_cut=0):
    """DESIGN.md section 6, stated so that a measured line falsifies or confirms it: the gradient exchange of one step
    over xGMI (point-to-point links, 76.8 GB/s per direction each; every GPU has a direct link to each of its N - 1
    peers), a ring / mesh all-reduce moving 2 (N - 1) / N x S bytes per rank over those links; + 0.3 ms of RCCL kernels
    sharing the CUs.  Two forms of the step (graph_step.GraphedTrainStep):
      uncut: every Q-Former gradient is ready when the Q-Former's backward pass and its layer-batched weight gradients
             are done (70 % of a single-GPU step); the wire runs under the encoder's backward pass and the bucket-by-bucket update;
      cut k: the layers from k on (half the bytes at k = 6) are ready at ~52 % of the step, the rest at 70 %.
    `forms` holds both at N = 2 / 4 / 8 (predicted ms per step and the part of the exchange that is NOT hidden), so that
    one SCALE run says which form to keep; N = 8 also with RCCL's measured ~180 GB/s of algorithm bandwidth on this
    class of node instead of the 7-link ideal."""
    link = 76.8e9

    def one(n, cut, rccl_180=False):
        wire = bytes_per_step / 180e9 * 1e3 if rccl_180 else 2.0 * (n - 1) / n * bytes_per_step / ((n - 1) * link) * 1e3
        floor = single_gpu_ms + 0.3                        # compute chain + RCCL's kernels on the CUs
        if cut:
            upper_done = max(0.52 * single_gpu_ms + wire / 2, 0.7 * single_gpu_ms)
            end = upper_done + wire / 2 + 0.1
        else:
            end = 0.7 * single_gpu_ms + wire + 0.1
        return {"wire_ms": round(wire, 3), "predicted_ms_per_step": round(max(floor, end), 3),
                "exposed_ms": round(max(0.0, end - floor), 3)}

    forms = {}
    for n in (2, 4, 8):
        forms["N=%d" % n] = {"uncut": one(n, False), "cut": one(n, True)}
    forms["N=8 at rccl 180 GB/s"] = {"uncut": one(8, False, True), "cut": one(8, True, True)}
    if world <= 1:
        return {"wire_ms": 0.0, "predicted_ms_per_step": None, "note": "no wire at world size 1", "forms": forms}
    mine = one(world, bool(qf_cut))
    out = {"links_per_gpu": world - 1, "link_gbs_per_direction": 76.8, "wire_ms": mine["wire_ms"],
           "assumed_single_gpu_ms": single_gpu_ms, "form": "cut %d" % qf_cut if qf_cut else "uncut",
           "predicted_ms_per_step": mine["predicted_ms_per_step"], "predicted_exposed_ms": mine["exposed_ms"],
           "forms": forms}
    if world == 8:
        r = one(8, bool(qf_cut), True)
        out["wire_ms_at_rccl_180_gbs"] = r["wire_ms"]
        out["predicted_ms_per_step_at_rccl_180_gbs"] = r["predicted_ms_per_step"]
    return out


def refuse_probes():
    """Measurement switches (tools/probes/geo_probes.py) skip or replace work inside a captured step; the product reads
    none of them any more, but a bench line produced with one in the environment would invite the question."""
    probes = sorted(k for k in os.environ if k.startswith("SIG3D_PROBE_"))
    if probes:
        raise SystemExit("bench.py: refusing to run with %s set (probe switches belong to tools/ab_step.py)" % ", ".join(probes))


def main():
    args = parse_args()
    refuse_probes()

    rank, local, world = init_distributed()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d in the environment (an external launcher must "
                         "start exactly --gpus ranks; without WORLD_SIZE bench.py starts them itself)"
                         % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs an MI355X; the HIP path has no CPU fallback"
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    if not args.no_gemm_tuning:
        gemm_tuning.enable(tune_missing=True)  # committed winners; unseen shapes tuned in warm-up

    # ranks that really take part in the collectives: an all-reduce of ones over the process group
    ranks_seen, backend = 1, None
    if dist.is_initialized():
        one = torch.ones(1, device=device)
        dist.all_reduce(one)
        ranks_seen, backend = int(one.item()), dist.get_backend()
        assert ranks_seen == dist.get_world_size() == world

    # ---- headline: SURVEY.md 8d distribution (volume-uniform points), compact set abstraction where it pays
    head = measure(args, rank, world, device, args.steps, args.warmup, surface=args.surface)
    dt, recs, model = head["dt"], head["recs"], head["model"]

    if rank == 0:
        def kernel_ms(name):
            return [s.elapsed_time(e) for s, e, _ in recs[name]]

        # dense grouping launches alone (the kernel round 1 reported)
        grp, grp_bytes = [], 0
        for name, pm in (("sig3d_query_group_fused", False), ("sig3d_query_group_fused_pm", True)):
            for s_ev, e_ev, ints in recs[name]:
                grp.append(s_ev.elapsed_time(e_ev))
                grp_bytes += group_algorithmic_bytes(ints[0], ints[1], ints[2], ints[5] if pm else ints[4], ints[3])
        cgrp = kernel_ms("sig3d_query_group_compact")
        pair_ms, pair_bytes, pair_launches, pair_parts = pair_roofline(recs, head["distinct"])
        pair_gbs = pair_bytes / (pair_ms * 1e-3) / 1e9 if pair_ms else 0.0
        # the largest HBM-bound kernel of the step by time is the flat AdamW update: per parameter it reads
        # p, g, m, v and writes p, m, v (28 B: clip + update in one pass; the gradients are dropped, not zeroed)
        adam = kernel_ms("sig3d_adamw_table") + kernel_ms("sig3d_adamw_flat")
        n_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
        adam_gbs = 28.0 * n_params * KSTEPS / (sum(adam) * 1e-3) / 1e9 if adam else 0.0
        tr = kernel_ms("sig3d_transpose_cn")
        achieved = grp_bytes / (sum(grp) * 1e-3) / 1e9 if grp else 0.0
        bq = kernel_ms("sig3d_ball_query") + kernel_ms("sig3d_ball_query_grid") + kernel_ms("sig3d_ball_query_levels") \
            + kernel_ms("sig3d_ball_query_levels_ex")
        fps = kernel_ms("sig3d_furthest_point_sampling") + kernel_ms("sig3d_furthest_point_sampling_blocks")
        # HBM traffic cannot be read from inside this process: it comes from the rocprofv3 --pmc passes of the
        # commit named in the file (FETCH_SIZE doubled per MI355X_MICROARCH.md, + WRITE_SIZE; tools/pmc_traffic.py)
        traffic, traffic_commit, traffic_stale = None, None, None
        pmcs = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_group_pair.json"))
        if pmcs:
            j = json.load(open(os.path.join(ROOT, "profiles", pmcs[-1])))
            traffic, traffic_commit = round(j["traffic_bytes_per_step"]), j.get("commit")
            # stale = a source that decides the pair's launches or their traffic (kernels, the SA modules' Python, the
            # geometry plan) differs from the tree the counters were read in; by content hash (no git on the GPU box),
            # null for files of rounds that stored no hashes
            if j.get("source_hashes"):
                traffic_stale = pair_traffic_is_stale(j["source_hashes"])
        out = {
            "metric": "QA samples/sec fwd+bwd (SQA3D, 40k pts, B=8)",
            "value": round(world * BATCH * args.steps / dt, 3),
            "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            # ranks counted by an all-reduce of ones ("nccl" is RCCL on ROCm); 1 / None without a process group
            "rccl_ranks": ranks_seen, "dist_backend": backend,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic-surface (NOT the headline distribution)" if args.surface else "synthetic",
            "config": {"workload": "SQA3D train step fwd+bwd+AdamW, 40k pts/scene, B=8/GPU, "
                                   "SA1-4 -> 256 tokens -> situational re-encode -> Q-Former "
                                   "(32 queries + 20 question tokens, 12 layers)",
                       "global_batch": world * BATCH, "points_per_scene": N_POINTS,
                       "parallelism": "dp%d" % world,
                       "geometry_chains_in_flight": 0 if args.no_prefetch or args.no_graph else args.geo_depth},
            # the pair the north star names, all four levels of a step, compact launches at their own bytes;
            # dense-equivalent = SURVEY.md 8d's 314.8 MB per step over the same time
            "roofline": {"bound": "hbm", "kernel": "ball_query (all levels, cell-binned centres) + query_group (fused / point-major / "
                                                   "compact), SA1-4",
                         "achieved": round(pair_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(pair_gbs / HBM_PEAK_GBS, 4),
                         # HBM bytes of the same kernels per step (PMC, FETCH_SIZE x 2 + WRITE_SIZE) and the commit
                         # they were measured at; per step like algorithmic_bytes_per_step
                         "traffic": traffic, "traffic_commit": traffic_commit, "traffic_stale": traffic_stale,
                         "algorithmic_bytes_per_step": round(pair_bytes / KSTEPS),
                         "launches_per_step": pair_launches // KSTEPS, "ms_per_step": round(pair_ms / KSTEPS, 4),
                         # per C-ABI entry point and step: launches, event-bracketed ms, algorithmic bytes
                         "parts": {k: {"launches": v["launches"] // KSTEPS, "ms": round(v["ms"] / KSTEPS, 4),
                                       "algorithmic_bytes": v["algorithmic_bytes"] // KSTEPS}
                                   for k, v in pair_parts.items()},
                         # what ANY implementation of this launch list would take at least: 1.5 us per kernel
                         # boundary + the algorithmic bytes at the 6.3 TB/s this part's float4 copy reaches
                         # (MI355X_MICROARCH.md); vs_floor = floor / measured
                         "floor_model": {"launches": pair_launches // KSTEPS, "us_per_launch": 1.5, "stream_gbs": 6300.0,
                                         "floor_ms": round((pair_launches // KSTEPS) * 1.5e-3 + pair_bytes / KSTEPS / 6.3e12 * 1e3, 4),
                                         "vs_floor": round(((pair_launches // KSTEPS) * 1.5e-3 + pair_bytes / KSTEPS / 6.3e12 * 1e3)
                                                           / (pair_ms / KSTEPS), 4) if pair_ms else None}},
            # the neighbour search against the bound it really has: a distance test is 8 flop (3 sub, 3 mul, 2 add, each
            # individually rounded) on a 16-byte centre record read from LDS; peaks from MI355X_MICROARCH.md
            # (157.3 TFLOP/s f32 vector, ~150 TB/s of ds_read_b128 with every CU streaming)
            "roofline_ball_query": None if not (head["bq_tests"] and bq) else {
                "bound": "valu/lds", "kernel": "bqc_scatter_kernel + bqc_rank_kernel (SA1-4 in one launch pair)",
                "tests_per_step": head["bq_tests"], "flop_per_test": 8,
                "achieved": round(head["bq_tests"] * 8 / (sum(bq) / len(bq) * 1e-3) / 1e12, 4), "peak": VALU_F32_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(head["bq_tests"] * 8 / (sum(bq) / len(bq) * 1e-3) / 1e12 / VALU_F32_PEAK_TFLOPS, 5),
                "lds_bytes_per_step": head["bq_tests"] * 16,
                "lds_frac": round(head["bq_tests"] * 16 / (sum(bq) / len(bq) * 1e-3) / 1e9 / LDS_PEAK_GBS, 5),
                "avg_launch_pair_us": round(sum(bq) / len(bq) * 1e3, 2),
                "note": "latency-bound: 40 000 points per scene walk ~5 candidate centres each through dependent LDS reads"},
            "roofline_group_dense": {"bound": "hbm", "kernel": "query_group_fused_kernel + query_group_fused_pm_kernel "
                                                               "(levels that form the dense grouped tensor)",
                                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                                     "algorithmic_bytes_per_launch": round(grp_bytes / max(len(grp), 1)),
                                     "launches": len(grp), "avg_launch_us": round(sum(grp) / max(len(grp), 1) * 1e3, 2)},
            "roofline_adamw": {"bound": "hbm", "kernel": "adamw_table_kernel", "achieved": round(adam_gbs, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(adam_gbs / HBM_PEAK_GBS, 4),
                               "algorithmic_bytes_per_launch": 28 * n_params,
                               "avg_launch_us": round(sum(adam) / max(len(adam), 1) * 1e3, 1)},
            "kernels_ms_per_step": {"query_group_fused": round(sum(grp) / KSTEPS, 4),
                                    "query_group_compact": round(sum(cgrp) / KSTEPS, 4),
                                    "point_major_transposes": round(sum(tr) / KSTEPS, 4),
                                    "ball_query": round(sum(bq) / KSTEPS, 4),
                                    "furthest_point_sampling": round(sum(fps) / KSTEPS, 4),
                                    # SA1's first SharedMLP layer formed from the raw point-major scan (no grouping launch)
                                    "sa_first_layer_fwd": round(sum(kernel_ms("sig3d_sa_first_layer_fwd")) / KSTEPS, 4),
                                    "sa_first_layer_dw": round(sum(kernel_ms("sig3d_sa_first_layer_dw")) / KSTEPS, 4)},
            "compact_levels": head["compact_info"],
            "distinct_neighbour_fraction": head["distinct"],
            "launch_mode": "hipGraph replay" if head["use_graph"] else "eager",
            "library_gemms": "default heuristic" if args.no_gemm_tuning else "tuned (TunableOp)",
            "final_loss": round(head["final_loss"], 5),
            "fps_timeouts": _lib.fps_timeouts(),
        }
        if head["comm"] is not None:
            # data-parallel form (N > 1, or --force-reducer): what the gradient exchange costs the step (ddp.CommStats)
            out["comm"] = dict(head["comm"])
            out["comm"]["model"] = comm_model(world, head["comm"]["bytes_per_step"], qf_cut=args.qf_cut)
            out["comm"]["qf_cut"] = args.qf_cut
    # ---- beside the headline (N = 1): the same step with every level dense, and on surface-shaped scenes
    if world == 1 and not args.no_variants:
        vsteps, vwarm = min(args.steps, 10), min(args.warmup, 3)
        variants = {}
        for key, kw in (("dense_sa (SIG3D_COMPACT=0), 8d data", dict(surface=False, compact=False)),
                        ("synthetic-surface data", dict(surface=True, compact=True))):
            del head, model
            torch.cuda.empty_cache()
            head = model = None
            v = measure(args, rank, world, device, vsteps, vwarm, kernels=False, **kw)
            variants[key] = {"ms_per_step": round(v["dt"] / vsteps * 1e3, 3),
                             "value": round(BATCH * vsteps / v["dt"], 1), "unit": "samples/s", "steps": vsteps,
                             "compact_levels": v["compact_info"], "distinct_neighbour_fraction": v["distinct"],
                             "final_loss": round(v["final_loss"], 5)}
            del v
        # BASELINE config 2 (forward only, B = 4, eval mode, no autograd): the latency of ONE batch through a captured
        # graph (bound by the 2047 dependent FPS rounds of SA1) and the throughput with geometry chains of two
        # batches in flight (serve.PipelinedForward)
        variants["config 2: forward only, B=4"] = forward_only_variant(device)
        # BASELINE config 5 at its single-GPU shape: the BLIP-2 point branch over 80 000 / 5000 point tokens per scene
        variants["config 5: Blip2 point Q-Former, B=4, d_enc 1408"] = config5_variant(device)
        if rank == 0:
            out["variants"] = variants
    if rank == 0 and world == 1 and not args.no_ops_roofline:
        out["roofline_ops"] = ops_roofline(device)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            if model is None:   # the variants released the headline's model: same seed, initial weights
                torch.manual_seed(1234)
                model = SIG3DQFormer(num_answers=NUM_ANSWERS).to(device).train()
            out["cpu_baseline"] = cpu_baseline(model, 1234)
        line = json.dumps(out)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio (flushed at exit when stdout is a file): push it
        # out first so that the JSON line is the LAST line of rank 0's stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(line, flush=True)


if __name__ == "__main__":
    main()

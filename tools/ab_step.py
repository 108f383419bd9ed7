"""Same-process A/B of the graphed training step: GraphedTrainStep objects over ONE model / optimizer, built with a
module-level switch at value A and at value B, replayed alternately (A B B A ...) in blocks of `--replays` steps timed
with events.  Box-to-box spread (+-0.1 ms) and the warm-up drift of one box (8.05 -> 8.27 ms within a minute) cancel in
the paired differences.

WHAT THE +- DOES NOT COVER (found late in round 5 with a null switch: `env:SIG3D_NULL_SWITCH 0 1` gave +0.077 +- 0.005 and
-0.055 +- 0.002 ms for two IDENTICAL arms): every built step object has its own persistent offset of up to +-0.07 ms (where
its buffers and queues landed), the same in every replay.  One build per arm therefore resolves ~0.1 ms, not 0.003.
`--builds N` (default 3) builds N objects per arm, alternately, and reports the arm means with the standard error ACROSS
builds; a difference counts when it exceeds that.  SIG3D_GEO_DEPTH (chains in flight) defaults to 3 as in bench.py.

WHAT IT CANNOT SEE AT ALL (round 6): how long the geometry chains themselves take.  Every timed block runs behind a re-primed
pipeline (three chains = 22 ms of slack), so a chain that needs 14 ms beside the step instead of 7.7 looks free here, while
the 20 timed steps of bench.py end when their last chains end (+0.33 ms per step).  A change that touches the chain is
measured with tools/probes/chain_slack.py and with alternating bench.py runs (tools/probes/bench_fps_forms.sh) as well.

python tools/ab_step.py situation3d_amd.qformer.FUSED_EMBED False True [--rounds 24] [--replays 10]
python tools/ab_step.py env:SIG3D_GEO_HANDSHAKE 0 1
python tools/ab_step.py env:SIG3D_GEO_DEPTH 1 2
"""
import argparse, importlib, os, statistics, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import bench  # noqa: E402
from situation3d_amd import gemm_tuning  # noqa: E402
from situation3d_amd.graph_step import GraphedTrainStep  # noqa: E402
from situation3d_amd.model import SIG3DQFormer  # noqa: E402
from situation3d_amd.trainer import build_optimizer  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes"))
import geo_probes  # noqa: E402

geo_probes.install()      # SIG3D_PROBE_* switches live in tools/, not in the product (bench.py refuses them)

ap = argparse.ArgumentParser()
ap.add_argument("switch", help="module.ATTRIBUTE, e.g. situation3d_amd.qformer.FUSED_EMBED")
ap.add_argument("a"); ap.add_argument("b")
ap.add_argument("--rounds", type=int, default=24)
ap.add_argument("--replays", type=int, default=10)
ap.add_argument("--builds", type=int, default=3, help="step objects per arm (their spread is the real uncertainty)")
args = ap.parse_args()
if args.switch.startswith("env:"):        # an environment variable read when the step is BUILT (graph_step / geometry)
    class _Env:
        pass
    mod, attr = _Env(), args.switch[4:]
    _Env.__setattr__ = lambda self, k, v: os.environ.__setitem__(k, str(v))
else:
    mod_name, attr = args.switch.rsplit(".", 1)
    mod = importlib.import_module(mod_name)
lit = lambda s: {"True": True, "False": False}.get(s, int(s) if s.lstrip("-").isdigit() else s)  # noqa: E731
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
gemm_tuning.enable(tune_missing=True)
torch.manual_seed(1234)
model = SIG3DQFormer(num_answers=bench.NUM_ANSWERS).to(dev).train()
opt = build_optimizer(model, name="flat_adamw")
batches = [bench.synthetic_batch(bench.BATCH, bench.N_POINTS, 1234 + i, dev) for i in range(4)]
work = torch.cuda.Stream(dev, priority=int(os.environ.get("AB_WORK_PRIORITY", "0")))   # -1: a high-priority queue
steps = {}
names = []
with torch.cuda.stream(work):
    for i in range(args.builds):
        for arm, val in (("A", lit(args.a)), ("B", lit(args.b))):
            setattr(mod, attr, val)
            steps[arm + str(i)] = GraphedTrainStep(model, opt, batches[0], prefetch_geometry=True,
                                                   prefetch_depth=int(os.environ.get("SIG3D_GEO_DEPTH", "3")))
            names.append(arm + str(i))
    k = [0]

    def block(name):
        g = steps[name]
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        d = g.prefetch_depth
        for _ in range(1 + d):                  # re-prime this variant's geometry pipeline (untimed)
            g(batches[k[0] % 4], upcoming=[batches[(k[0] + 1 + i) % 4] for i in range(d)]); k[0] += 1
        s.record(work)
        for _ in range(args.replays):
            g(batches[k[0] % 4], upcoming=[batches[(k[0] + 1 + i) % 4] for i in range(d)]); k[0] += 1
        e.record(work)
        e.synchronize()
        return s.elapsed_time(e) / args.replays

    for name in names + names[::-1]:
        block(name)                             # warm-up
    t = {n: [] for n in names}
    for r in range(args.rounds):
        order = names + names[::-1] if r % 2 == 0 else names[::-1] + names
        for name in order:
            t[name].append(block(name))
per = {n: statistics.mean(v) for n, v in t.items()}
arm = {x: [per[n] for n in names if n[0] == x] for x in "AB"}
mean = {x: statistics.mean(arm[x]) for x in "AB"}
se = {x: (statistics.stdev(arm[x]) / len(arm[x]) ** 0.5 if len(arm[x]) > 1 else float("nan")) for x in "AB"}
print("%s: A=%s %.3f ms [%s]   B=%s %.3f ms [%s]   B - A = %+.3f ms +- %.3f (standard error across %d builds per arm; "
      "%d rounds x 2 x %d replays each)"
      % (args.switch, args.a, mean["A"], " ".join("%.3f" % v for v in arm["A"]), args.b, mean["B"],
         " ".join("%.3f" % v for v in arm["B"]), mean["B"] - mean["A"], (se["A"] ** 2 + se["B"] ** 2) ** 0.5, args.builds,
         args.rounds, args.replays))

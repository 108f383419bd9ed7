"""How much slack does the geometry chain have?  For every step: when the step's stream arrives at the hand-over
(ready) and when the chain of that batch finished (done), from events recorded around GeometryPipeline.advance.
done - ready > 0: the step WAITED for its geometry."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from situation3d_amd import gemm_tuning
from situation3d_amd.graph_step import GraphedTrainStep
from situation3d_amd.model import SIG3DQFormer
from situation3d_amd.trainer import build_optimizer
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
gemm_tuning.enable(tune_missing=True)
torch.manual_seed(1234)
model = SIG3DQFormer(num_answers=bench.NUM_ANSWERS).to(dev).train()
opt = build_optimizer(model, name="flat_adamw")
batches = [bench.synthetic_batch(bench.BATCH, bench.N_POINTS, 1234 + i, dev) for i in range(4)]
work = torch.cuda.Stream(dev)
recs = []
with torch.cuda.stream(work):
    depth = int(os.environ.get("SIG3D_GEO_DEPTH", "1"))
    g = GraphedTrainStep(model, opt, batches[0], prefetch_geometry=True, prefetch_depth=depth)
    pipe = g._pipe
    orig = pipe.advance
    state = {"done": [], "n": 0}

    def advance(pc, upcoming, token=None, upcoming_tokens=None):
        ready = torch.cuda.Event(enable_timing=True); ready.record(work)
        # the chain consumed NOW was launched `depth` calls ago (round-robin over the slots)
        prev_done = state["done"][-depth] if len(state["done"]) >= depth else None
        orig(pc, upcoming, token, upcoming_tokens)
        slot = pipe.slots[(pipe.calls - 1) % pipe.depth]
        done = torch.cuda.Event(enable_timing=True); done.record(slot["stream"])     # end of the chain launched NOW
        start = torch.cuda.Event(enable_timing=True); start.record(work)
        if prev_done is not None:
            recs.append((ready, prev_done, start))
        state["done"].append(done)
        state.setdefault("span", []).append((start, done))      # this chain: ticket signalled -> chain finished
    pipe.advance = advance
    for i in range(120):
        up = [batches[(i + 1 + k) % 4] for k in range(depth)]
        g(batches[i % 4], upcoming=up)
    torch.cuda.synchronize()
wait = [r.elapsed_time(d) for r, d, s in recs[20:]]          # ready -> chain done (ms); > 0: the step waited
print("steps %d: chain done minus step ready: mean %+.3f ms, min %+.3f, max %+.3f; steps that waited: %d (%.0f %%), mean wait %.3f ms"
      % (len(wait), statistics.mean(wait), min(wait), max(wait), sum(w > 0 for w in wait),
         100.0 * sum(w > 0 for w in wait) / len(wait), statistics.mean([max(w, 0.0) for w in wait])))
q = sorted(wait)
print("quantiles (ms): 10%% %+.3f  50%% %+.3f  90%% %+.3f  99%% %+.3f" % (q[len(q) // 10], q[len(q) // 2], q[9 * len(q) // 10], q[-1]))
span = [a.elapsed_time(b) for a, b in state["span"][20:]]
print("chain duration (ticket signalled on the step's stream -> last kernel of the chain): mean %.3f ms, min %.3f, max %.3f"
      % (statistics.mean(span), min(span), max(span)))

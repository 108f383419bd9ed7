"""Forward-only throughput (BASELINE config 2: B=4, 40k points, eval mode, no autograd), eager and as a
captured hipGraph.  python tools/infer_bench.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from situation3d_amd import gemm_tuning
from situation3d_amd.model import SIG3DQFormer

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
gemm_tuning.enable(tune_missing=True)
torch.manual_seed(0)
B = 4
model = SIG3DQFormer(num_answers=bench.NUM_ANSWERS).to(dev).eval()
batch = bench.synthetic_batch(B, bench.N_POINTS, 7, dev)
work = torch.cuda.Stream(dev)
with torch.cuda.stream(work), torch.no_grad():
    for _ in range(3):
        out = model(dict(batch))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = model(dict(batch))
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 10
    g = torch.cuda.CUDAGraph()
    with gemm_tuning.no_tuning(), torch.cuda.graph(g, stream=work):
        out = model(dict(batch))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    graphed = (time.perf_counter() - t0) / 20
    # serving loop: geometry of batch i+1 under the forward of batch i (situation3d_amd/serve.py)
    from situation3d_amd.serve import GraphedForward
    batches = [bench.synthetic_batch(B, bench.N_POINTS, 100 + i, dev) for i in range(3)]
    step = GraphedForward(model, batches[0])
    refs = [model(dict(bt))["answer_scores"].clone() for bt in batches]
    for i in range(6):
        out = step(batches[i % 3], batches[(i + 1) % 3])
        assert torch.equal(out["answer_scores"], refs[i % 3]), "prefetched forward differs from the inline one"
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(30):
        step(batches[i % 3], batches[(i + 1) % 3])
    torch.cuda.synchronize()
    piped = (time.perf_counter() - t0) / 30
print("forward only, B=%d x %d pts, geometry prefetched one batch ahead: %.2f ms (%.0f samples/s)"
      % (B, bench.N_POINTS, piped * 1e3, B / piped))
# the FPS chain of the prefetch branch costs the same for 8 scenes as for 4 (one cooperative launch holds 8):
with torch.cuda.stream(work), torch.no_grad():
    b8 = [bench.synthetic_batch(8, bench.N_POINTS, 200 + i, dev) for i in range(3)]
    step8 = GraphedForward(model, b8[0])
    for i in range(4):
        step8(b8[i % 3], b8[(i + 1) % 3])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(30):
        step8(b8[i % 3], b8[(i + 1) % 3])
    torch.cuda.synchronize()
    p8 = (time.perf_counter() - t0) / 30
print("forward only, B=8 x %d pts, geometry prefetched one batch ahead: %.2f ms (%.0f samples/s)"
      % (bench.N_POINTS, p8 * 1e3, 8 / p8))
# two geometry chains in flight (serve.PipelinedForward): the FPS chain of one batch no longer bounds the step
from situation3d_amd.serve import PipelinedForward
for bsz in (4, 8):
    with torch.cuda.stream(work), torch.no_grad():
        bs = [bench.synthetic_batch(bsz, bench.N_POINTS, 300 + i, dev) for i in range(4)]
        refs2 = [model(dict(bt))["answer_scores"].clone() for bt in bs]
        for depth, hp in ((2, True), (3, True), (2, False), (3, False)):
            pipe = PipelinedForward(model, bs[0], depth=depth, high_priority=hp)
            for i in range(9):
                out = pipe(bs[i % 4], [bs[(i + 1 + k) % 4] for k in range(depth)])
                assert torch.equal(out["answer_scores"], refs2[i % 4]), "pipelined forward differs from the inline one"
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(9, 49):
                pipe(bs[i % 4], [bs[(i + 1 + k) % 4] for k in range(depth)])
            torch.cuda.synchronize()
            pp = (time.perf_counter() - t0) / 40
            print("forward only, B=%d x %d pts, %d geometry chains in flight (%s-priority streams): %.2f ms (%.0f samples/s)"
                  % (bsz, bench.N_POINTS, depth, "high" if hp else "normal", pp * 1e3, bsz / pp))
from situation3d_amd import _lib
print("cooperative-FPS timeouts over the whole run:", _lib.fps_timeouts())
print("forward only, B=%d x %d pts: eager %.2f ms (%.0f samples/s), hipGraph %.2f ms (%.0f samples/s)"
      % (B, bench.N_POINTS, eager * 1e3, B / eager, graphed * 1e3, B / graphed))

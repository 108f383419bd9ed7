"""What does a fork inside a hipGraph cost per node?  The Q-Former's forward + backward (~450 kernel nodes) alone, with a
one-element fill on a forked branch (joined at once / joined at the end), and beside a SEPARATE graph on another stream
(bounded AdamW) -- tools/adamw_overlap.py measured forked = sum + 0.7 ms whatever the update's grid was."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from situation3d_amd import _lib, gemm_tuning
from situation3d_amd.model import SIG3DQFormer
from situation3d_amd.trainer import build_optimizer, train_step
dev = torch.device("cuda", 0)
gemm_tuning.enable(tune_missing=True)
torch.manual_seed(0)
model = SIG3DQFormer(num_answers=bench.NUM_ANSWERS).to(dev).train()
opt = build_optimizer(model, name="flat_adamw")
batch = bench.synthetic_batch(bench.BATCH, bench.N_POINTS, 7, dev)
main, side = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(main)
    for _ in range(n):
        fn()
    e.record(main); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


with torch.cuda.stream(main):
    for _ in range(2):
        train_step(model, opt, dict(batch))
    tokens = torch.randn(bench.BATCH, 256, 256, device=dev, requires_grad=True)
    q = batch["q_feat"]
    ones = torch.ones(bench.BATCH, 32, dtype=q["attention_mask"].dtype, device=dev)
    mask = torch.cat([ones, q["attention_mask"]], dim=1)
    scratch = torch.zeros(64, device=dev)

    def qf():
        out = model.Qformer.bert(query_embeds=model.query_tokens.expand(bench.BATCH, -1, -1), input_ids=q["input_ids"],
                                 attention_mask=mask, encoder_hidden_states=tokens, encoder_attention_mask=None,
                                 return_dict=True)
        h = getattr(out, "query_hidden_state", None)
        h = out.last_hidden_state[:, :32] if h is None else h
        h.float().pow(2).mean().backward()

    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    opt._upload()
    g0 = opt.param_groups[0]; b1, b2 = g0["betas"]
    lr0 = torch.zeros(1, device=dev)

    def launch_adam(bound=256):
        _lib.call("sig3d_adamw_table_bounded", len(opt._static), _lib.ptr(opt._table), _lib.ptr(opt._step), ctypes.c_float(0.0),
                  _lib.ptr(lr0), ctypes.c_float(b1), ctypes.c_float(b2), ctypes.c_float(g0["eps"]), ctypes.c_float(opt.clip_value),
                  int(bound), _lib.stream_ptr(dev))

    qf(); qf(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with gemm_tuning.no_tuning(), torch.cuda.graph(g, stream=main):
        qf()
    print("nodes: see rocprof; Q-Former fwd+bwd alone        %.3f ms" % timed(g.replay))
    g1 = torch.cuda.CUDAGraph()
    with gemm_tuning.no_tuning(), torch.cuda.graph(g1, stream=main):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            scratch.fill_(1.0)
        main.wait_stream(side)
        qf()
    print("fork (1-element fill) joined at once             %.3f ms" % timed(g1.replay))
    g2 = torch.cuda.CUDAGraph()
    with gemm_tuning.no_tuning(), torch.cuda.graph(g2, stream=main):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            scratch.fill_(1.0)
        qf()
        main.wait_stream(side)
    print("fork (1-element fill) joined at the end          %.3f ms" % timed(g2.replay))
    for bound in (256, 128, 64):
        ga = torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga, stream=side):
            with torch.cuda.stream(side):
                launch_adam(bound)
        def alone_side():
            side.wait_stream(main); 
            with torch.cuda.stream(side):
                ga.replay()
            main.wait_stream(side)
        t_a = timed(alone_side)
        def both():
            side.wait_stream(main)
            with torch.cuda.stream(side):
                ga.replay()
            g.replay()
            main.wait_stream(side)
        t = timed(both)
        print("AdamW grid %3d as its own graph on a side stream: alone %.3f ms, beside the Q-Former %.3f ms" % (bound, t_a, t))
    # a second ACTIVE queue without resource contention: one spinning thread (torch.cuda._sleep) beside the graph
    def spin_beside():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            torch.cuda._sleep(int(5.0e-3 * 2.4e9))
        g.replay()
        main.wait_stream(side)
    print("one spinning thread (~5 ms) on a side stream beside the graph   %.3f ms" % timed(spin_beside))
    def fill_beside():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            scratch.fill_(1.0)
        g.replay()
        main.wait_stream(side)
    print("eager 1-element fill on a side stream beside the graph          %.3f ms" % timed(fill_beside))
    g3 = torch.cuda.CUDAGraph()
    with gemm_tuning.no_tuning(), torch.cuda.graph(g3, stream=main):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            torch.cuda._sleep(int(5.0e-3 * 2.4e9))
        qf()
        main.wait_stream(side)
    print("fork (spinning thread ~5 ms) joined at the end                  %.3f ms" % timed(g3.replay))
    print("graph alone again (side queue now exists, idle)                 %.3f ms" % timed(g.replay))
    def fill_spin():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            scratch.fill_(1.0)
            torch.cuda._sleep(int(5.0e-3 * 2.4e9))
        g.replay()
        main.wait_stream(side)
    print("fill + spinning thread on the side stream beside the graph      %.3f ms" % timed(fill_spin))
    def fill_nojoin():
        with torch.cuda.stream(side):
            scratch.fill_(1.0)
        g.replay()
    print("fill on the side stream, no waits either way                    %.3f ms" % timed(fill_nojoin))
    def fill_forkonly():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            scratch.fill_(1.0)
        g.replay()
    print("fill on the side stream, fork wait only                         %.3f ms" % timed(fill_forkonly))
    def fill_joinonly():
        with torch.cuda.stream(side):
            scratch.fill_(1.0)
        g.replay()
        main.wait_stream(side)
    print("fill on the side stream, join wait only                         %.3f ms" % timed(fill_joinonly))
    print("graph alone again                                               %.3f ms" % timed(g.replay))
    # the fork wait as a device-side handshake (sig3d_ticket_signal / sig3d_ticket_wait) instead of a barrier packet
    words = torch.zeros(4, dtype=torch.int32, device=dev)
    tick, seen, err = words[0:1], words[1:2], words[2:3]
    def handshake():
        _lib.call("sig3d_ticket_signal", _lib.ptr(tick), _lib.stream_ptr(dev))
        with torch.cuda.stream(side):
            _lib.call("sig3d_ticket_wait", _lib.ptr(tick), _lib.ptr(seen), 2000000, _lib.ptr(err), _lib.stream_ptr(dev))
            scratch.fill_(1.0)
        g.replay()
    print("fill on the side stream behind a device-side handshake          %.3f ms" % timed(handshake))
    def handshake_join():
        _lib.call("sig3d_ticket_signal", _lib.ptr(tick), _lib.stream_ptr(dev))
        with torch.cuda.stream(side):
            _lib.call("sig3d_ticket_wait", _lib.ptr(tick), _lib.ptr(seen), 2000000, _lib.ptr(err), _lib.stream_ptr(dev))
            scratch.fill_(1.0)
        g.replay()
        main.wait_stream(side)
    print("  ... and joined with a stream wait at the end                  %.3f ms" % timed(handshake_join))
    torch.cuda.synchronize()
    print("handshake words (ticket, consumed, error):", words[:3].tolist())
    # the same blocked barrier on a HIGH-priority stream (RCCL's stream in the data-parallel step: ddp._pg_options)
    side_hi = torch.cuda.Stream(dev, priority=-1)
    def fill_forkonly_hi():
        side_hi.wait_stream(main)
        with torch.cuda.stream(side_hi):
            scratch.fill_(1.0)
        g.replay()
    print("fill on a HIGH-priority side stream, fork wait only             %.3f ms" % timed(fill_forkonly_hi))
    print("graph alone again                                               %.3f ms" % timed(g.replay))

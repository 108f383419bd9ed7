"""N>1 path on CPU: two gloo ranks, the bucketed gradient reducer must reproduce the reference
DDP semantics (mean of per-rank gradients, identical parameters after a step on every rank;
runner_base.py:88-95), including parameters that receive no gradient."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.body = nn.Sequential(nn.Linear(16, 64), nn.GELU(), nn.Linear(64, 64), nn.GELU(),
                                  nn.Linear(64, 4))
        self.unused = nn.Linear(8, 8)  # never used in forward: gets no gradient

    def forward(self, x):
        return self.body(x)


def _make_model():
    torch.manual_seed(0)
    return _Net()


def _data(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(8, 16, generator=g), torch.randn(8, 4, generator=g)


def _worker(rank, world, port, out_dir, deferred):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from situation3d_amd.ddp import GradBucketReducer, init_distributed
    init_distributed(backend="gloo")
    model = _make_model()
    # tiny buckets: several collectives, exercised in backward order
    reducer = GradBucketReducer(model.parameters(), bucket_bytes=4096)
    assert reducer.num_collectives() >= 3
    reducer.hooks_enabled = not deferred  # deferred: what the hipGraph step does (reduce_all)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2)
    x, y = _data(rank)
    for _ in range(2):
        reducer.zero_grad()
        loss = ((model(x) - y) ** 2).mean()
        loss.backward()
        if deferred:
            reducer.reduce_all()
        else:
            reducer.finish()
        grads = [p.grad.clone() for p in model.parameters()]
        opt.step()
    torch.save({"grads": grads, "params": [p.detach().clone() for p in model.parameters()]},
               os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("deferred", [False, True])
def test_bucket_reducer_world2(tmp_path, deferred):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), deferred), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b), "ranks diverged"
    for a, b in zip(r0["grads"], r1["grads"]):
        assert torch.equal(a, b)

    # single-process reference: mean of the two ranks' gradients, same optimiser
    model = _make_model()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2)
    for _ in range(2):
        opt.zero_grad(set_to_none=False)
        for p in model.parameters():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        for rank in range(world):
            x, y = _data(rank)
            (((model(x) - y) ** 2).mean() / world).backward()
        grads = [p.grad.clone() for p in model.parameters()]
        opt.step()
    for a, b in zip(r0["grads"], grads):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
    for a, b in zip(r0["params"], model.parameters()):
        torch.testing.assert_close(a, b.detach(), rtol=1e-5, atol=1e-6)


def test_bucket_reducer_single_process_is_identity():
    from situation3d_amd.ddp import GradBucketReducer
    model = _make_model()
    reducer = GradBucketReducer(model.parameters(), bucket_bytes=1 << 20)
    x, y = _data(0)
    reducer.zero_grad()
    ((model(x) - y) ** 2).mean().backward()
    reducer.finish()
    ref = _make_model()
    ((ref(x) - y) ** 2).mean().backward()
    for p, q in zip(model.parameters(), ref.parameters()):
        if q.grad is None:
            assert p.grad.abs().max() == 0
        else:
            assert torch.equal(p.grad, q.grad)


def _flat_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from situation3d_amd.ddp import GradBucketReducer, init_distributed
    init_distributed(backend="gloo")
    # two flat gradient buffers (decay / no-decay groups of optim.FlatAdamW), sliced into small buckets
    g = torch.Generator().manual_seed(10 + rank)
    flats = [torch.randn(5000, generator=g), torch.randn(300, generator=g)]
    reducer = GradBucketReducer.from_flat(flats, bucket_bytes=4096)
    assert reducer.flat_mode and reducer.num_collectives() == 6
    # the protocol of FlatAdamW.step_after: launch everything, then consume bucket by bucket
    reducer.launch_all()
    consumed = []
    for b in reducer.buckets:
        reducer.wait(b)
        consumed.append(b["flat"].clone())   # "update" reads the reduced slice right away
    torch.save({"flats": [f.clone() for f in flats], "consumed": consumed}, os.path.join(out_dir, "f%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_launch_all_then_wait_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_flat_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "f0.pt"), torch.load(tmp_path / "f1.pt")
    exp = []
    for i, n in enumerate((5000, 300)):
        parts = []
        for rank in range(world):
            g = torch.Generator().manual_seed(10 + rank)
            both = [torch.randn(5000, generator=g), torch.randn(300, generator=g)]
            parts.append(both[i])
        exp.append(sum(parts) / world)
    for got0, got1, e in zip(r0["flats"], r1["flats"], exp):
        assert torch.equal(got0, got1)
        torch.testing.assert_close(got0, e, rtol=1e-6, atol=1e-6)
    # every bucket was already reduced at the moment it was consumed
    torch.testing.assert_close(torch.cat(r0["consumed"]), torch.cat(exp), rtol=1e-6, atol=1e-6)


def _rows_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from situation3d_amd.ddp import SparseRowExchange, embedding_rows, init_distributed
    init_distributed(backend="gloo")
    torch.manual_seed(0)
    table = nn.Parameter(torch.randn(50, 8))
    g = torch.Generator().manual_seed(7 + rank)
    ids = torch.randint(0, 50, (3, 5), generator=g)
    ids[0, :3] = 4                      # duplicates on a rank; id 4 is also hit on the other rank
    target = torch.randn(3, 5, 8, generator=g)
    sink = SparseRowExchange(ids.numel(), 8, "cpu")
    out = embedding_rows(table, ids, sink)
    ((out - target) ** 2).sum().backward()
    assert table.grad is None           # no dense gradient from autograd
    sink.launch()
    dense = torch.zeros(50, 8)
    sink.finish_into(dense)
    torch.save({"dense": dense, "ids": ids, "target": target}, os.path.join(out_dir, "rows%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_embedding_rows_exchange_equals_mean_of_dense_gradients_world2(tmp_path):
    """ddp.SparseRowExchange: ids + per-position rows all-gathered, scatter-added / world == what an all-reduce
    (mean) of the dense embedding gradients gives -- duplicates within and across ranks included."""
    world, port = 2, _free_port()
    mp.spawn(_rows_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / ("rows%d.pt" % k)) for k in range(world)]
    assert torch.equal(r[0]["dense"], r[1]["dense"])
    torch.manual_seed(0)
    table = nn.Parameter(torch.randn(50, 8))
    for k in range(world):
        out = torch.nn.functional.embedding(r[k]["ids"], table)
        (((out - r[k]["target"]) ** 2).sum() / world).backward()
    torch.testing.assert_close(r[0]["dense"], table.grad, rtol=1e-5, atol=1e-6)
    assert (table.grad[4] != 0).any()


def test_embedding_rows_single_process():
    from situation3d_amd.ddp import SparseRowExchange, embedding_rows
    torch.manual_seed(1)
    table = nn.Parameter(torch.randn(20, 4))
    ids = torch.tensor([[1, 1, 3], [19, 0, 1]])
    sink = SparseRowExchange(6, 4, "cpu")
    embedding_rows(table, ids, sink).pow(2).sum().backward()
    dense = torch.zeros(20, 4)
    sink.launch()
    sink.finish_into(dense)
    ref = nn.Parameter(table.detach().clone())
    torch.nn.functional.embedding(ids, ref).pow(2).sum().backward()
    torch.testing.assert_close(dense, ref.grad)


def test_embedding_rows_respect_padding_idx():
    """nn.Embedding(padding_idx=0) (Qformer.py:56): the [PAD] row gets a zero gradient even when its positions
    receive upstream gradient; the row exchange must agree with the dense backward."""
    from situation3d_amd.ddp import SparseRowExchange, embedding_rows
    torch.manual_seed(2)
    table = nn.Parameter(torch.randn(20, 4))
    ids = torch.tensor([[0, 1, 3], [19, 0, 0]])
    sink = SparseRowExchange(6, 4, "cpu", padding_idx=0)
    embedding_rows(table, ids, sink).pow(2).sum().backward()
    dense = torch.zeros(20, 4)
    sink.launch()
    sink.finish_into(dense)
    ref = nn.Parameter(table.detach().clone())
    torch.nn.functional.embedding(ids, ref, padding_idx=0).pow(2).sum().backward()
    assert float(ref.grad[0].abs().max()) == 0.0
    torch.testing.assert_close(dense, ref.grad)


def _cut_net():
    from situation3d_amd.qformer import init_Qformer

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = nn.Linear(8, 8)
            self.Qformer, self.query_tokens = init_Qformer(32, 64, hidden_size=64, num_hidden_layers=6,
                                                           num_attention_heads=2, intermediate_size=128,
                                                           max_position_embeddings=64, vocab_size=50)
            self.answer_cls = nn.Linear(64, 4)
    torch.manual_seed(0)
    return Net()


def _cut_layout(net, cut):
    """What graph_step.GraphedTrainStep does with a FlatAdamW built by build_optimizer(qf_cut=cut), on the host: the two
    groups' storage order, their flat gradient buffers, and the flat ranges of the three pieces of the backward pass."""
    from situation3d_amd.optim import flat_offsets, part_runs
    from situation3d_amd.trainer import storage_layout
    decay = [p for n, p in net.named_parameters() if "bias" not in n and "LayerNorm.weight" not in n]
    no_decay = [p for n, p in net.named_parameters() if "bias" in n or "LayerNorm.weight" in n]
    layers = list(net.Qformer.bert.encoder.layer)
    upper = [p for m in layers[cut:] + [net.answer_cls] for p in m.parameters()]
    up = {id(p) for p in upper}
    lower = [p for p in net.Qformer.parameters() if id(p) not in up] + [net.query_tokens]
    flats, pieces = [], [[], [], []]      # pieces: upper | lower | the rest (encoder)
    for laid in storage_layout(net, [decay, no_decay], qf_cut=cut):
        offs, total = flat_offsets(laid)
        flats.append((laid, offs, torch.zeros(total)))
        for k, runs in enumerate(part_runs(list(zip(laid, offs)), total, [upper, lower])):
            pieces[k] += [flats[-1][2][lo:hi] for lo, hi in runs]
    return flats, pieces, (upper, lower)


def _cut_worker(rank, world, port, out_dir, cut):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from situation3d_amd.ddp import GradBucketReducer, init_distributed
    init_distributed(backend="gloo")
    net = _cut_net()
    flats, pieces, (upper, lower) = _cut_layout(net, cut)
    reds = [GradBucketReducer.from_flat(ps, bucket_bytes=1 << 16) for ps in pieces]
    g = torch.Generator().manual_seed(50 + rank)
    grads = {id(p): torch.randn(p.shape, generator=g) for p in net.parameters()}

    def produce(params):     # a piece of the backward pass has run: its gradients are in the flat buffers
        want = {id(p) for p in params}
        for laid, offs, flat in flats:
            for p, off in zip(laid, offs):
                if id(p) in want:
                    flat[off:off + p.numel()] = grads[id(p)].reshape(-1)

    # the step's order (graph_step.GraphedTrainStep.__call__): upper piece -> on the wire; lower piece computes ->
    # on the wire; encoder computes -> on the wire; then the update consumes bucket by bucket, piece by piece
    produce(upper); reds[0].launch_all()
    produce(lower); reds[1].launch_all()
    produce(list(net.encoder.parameters())); reds[2].launch_all()
    for r in reds:
        for b in r.buckets:
            r.wait(b)
    out = {}
    for laid, offs, flat in flats:
        for p, off in zip(laid, offs):
            out[[n for n, q in net.named_parameters() if q is p][0]] = flat[off:off + p.numel()].clone().view_as(p)
    torch.save({"reduced": out, "runs": [len(ps) for ps in pieces], "buckets": [len(r.buckets) for r in reds]},
               os.path.join(out_dir, "c%d_%d.pt" % (cut, rank)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("cut", [3])
def test_cut_backward_pieces_over_two_arena_storage_reduce_to_the_mean_world2(tmp_path, cut):
    """The data-parallel step with the backward pass also cut inside the Q-Former (graph_step, qf_cut): over the
    two-arena storage (trainer.storage_layout(qf_cut)) every piece owns a few long stretches of the flat gradients;
    three bucket sets launched piece by piece, in the step's order, leave every parameter's mean gradient on both ranks."""
    world, port = 2, _free_port()
    mp.spawn(_cut_worker, args=(world, port, str(tmp_path), cut), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / ("c%d_0.pt" % cut)), torch.load(tmp_path / ("c%d_1.pt" % cut))
    net = _cut_net()
    gens = [torch.Generator().manual_seed(50 + rank) for rank in range(world)]
    per_rank = [{n: torch.randn(p.shape, generator=g) for n, p in net.named_parameters()} for g in gens]
    for n, _ in net.named_parameters():
        torch.testing.assert_close(r0["reduced"][n], (per_rank[0][n] + per_rank[1][n]) / 2, rtol=1e-6, atol=1e-6)
        assert torch.equal(r0["reduced"][n], r1["reduced"][n])
    # stretches per piece: (decay, no-decay) x (arena [+ heads / embeddings]) -- not a slice per kind
    assert r0["runs"][0] <= 4 and r0["runs"][1] <= 4, r0["runs"]
    # ... while ONE kind-major arena interleaves the pieces kind by kind
    _, one_arena, _ = _cut_layout_pieces_one_arena(net, cut)
    assert len(one_arena[0]) + len(one_arena[1]) > 3 * (r0["runs"][0] + r0["runs"][1])


def _cut_layout_pieces_one_arena(net, cut):
    from situation3d_amd.optim import flat_offsets, part_runs
    from situation3d_amd.trainer import storage_layout
    decay = [p for n, p in net.named_parameters() if "bias" not in n and "LayerNorm.weight" not in n]
    no_decay = [p for n, p in net.named_parameters() if "bias" in n or "LayerNorm.weight" in n]
    layers = list(net.Qformer.bert.encoder.layer)
    upper = [p for m in layers[cut:] + [net.answer_cls] for p in m.parameters()]
    up = {id(p) for p in upper}
    lower = [p for p in net.Qformer.parameters() if id(p) not in up] + [net.query_tokens]
    pieces = [[], [], []]
    for laid in storage_layout(net, [decay, no_decay], qf_cut=None):
        offs, total = flat_offsets(laid)
        for k, runs in enumerate(part_runs(list(zip(laid, offs)), total, [upper, lower])):
            pieces[k] += runs
    return None, pieces, None

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

"""world_size-2 gloo test of the graph-sharded data-parallel step (esc_gnn_amd.parallel) on CPU.
Compute = the oracle model (tests may use it); checked against a single-process evaluation of the
same sharded objective: sum_r sum_i |err| / N_global with per-shard BatchNorm statistics."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_collate


def _shards(world):
    import esc_gnn_amd as E
    graphs, _, _ = load_collate("mixed4")
    datas = [E.Data(**{k: torch.tensor(v) for k, v in g.items()}) for g in graphs]
    out = []
    for r in range(world):
        lo, hi = E.parallel.shard_slice(len(datas), r, world)
        b = E.Batch.from_data_list(datas[lo:hi])
        # the dataset's x is ones[n,10]; that makes x_embedding's BatchNorms zero-variance and their fp32
        # gradients pure amplified noise, so this test feeds a deterministic non-constant x instead
        g = torch.Generator().manual_seed(100 + r)
        b.x = torch.randn(b.x.shape, generator=g)
        out.append(dict(x=b.x, edge_index=b.edge_index, pos_enc=b.pos_enc, pos_index=b.pos_index,
                        pos_batch=b.pos_batch, batch=b.batch, y=b.y.float()))
    return out


def _model():
    import ref_model as rm
    torch.manual_seed(7)
    return rm.NestedGINEffRef(2, 8)


def _worker(rank, world, port, q, mode="weighted"):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import esc_gnn_amd as E
    torch.set_num_threads(1)
    m = _model()
    if rank == 1:                                  # replicas must be made identical by the broadcast
        with torch.no_grad():
            for p in m.parameters():
                p.add_(1.0)
    E.parallel.broadcast_parameters(m, 0)
    bucket = E.parallel.FlatBucket(m.parameters())
    b = _shards(world)[rank]
    bucket.zero_grad()
    pred = m(b["x"], b["edge_index"], b["pos_enc"], b["pos_index"], b["pos_batch"], b["batch"])
    if mode == "weighted":                         # local-mean gradients, scaled around the collective
        torch.nn.functional.l1_loss(pred, b["y"].view(-1, 1)).backward()
        total = bucket.all_reduce_weighted(b["x"].size(0))
    else:                                          # sum-gradients; the division belongs to the optimiser launch
        (pred - b["y"].view(-1, 1)).abs().sum().backward()
        total = bucket.all_reduce_sum(b["x"].size(0))
        bucket.flat_grad.div_(total)               # what esc_adam_step_scaled does per element
    # parameters / gradients are views into the (64-byte aligned, padded) flat buffers: compare them unpadded
    grads = torch.cat([p.grad.reshape(-1) for p in bucket.params])
    params = torch.cat([p.data.reshape(-1) for p in bucket.params])
    assert all(p.grad.data_ptr() == bucket.flat_grad[o:].data_ptr() and o % bucket.ALIGN == 0
               for p, o in zip(bucket.params, bucket.offsets))
    q.put((rank, grads.detach().numpy().copy(), float(total), params.detach().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["weighted", "sum"])
def test_two_rank_gradient_equals_sharded_objective(mode):
    world, port = 2, 29000 + (os.getpid() + (7 if mode == "sum" else 0)) % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    res = [(r, torch.from_numpy(g), t, torch.from_numpy(p)) for r, g, t, p in res]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference of the same objective
    torch.set_num_threads(1)
    import esc_gnn_amd as E
    m = _model()
    shards = _shards(world)
    n_glob = sum(s["x"].size(0) for s in shards)
    loss = 0
    for s in shards:
        pred = m(s["x"], s["edge_index"], s["pos_enc"], s["pos_index"], s["pos_batch"], s["batch"])
        loss = loss + (pred - s["y"].view(-1, 1)).abs().sum() / n_glob
    loss.backward()
    want = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    for rank, grad, total, params in res:
        assert total == n_glob
        err = float((grad - want).abs().max()) / max(1.0, float(want.abs().max()))
        assert err < 1e-5, (rank, err)
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][3], res[1][3])


def test_shard_slice_covers_batch():
    import esc_gnn_amd as E
    for B in (128, 7, 3):
        for W in (1, 2, 4, 8):
            parts = [E.parallel.shard_slice(B, r, W) for r in range(W)]
            assert parts[0][0] == 0 and parts[-1][1] == B
            assert all(parts[i][1] == parts[i + 1][0] for i in range(W - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1

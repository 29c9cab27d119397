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
    late = E.parallel.edge_pipeline_parameters(m) if mode == "two_bucket" else None
    bucket = E.parallel.FlatBucket(m.parameters(), late=late)
    b = _shards(world)[rank]
    bucket.zero_grad()
    pred = m(b["x"], b["edge_index"], b["pos_enc"], b["pos_index"], b["pos_batch"], b["batch"])
    if mode == "weighted":                         # local-mean gradients, scaled around the collective
        torch.nn.functional.l1_loss(pred, b["y"].view(-1, 1)).backward()
        total = bucket.all_reduce_weighted(b["x"].size(0))
    elif mode == "two_bucket":                     # the node pipeline's bucket first (overlaps the edge tail), then the rest
        (pred - b["y"].view(-1, 1)).abs().sum().backward()
        names = {id(p): n for n, p in m.named_parameters()}
        k = [i for i, p in enumerate(bucket.params) if id(p) in {id(q) for q in late}]
        assert k and k == list(range(k[0], len(bucket.params))), "late parameters must form the tail of the bucket"
        assert all(names[id(p)].split(".")[0] in ("z_initial", "z_embedding") or ".lin." in names[id(p)] for p in late)
        assert 0 < bucket.early_numel < bucket.numel and bucket.early_numel == bucket.offsets[k[0]]
        before = bucket.flat_grad[bucket.early_numel:].clone()
        bucket.all_reduce_early()
        assert torch.equal(bucket.flat_grad[bucket.early_numel:], before)      # the late bucket has not moved yet
        total = bucket.all_reduce_late(b["x"].size(0))
        bucket.flat_grad.div_(total)
    else:                                          # sum-gradients; the division belongs to the optimiser launch
        (pred - b["y"].view(-1, 1)).abs().sum().backward()
        total = bucket.all_reduce_sum(b["x"].size(0))
        bucket.flat_grad.div_(total)               # what esc_adam_step_scaled does per element
    # parameters / gradients are views into the (64-byte aligned, padded) flat buffers: compare them unpadded
    # (in model order, whatever the bucket layout)
    grads = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    params = torch.cat([p.data.reshape(-1) for p in m.parameters()])
    assert all(p.grad.data_ptr() == bucket.flat_grad[o:].data_ptr() and o % bucket.ALIGN == 0
               for p, o in zip(bucket.params, bucket.offsets))
    # BatchNorm buffers drift apart under per-shard statistics; broadcast_buffers re-aligns them before eval / checkpoints
    bufs_before = torch.cat([v.reshape(-1).float() for v in m.buffers()])
    E.parallel.broadcast_buffers(m, 0)
    bufs_after = torch.cat([v.reshape(-1).float() for v in m.buffers()])
    q.put((rank, grads.detach().numpy().copy(), float(total), params.detach().numpy().copy(),
           bufs_before.numpy().copy(), bufs_after.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["weighted", "sum", "two_bucket"])
def test_two_rank_gradient_equals_sharded_objective(mode):
    world, port = 2, 29000 + (os.getpid() + {"weighted": 0, "sum": 7, "two_bucket": 13}[mode]) % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    bufs = [(torch.from_numpy(x[4]), torch.from_numpy(x[5])) for x in res]
    res = [(x[0], torch.from_numpy(x[1]), x[2], torch.from_numpy(x[3])) for x in res]
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
    assert not torch.equal(bufs[0][0], bufs[1][0])           # per-shard running statistics differ ...
    assert torch.equal(bufs[0][1], bufs[1][1]) and torch.equal(bufs[0][1], bufs[0][0])   # ... until rank 0's are broadcast


def test_shard_slice_covers_batch():
    import esc_gnn_amd as E
    for B in (128, 7, 3):
        for W in (1, 2, 4, 8):
            parts = [E.parallel.shard_slice(B, r, W) for r in range(W)]
            assert parts[0][0] == 0 and parts[-1][1] == B
            assert all(parts[i][1] == parts[i + 1][0] for i in range(W - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def test_edge_balanced_slices_cover_the_batch_and_balance_the_edge_count():
    """SURVEY 8e "optionally balance by sum E": contiguous slices cut where the running edge count crosses k/W of the total"""
    import random
    import esc_gnn_amd as E
    random.seed(3)
    for _ in range(300):
        n, W = random.randint(1, 60), random.randint(1, 8)
        w = [random.choice([70, 96, 105, 180]) for _ in range(n)]
        parts = [E.parallel.shard_slice_balanced(w, r, W) for r in range(W)]
        assert parts[0][0] == 0 and parts[-1][1] == n and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        if n >= W:
            assert all(hi > lo for lo, hi in parts)
    w = [180] * 16 + [70] * 112                                 # 16 big graphs first: equal COUNTS would give rank 0 1.9x the mean
    even = [sum(w[lo:hi]) for lo, hi in (E.parallel.shard_slice(len(w), r, 8) for r in range(8))]
    bal = [sum(w[lo:hi]) for lo, hi in (E.parallel.shard_slice_balanced(w, r, 8) for r in range(8))]
    assert max(bal) <= 1.15 * sum(w) / 8 < max(even)

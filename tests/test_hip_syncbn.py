"""SyncBN under graph-sharded data parallelism: two ranks, each with half of the graphs of a batch and
BatchNorm statistics all-reduced over the ranks, reproduce the single-process forward and gradients of the full batch.
Both ranks share the one GPU of the test box (gloo carries the collectives)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, require_gpu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup():
    import esc_gnn_amd as E
    from esc_gnn_amd.datasets import build_count_dataset
    graphs = build_count_dataset(0, 12, h=2, use_rd=True, self_loop=True)
    gen = torch.Generator().manual_seed(3)
    for g in graphs:
        g.x = torch.randn(g.x.shape, generator=gen)
        g.y = torch.randn(g.x.size(0), generator=gen)
    store = E.DeviceGraphStore(graphs, DEV)
    torch.manual_seed(5)
    model = E.NestedGIN_eff(None, 2, 32, use_rd=True, graph_pred=False, dropout=0, edge_nest=True, use_cycle=True).to(DEV)
    return E, store, model


def _worker(rank, world, port, q, path="autograd"):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    E, store, model = _setup()
    E.nn.BatchNorm1d.convert_sync(model)
    model.train()
    ids = torch.arange(12)
    lo, hi = E.parallel.shard_slice(12, rank, world)
    b = store.collate(ids[lo:hi])
    n_glob = int(store.h_node_ptr[12])
    if path == "engine":                                     # the whole-step engine exchanges the statistics itself
        assert E.engine.engine_supports(model)
        eng = E.StepEngine(model)
        loss, pred = eng.train_step(b, loss_denom=n_glob, return_pred=True)
    elif path == "engine_node":                              # model(batch) as ONE autograd node on the engine
        model.engine_forward = True
        pred = model(b)
        assert type(pred.grad_fn).__name__.startswith("_EngineNode")
        loss = E.ops.l1_loss(pred, b.y, denom=n_glob)
        loss.backward()
    else:
        model.engine_forward = False
        pred = model(b)
        loss = E.ops.l1_loss(pred, b.y, denom=n_glob)            # this rank's share of the global mean
        loss.backward()
    grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    dist.all_reduce(grads)
    tot = loss.detach().clone()
    dist.all_reduce(tot)
    bufs = torch.cat([v.reshape(-1).float() for k, v in model.named_buffers() if "num_batches" not in k])
    q.put((rank, pred.detach().cpu().numpy(), grads.cpu().numpy(), float(tot), bufs.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("path", ["autograd", "engine", "engine_node"])
def test_two_rank_syncbn_equals_full_batch(path):
    """per-op autograd path (ops.sync_batch_norm_act), the whole-step engine (esc_engine_set_collective: 13 forward + 13
    backward exchanges per step issued from C++ through the provider) and the engine as an autograd node."""
    require_gpu()
    world, port = 2, 29600 + (os.getpid() + {"autograd": 0, "engine": 11, "engine_node": 23}[path]) % 300
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, path)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    E, store, model = _setup()
    model.train()
    b = store.collate(torch.arange(12))
    pred = model(b)
    loss = E.ops.l1_loss(pred, b.y)
    loss.backward()
    want_g = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu()
    want_b = torch.cat([v.reshape(-1).float() for k, v in model.named_buffers() if "num_batches" not in k]).cpu()
    got_pred = torch.cat([torch.from_numpy(r[1]) for r in res])
    assert torch.allclose(got_pred, pred.detach().cpu(), rtol=1e-5, atol=1e-5)
    for r in res:
        assert abs(r[3] - float(loss.detach())) <= 1e-5 * max(1.0, abs(float(loss.detach())))
        g = torch.from_numpy(r[2])
        assert float((g - want_g).norm()) <= 1e-4 * float(want_g.norm()) + 1e-6
        assert float((g - want_g).abs().max()) <= 1e-4 * max(1.0, float(want_g.abs().max()))
        assert torch.allclose(torch.from_numpy(r[4]), want_b, rtol=1e-5, atol=1e-5)
    assert (res[0][2] == res[1][2]).all()


def _mol_worker(rank, world, port, q, kind):
    """ZINC / OGB step engines with SyncBN (r03): every BatchNorm of the engine — node rows, edge rows, per-graph rows of the
    readout / virtual-node MLPs, 2H-wide hidden layers — exchanges its statistics through the collective provider"""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_hip_dp_engines import _setup as setup_engine
    E, store, model, make, late, targets = setup_engine(kind)
    E.nn.BatchNorm1d.convert_sync(model)
    model.train()
    G = len(store)
    lo, hi = E.parallel.shard_slice(G, rank, world)
    b = store.collate(torch.arange(G)[lo:hi])
    eng = make(model)
    n_glob = G if kind == "zinc" else G * model.num_tasks
    loss, pred = eng.train_step(b, loss_denom=n_glob, return_pred=True)
    grads = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in model.parameters()])
    dist.all_reduce(grads)
    tot = loss.detach().clone()
    dist.all_reduce(tot)
    bufs = torch.cat([v.reshape(-1).float() for k, v in model.named_buffers() if "num_batches" not in k])
    q.put((rank, pred.detach().cpu().numpy(), grads.cpu().numpy(), float(tot), bufs.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["zinc", "ogb"])
def test_two_rank_syncbn_molecule_engines_equal_full_batch(kind):
    """configs 4 / 5 "data-parallel over 8xMI355X" with statistics over all ranks: two ranks with half of the graphs each
    reproduce the single-process step over the full batch (predictions, loss, every gradient, running statistics)"""
    require_gpu()
    world, port = 2, 29700 + (os.getpid() + {"zinc": 5, "ogb": 29}[kind]) % 250
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mol_worker, args=(r, world, port, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path[:0] = [os.path.join(ROOT, "tests")]
    from test_hip_dp_engines import _setup as setup_engine
    E, store, model, make, late, targets = setup_engine(kind)
    eng = make(model)
    b = store.collate(torch.arange(len(store)))
    loss, pred = eng.train_step(b, return_pred=True)
    want_g = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in model.parameters()]).cpu()
    want_b = torch.cat([v.reshape(-1).float() for k, v in model.named_buffers() if "num_batches" not in k]).cpu()
    got_pred = torch.cat([torch.from_numpy(r[1]) for r in res])
    assert torch.allclose(got_pred, pred.detach().cpu(), rtol=1e-5, atol=1e-5)
    for r in res:
        assert abs(r[3] - float(loss.detach())) <= 1e-5 * max(1.0, abs(float(loss.detach())))
        g = torch.from_numpy(r[2])
        assert float((g - want_g).norm()) <= 1e-4 * float(want_g.norm()) + 1e-6
        assert float((g - want_g).abs().max()) <= 1e-4 * max(1.0, float(want_g.abs().max()))
        assert torch.allclose(torch.from_numpy(r[4]), want_b, rtol=1e-5, atol=1e-5)
    assert (res[0][2] == res[1][2]).all()

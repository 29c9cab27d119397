"""Graph-sharded data parallelism of the three step engines on two ranks (both share the test box's one GPU, gloo carries
the collectives): every rank runs its shard with sum-form gradients (loss_denom=1), the flat bucket is all-reduced — in
two buckets for the counting model, like bench.py and run_graphcount — and the division by the global target count rides
on the Adam launch.  Checked against ONE process that runs the two shards one after the other and adds the gradients:
identical update, identical parameters on both ranks afterwards.  (BatchNorm statistics are per shard: plain DP.)"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_collate, require_gpu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(kind):
    import esc_gnn_amd as E
    torch.manual_seed(7)
    if kind == "count":
        from esc_gnn_amd.datasets import build_count_dataset
        graphs = build_count_dataset(0, 12, h=2, use_rd=True, self_loop=True)
        gen = torch.Generator().manual_seed(3)
        for g in graphs:
            g.x = torch.randn(g.x.shape, generator=gen)
            g.y = torch.randn(g.x.size(0), generator=gen)
        model = E.NestedGIN_eff(None, 2, 32, use_rd=True, graph_pred=False, dropout=0, edge_nest=True, use_cycle=True)
        make = lambda m: E.StepEngine(m)
        late = E.parallel.edge_pipeline_parameters
        targets = lambda b: b.x.size(0)
    elif kind == "zinc":
        from esc_gnn_amd.zinc_models import NestedGIN_eff as ZincModel
        from esc_gnn_amd.engine import ZincStepEngine
        gs, _, _ = load_collate("zinc3")
        graphs = [E.Data(**{k: torch.tensor(v) for k, v in g.items()}) for g in gs] * 2
        model = ZincModel(None, 2)
        make, late, targets = (lambda m: ZincStepEngine(m)), (lambda m: None), (lambda b: b.num_graphs)
    else:
        from esc_gnn_amd.ogb_mol_gnn import GNN
        from esc_gnn_amd.engine import OgbStepEngine
        gs, _, _ = load_collate("molhiv4")
        graphs = [E.Data(**{k: torch.tensor(v) for k, v in g.items()}) for g in gs] * 2
        model = GNN("ogbg-molhiv", 1, num_layer=2, emb_dim=32, gnn_type="gin_eff", virtual_node=True, residual=True, drop_ratio=0.0)
        with torch.no_grad():
            model.gnn_node.virtualnode_embedding.weight.normal_(0, 0.1)
        make, late, targets = (lambda m: OgbStepEngine(m)), (lambda m: None), (lambda b: b.num_graphs)
    store = E.DeviceGraphStore(graphs, DEV)
    return E, store, model.to(DEV).train(), make, late, targets


def _flat(model):
    return torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()


def _worker(rank, world, port, q, kind):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    E, store, model, make, late, targets = _setup(kind)
    G = len(store)
    opt = E.optim.FlatAdam(model.parameters(), lr=1e-2, late=late(model))
    eng = make(model)
    lo, hi = E.parallel.shard_slice(G, rank, world)
    b = store.collate(torch.arange(G)[lo:hi])
    if kind == "count":                                      # the two-bucket exchange of bench.py / run_graphcount
        eng.begin_step(b, loss_denom=1)
        opt.all_reduce_early()
        eng.end_step()
        opt.step(grad_denom=opt.all_reduce_late(targets(b)))
    else:
        eng.train_step(b, loss_denom=1)
        opt.step(grad_denom=opt.all_reduce_sum(targets(b)))
    torch.cuda.synchronize()
    q.put((rank, _flat(model)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["count", "zinc", "ogb"])
def test_two_rank_data_parallel_step_equals_the_sum_of_its_shards(kind):
    require_gpu()
    world, port = 2, 29100 + (os.getpid() + {"count": 0, "zinc": 17, "ogb": 41}[kind]) % 300
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][1], res[1][1])                     # identical replicas after the step
    # one process: the two shards one after the other, sum-form gradients added, one Adam step on sum / count
    E, store, model, make, late, targets = _setup(kind)
    G = len(store)
    opt = E.optim.FlatAdam(model.parameters(), lr=1e-2, late=late(model))
    eng = make(model)
    total, count = torch.zeros_like(opt.flat_grad), 0
    saved = {k: v.clone() for k, v in model.state_dict().items()}
    for r in range(world):
        lo, hi = E.parallel.shard_slice(G, r, world)
        b = store.collate(torch.arange(G)[lo:hi])
        model.load_state_dict(saved)                                 # (each rank starts from the same BatchNorm buffers)
        eng.train_step(b, loss_denom=1)
        total += opt.flat_grad
        count += targets(b)
    model.load_state_dict(saved)
    opt.flat_grad.copy_(total)
    opt.step(grad_denom=torch.tensor([float(count)], device=DEV))
    want = _flat(model)
    err = np.abs(res[0][1] - want).max()
    assert err <= 2e-6 * max(1.0, np.abs(want).max()), err

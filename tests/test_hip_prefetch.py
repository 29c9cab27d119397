"""harness.prefetched: the next batch is collated (and its embedding plans built) on a side stream while the current one
trains.  The loop must produce exactly the numbers of the plain loop — a race between the side stream's collate and the
training stream (a batch read before it is complete, or a block reused while a kernel still reads it) shows as a
difference — and must hand out the same batches in the same order."""
import pytest
import torch

from conftest import require_gpu


def _run(kind, use_prefetch, steps=14):
    import esc_gnn_amd as E
    from esc_gnn_amd.datasets import build_feature_dataset, synthetic_ogbmol_graphs, synthetic_zinc_graphs
    from esc_gnn_amd.harness import prefetched
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    if kind == "ogb":
        from esc_gnn_amd.engine import OgbStepEngine
        from esc_gnn_amd.ogb_mol_gnn import GNN
        graphs = build_feature_dataset(synthetic_ogbmol_graphs(0, 112), 2, use_rd=True, self_loop=True)
        model = GNN("ogbg-molhiv", 1, num_layer=2, emb_dim=32, gnn_type="gin_eff", virtual_node=True, residual=True,
                    drop_ratio=0.3, use_rd=True).to(dev).train()
        eng = OgbStepEngine(model)
    else:
        from esc_gnn_amd.engine import ZincStepEngine
        from esc_gnn_amd.zinc_models import NestedGIN_eff
        graphs = build_feature_dataset(synthetic_zinc_graphs(0, 112), 2, use_rd=True, self_loop=False)
        model = NestedGIN_eff(None, num_layers=2).to(dev).train()
        eng = ZincStepEngine(model)
    store = E.DeviceGraphStore(graphs, dev)
    opt = E.optim.FlatAdam(model.parameters(), lr=1e-3)      # AFTER the engine: re-homes parameters and gradients (see below)
    bs = 16
    ids = [torch.arange(i * bs, (i + 1) * bs) % len(store) for i in range(steps)]
    seen, losses = [], []
    batches = (store.collate(i) for i in ids)
    if use_prefetch:
        batches = prefetched(batches, dev, eng.prepare)
    for b in batches:
        seen.append(int(b.edge_index.size(1)))
        losses.append(eng.train_step(b))
        opt.step()
    torch.cuda.synchronize()
    return seen, torch.stack(losses).cpu(), [p.detach().cpu().clone() for p in model.parameters()]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["ogb", "zinc"])
def test_prefetched_loop_equals_plain_loop(kind):
    require_gpu()
    from esc_gnn_amd import _native as nv
    nv.call("esc_engine_set_two_stream_min_edges", 0)          # the engine's own second stream on, as at full size
    try:
        seen_a, loss_a, par_a = _run(kind, False)
        seen_b, loss_b, par_b = _run(kind, True)
    finally:
        nv.call("esc_engine_set_two_stream_min_edges", 12000)
    assert seen_a == seen_b
    assert torch.equal(loss_a, loss_b), (loss_a - loss_b).abs().max()
    for a, b in zip(par_a, par_b):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["count", "ogb", "zinc"])
def test_step_engine_follows_rehomed_parameters(kind):
    """A step engine holds raw device addresses.  FlatAdam (built after the engine) moves every parameter and gradient
    into its bucket, zero_grad(set_to_none=True) drops the gradients: the next step must notice and re-read the
    addresses instead of reading freed weights and scattering gradients over freed memory (found as a hang of the
    bag backward whose plan had been overwritten).  Checked against an engine built after the optimiser."""
    require_gpu()
    import copy
    import esc_gnn_amd as E
    from esc_gnn_amd.datasets import build_feature_dataset, synthetic_ogbmol_graphs, synthetic_zinc_graphs
    dev = torch.device("cuda:0")
    torch.manual_seed(9)
    if kind == "ogb":
        from esc_gnn_amd.engine import OgbStepEngine as Eng
        from esc_gnn_amd.ogb_mol_gnn import GNN
        graphs = build_feature_dataset(synthetic_ogbmol_graphs(0, 32), 2, use_rd=True, self_loop=True)
        model = GNN("ogbg-molhiv", 1, num_layer=2, emb_dim=32, gnn_type="gin_eff", virtual_node=True, residual=True,
                    drop_ratio=0.0).to(dev).train()
    elif kind == "zinc":
        from esc_gnn_amd.engine import ZincStepEngine as Eng
        from esc_gnn_amd.zinc_models import NestedGIN_eff
        graphs = build_feature_dataset(synthetic_zinc_graphs(0, 32), 2, use_rd=True, self_loop=False)
        model = NestedGIN_eff(None, num_layers=2).to(dev).train()
    else:
        from esc_gnn_amd.engine import StepEngine as Eng
        from esc_gnn_amd.datasets import build_count_dataset
        graphs = build_count_dataset(0, 32, h=2)
        model = E.NestedGIN_eff(None, 2, 64, use_rd=True, graph_pred=False, dropout=0, edge_nest=True, use_cycle=True).to(dev).train()
    store = E.DeviceGraphStore(graphs, dev)
    ref = copy.deepcopy(model)
    eng = Eng(model)                                            # addresses of the module's own storage ...
    opt = E.optim.FlatAdam(model.parameters(), lr=1e-3)         # ... which FlatAdam now frees
    opt_ref = E.optim.FlatAdam(ref.parameters(), lr=1e-3)
    eng_ref = Eng(ref)
    for i in range(3):
        b = store.collate(torch.arange(16) + 16 * (i % 2))
        la = eng.train_step(b)
        lb = eng_ref.train_step(store.collate(torch.arange(16) + 16 * (i % 2)))
        assert torch.equal(la, lb), (i, float(la), float(lb))
        opt.step(); opt_ref.step()
        if i == 1:
            opt.zero_grad(set_to_none=True); opt_ref.zero_grad(set_to_none=True)
    for a, c in zip(model.parameters(), ref.parameters()):
        assert torch.equal(a, c)


def test_prefetched_passes_through_on_cpu_and_stops_cleanly():
    """no device: plain iteration, `warm` still applied; an empty producer ends at once"""
    from esc_gnn_amd.harness import prefetched
    got = list(prefetched(iter([1, 2, 3]), "cpu", warm=lambda x: None))
    assert got == [1, 2, 3]
    assert list(prefetched(iter([]), "cpu")) == []

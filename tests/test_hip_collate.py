"""Device collate (DeviceGraphStore.collate -> csrc/collate.hip) against the reference's batch.py
outputs (tests/golden/collate_*.npz): every reference-visible tensor bit-identical, and the compact
plan identical to the one derived from the batch tensors."""
import numpy as np
import pytest
import torch

from conftest import load_collate, require_gpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    require_gpu()
    import esc_gnn_amd
    return esc_gnn_amd


def _store(E, tag):
    graphs, ref, ng = load_collate(tag)
    datas = [E.Data(**{k: torch.tensor(v) for k, v in g.items()}) for g in graphs]
    return E.DeviceGraphStore(datas, "cuda:0"), datas, ref, ng


@pytest.mark.parametrize("tag", ["count3", "mixed4"])
def test_device_collate_matches_reference(E, tag):
    store, datas, ref, ng = _store(E, tag)
    b = store.collate(list(range(len(datas))))
    assert sorted(b.keys) == sorted(ref.keys())
    for k, v in ref.items():
        got = b[k].cpu()
        assert got.dtype == torch.tensor(v).dtype, k
        assert np.array_equal(got.numpy(), v), k
    assert b.num_graphs == ng
    plan = b.__dict__["_esc_plan"]
    want = E.BatchPlan.from_tensors(b.edge_index, b.x.size(0), b.pos_enc, b.pos_index, b.pos_batch)
    for f in E.BatchPlan.FIELDS:
        assert torch.equal(getattr(plan, f), getattr(want, f)), f


def test_subset_permutation_and_repeats(E):
    store, datas, _, _ = _store(E, "mixed4")
    for ids in ([2, 0], [3], [1, 1, 2], [3, 2, 1, 0]):
        b = store.collate(torch.tensor(ids))
        want = E.Batch.from_data_list([datas[i] for i in ids])
        for k in want.keys:
            assert torch.equal(b[k].cpu(), want[k]), (ids, k)
        plan = b.__dict__["_esc_plan"]
        ref = E.BatchPlan.from_tensors(b.edge_index, b.x.size(0), b.pos_enc, b.pos_index, b.pos_batch)
        for f in E.BatchPlan.FIELDS:
            assert torch.equal(getattr(plan, f), getattr(ref, f)), (ids, f)
    with pytest.raises(IndexError):
        store.collate([0, 9])
    with pytest.raises(ValueError):
        store.collate([])


def test_loader_and_model_on_collated_batch(E):
    store, datas, _, _ = _store(E, "mixed4")
    sizes = [b.num_graphs for b in E.DeviceLoader(store, batch_size=3)]
    assert sizes == [3, 1]
    torch.manual_seed(0)
    m = E.NestedGIN_eff(None, 2, 32, use_rd=True, graph_pred=False, dropout=0, edge_nest=True, use_cycle=True).to("cuda:0")
    b = store.collate([0, 1, 2, 3])
    host = E.Batch.from_data_list(datas)
    out_dev = m(b)
    out_host = m(host)                 # foreign batch: plan derived on the fly
    assert torch.equal(out_dev, out_host)


def test_store_cache_round_trip(E, tmp_path):
    import os
    store, datas, ref, _ = _store(E, "count3")
    path = os.path.join(tmp_path, "data_tr.pt")
    store.save(path)
    again = E.DeviceGraphStore.load(path, "cuda:0")
    a, b = store.collate([0, 1, 2]), again.collate([0, 1, 2])
    for k in a.keys:
        assert torch.equal(a[k], b[k]), k
    pa, pb = a.__dict__["_esc_plan"], b.__dict__["_esc_plan"]
    for f in E.BatchPlan.FIELDS:
        assert torch.equal(getattr(pa, f), getattr(pb, f)), f

"""Device collate (DeviceGraphStore.collate -> csrc/collate.hip) against the reference's batch.py
outputs (tests/golden/collate_*.npz): every reference-visible tensor bit-identical, and the compact
plan identical to the one derived from the batch tensors."""
import numpy as np
import pytest
import torch

from conftest import load_collate, require_gpu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def E():
    require_gpu()
    import esc_gnn_amd
    return esc_gnn_amd


def _store(E, tag):
    graphs, ref, ng = load_collate(tag)
    datas = [E.Data(**{k: torch.tensor(v) for k, v in g.items()}) for g in graphs]
    return E.DeviceGraphStore(datas, "cuda:0"), datas, ref, ng


@pytest.mark.parametrize("tag", ["count3", "mixed4", "zinc3", "molhiv4"])
def test_device_collate_matches_reference(E, tag):
    """every tensor of the reference's batch.py output (batch.py:25-149), bit for bit and dtype for dtype: fp32 x of the
    counting data, int64 categorical x (ZINC 1-D, molhiv [n, 9]) and the gathered edge_attr rows (1-D / [E, 3])"""
    store, datas, ref, ng = _store(E, tag)
    b = store.collate(list(range(len(datas))))
    assert sorted(b.keys) == sorted(ref.keys())
    for k, v in ref.items():
        got = b[k].cpu()
        assert got.dtype == torch.tensor(v).dtype, k
        assert tuple(got.shape) == tuple(v.shape), k
        assert np.array_equal(got.numpy(), v), k
    assert b.num_graphs == ng
    plan = b.__dict__["_esc_plan"]
    want = E.BatchPlan.from_tensors(b.edge_index, b.x.size(0), b.pos_enc, b.pos_index, b.pos_batch)
    for f in E.BatchPlan.FIELDS:
        assert torch.equal(getattr(plan, f), getattr(want, f)), f
    # graph pointers of the fill kernel = node ranges of the reference's `batch` vector
    gp = plan.graph_ptr.cpu().numpy()
    assert plan.graph_ptr.dtype == torch.int32 and plan.num_graphs == ng
    assert np.array_equal(gp, np.concatenate([[0], np.cumsum(np.bincount(ref["batch"], minlength=ng))]))
    # ... and the host collate agrees on the same goldens (same keys, same dtypes)
    host = E.Batch.from_data_list(datas)
    for k in ref:
        assert torch.equal(host[k], b[k].cpu()), k


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
        # the per-graph node pointers the fill kernel leaves for the readout pooling
        assert plan.graph_ptr.dtype == torch.int32 and plan.num_graphs == len(ids)
        assert plan.graph_ptr.cpu().tolist() == [0] + np.cumsum([datas[i].x.size(0) for i in ids]).tolist()
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


@pytest.mark.parametrize("n,n_keys,seed", [(0, 5, 0), (1, 1, 1), (63, 7, 2), (64, 300, 3), (2049, 255, 4), (15200, 2400, 5),
                                          (550_000, 1800, 6), (70_001, 70_000, 7)])
def test_plan_csr_kernel_is_a_stable_sort(E, n, n_keys, seed):
    """esc_plan_csr (csrc/plan.hip: radix passes over the positions) against a stable torch.sort + bincount + cumsum on
    the host: segment pointers and permutation bit-exact, for 1-, 2- and 3-pass key ranges, empty / one-element inputs,
    chunk boundaries; out-of-range keys are reported."""
    from esc_gnn_amd.plan import _csr
    g = torch.Generator().manual_seed(seed)
    key = torch.randint(0, n_keys, (n,), generator=g)
    if n > 10:
        key[: n // 3] = key[0]                     # a long run of one key: stability matters
    ptr, perm = _csr(key.to(DEV), n_keys)
    want_perm = torch.sort(key, stable=True)[1]
    want_ptr = torch.zeros(n_keys + 1, dtype=torch.int64)
    want_ptr[1:] = torch.cumsum(torch.bincount(key, minlength=n_keys), 0)
    assert ptr.dtype == torch.int32 and perm.dtype == torch.int32
    assert torch.equal(ptr.cpu().long(), want_ptr)
    assert torch.equal(perm.cpu().long(), want_perm)
    ptr2, none = _csr(key.to(DEV), n_keys, want_perm=False)
    assert none is None and torch.equal(ptr2, ptr)
    if n:
        bad = key.clone(); bad[n // 2] = n_keys
        with pytest.raises(IndexError):
            _csr(bad.to(DEV), n_keys)


def test_plan_cache_follows_the_index_tensors(E):
    """plan_of() caches the CSR/CSC plan on the Data object; replacing or editing edge_index (edge dropout, augmentation,
    a re-used Batch) must rebuild it instead of aggregating over the stale one."""
    _, b, _ = load_collate("count3")
    data = E.Data(**{k: torch.tensor(v) for k, v in b.items()}).to(DEV)
    p1 = E.plan_of(data)
    assert E.plan_of(data) is p1                                   # cached
    keep = torch.arange(0, data.edge_index.size(1), 2, device=DEV)
    data.edge_index = data.edge_index[:, keep]                      # assignment of a new tensor
    sel = torch.isin(data.pos_batch, keep)
    remap = torch.full((int(keep.max()) + 1,), -1, device=DEV, dtype=torch.long); remap[keep] = torch.arange(keep.numel(), device=DEV)
    data.pos_enc, data.pos_index, data.pos_batch = data.pos_enc[sel], data.pos_index[sel], remap[data.pos_batch[sel]]
    p2 = E.plan_of(data)
    assert p2 is not p1 and p2.num_edges == keep.numel() and p2.nnz == int(sel.sum())
    data.edge_index[0, 0] = data.edge_index[0, 1]                   # in-place edit: the version counter moves
    p3 = E.plan_of(data)
    assert p3 is not p2
    src = data.edge_index[0][p3.out_edge.long()]
    assert bool((src[1:] >= src[:-1]).all())


def test_model_forward_rebuilds_a_stale_plan(E):
    """NestedGIN_eff.forward calls data.to(device) BEFORE plan_of (run_graphcount.py:134-135): after edge dropout the cached
    plan must be rebuilt by that call chain, not stamped valid by Data.to() (it indexes the old E and Z: wrong sums or
    out-of-range int32 reads)."""
    _, b, _ = load_collate("count3")
    torch.manual_seed(0)
    model = E.NestedGIN_eff(None, 2, 32, use_rd=True, graph_pred=False, dropout=0, edge_nest=True, use_cycle=True).to(DEV).eval()
    data = E.Data(**{k: torch.tensor(v) for k, v in b.items()}).to(DEV)
    with torch.no_grad():
        model(data)
    p1 = data.__dict__["_esc_plan"]
    keep = torch.arange(0, data.edge_index.size(1), 2, device=DEV)
    sel = torch.isin(data.pos_batch, keep)
    remap = torch.full((int(keep.max()) + 1,), -1, device=DEV, dtype=torch.long); remap[keep] = torch.arange(keep.numel(), device=DEV)
    data.edge_index = data.edge_index[:, keep]
    data.pos_enc, data.pos_index, data.pos_batch = data.pos_enc[sel], data.pos_index[sel], remap[data.pos_batch[sel]]
    with torch.no_grad():
        got = model(data)
    p2 = data.__dict__["_esc_plan"]
    assert p2 is not p1 and p2.num_edges == keep.numel() and p2.nnz == int(sel.sum())
    fresh = E.Data(**{k: data[k].clone() for k in data.keys})
    with torch.no_grad():
        want = model(fresh)
    assert torch.equal(got, want)


def test_segment_pointers_beyond_the_radix_key_range(E):
    """esc_plan_csr sorts on 24 key bits; pointer-only calls (perm == NULL: keys already grouped, e.g. row_ptr keyed on the
    edge count) have no such limit"""
    from esc_gnn_amd.plan import _csr
    n_keys = (1 << 24) + 1000
    key = torch.sort(torch.randint(0, n_keys, (5000,), generator=torch.Generator().manual_seed(1)))[0]
    ptr, none = _csr(key.to(DEV), n_keys, want_perm=False)
    want = torch.zeros(n_keys + 1, dtype=torch.int64)
    want[1:] = torch.cumsum(torch.bincount(key, minlength=n_keys), 0)
    assert none is None and torch.equal(ptr.cpu().long(), want)
    with pytest.raises(RuntimeError):
        _csr(key.to(DEV), n_keys, want_perm=True)


@pytest.mark.parametrize("tag", ["mixed4", "molhiv4", "zinc3"])
def test_dataloader_pins_the_dataset_and_yields_the_same_batches(E, tag):
    """the reference's loop `for data in DataLoader(dataset, bs, shuffle=True): data = data.to(device)`
    (dataloader.py:11-48, run_graphcount.py:453-455,487): with a HIP device the loader collates on the device from a
    pinned copy — same graphs in the same (seeded) order, every tensor bit-identical to the host collate moved over"""
    graphs, _, _ = load_collate(tag)
    datas = [E.Data(**{k: torch.tensor(v) for k, v in g.items()}) for g in graphs] * 3
    fast = E.DataLoader(datas, batch_size=5, shuffle=True)
    host = E.DataLoader(datas, batch_size=5, shuffle=True, device=None)
    for epoch in range(2):
        torch.manual_seed(100 + epoch)
        got = list(fast)
        torch.manual_seed(100 + epoch)
        want = list(host)
        assert len(got) == len(want) == -(-len(datas) // 5) and fast.__dict__["_esc_store"] is not False
        for a, b in zip(got, want):
            assert sorted(a.keys) == sorted(b.keys) and a.num_graphs == b.num_graphs
            for k in b.keys:
                assert a[k].is_cuda and not b[k].is_cuda
                assert a[k].dtype == b[k].dtype and torch.equal(a[k].cpu(), b[k]), (epoch, k)
            assert a.to(DEV) is a and "_esc_plan" in a.__dict__          # the caller's .to(device) keeps the collate's plan
            back = a.to_data_list()
            assert len(back) == a.num_graphs
    # what the device store cannot reproduce stays on the reference's host path
    assert E.DataLoader(datas, batch_size=5, follow_batch=["pos_enc"]).__iter__().__class__.__name__ != "generator"
    odd = [E.Data(x=torch.ones(2, 1), edge_index=torch.tensor([[0], [1]]), mask=torch.tensor([True, False]))] * 4
    assert not next(iter(E.DataLoader(odd, batch_size=2))).x.is_cuda


def test_reference_data_slices_layout_round_trip(E, tmp_path):
    """the reference's InMemoryDataset cache is `torch.save((data, slices), path)` (GraphCountDataset.py:119-120): the
    store writes that layout as plain tensors (loadable with weights_only=True), reads it back — from a dict or from an
    object with attributes, as a PyG Data would be — and collates the same batches."""
    import os
    import types
    store, datas, ref, _ = _store(E, "mixed4")
    data, slices = store.to_data_slices()
    assert data["edge_index"].shape[0] == 2 and int(slices["x"][-1]) == data["x"].size(0)
    for g, d in enumerate(datas):                                   # per-graph pieces are the original graphs
        a, b = int(slices["edge_index"][g]), int(slices["edge_index"][g + 1])
        assert torch.equal(data["edge_index"][:, a:b], d.edge_index)
        za, zb = int(slices["pos_batch"][g]), int(slices["pos_batch"][g + 1])
        assert torch.equal(data["pos_batch"][za:zb], d.pos_batch)
    path = os.path.join(tmp_path, "data_tr.pt")
    torch.save((data, slices), path)
    d2, s2 = torch.load(path, weights_only=True)
    for src in (d2, types.SimpleNamespace(**d2)):
        again = E.DeviceGraphStore.from_data_slices(src, s2, "cuda:0")
        a, b = store.collate([0, 1, 2, 3]), again.collate([0, 1, 2, 3])
        for k in a.keys:
            assert torch.equal(a[k], b[k]), k

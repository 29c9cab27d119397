"""Host-side collate (esc_gnn_amd.Batch.from_data_list) against the outputs of the reference's
batch.py recorded in tests/golden/collate_*.npz."""
import numpy as np
import pytest
import torch

from conftest import load_collate
import esc_gnn_amd as E


def _to_data(g):
    return E.Data(**{k: torch.tensor(v) for k, v in g.items()})


@pytest.mark.parametrize("tag", ["count3", "mixed4", "zinc3", "molhiv4"])
def test_from_data_list_matches_reference(tag):
    graphs, ref, num_graphs = load_collate(tag)
    b = E.Batch.from_data_list([_to_data(g) for g in graphs])
    assert sorted(b.keys) == sorted(ref.keys())
    for k, v in ref.items():
        got = b[k]
        assert got.dtype == torch.tensor(v).dtype, k
        assert np.array_equal(got.numpy(), v), k
    assert b.num_graphs == num_graphs
    assert b.pos_batch.is_contiguous() and b.edge_index.is_contiguous()


def test_round_trip_and_follow_batch():
    graphs, ref, _ = load_collate("count3")
    datas = [_to_data(g) for g in graphs]
    b = E.Batch.from_data_list(datas, follow_batch=["pos_enc"])
    assert b["pos_enc_batch"].shape == b.pos_enc.shape
    back = b.to_data_list()
    assert len(back) == len(datas)
    for d0, d1 in zip(datas, back):
        for k in d0.keys:
            assert torch.equal(d0[k], d1[k]), k


def test_rules_bool_and_batch_key():
    a = E.Data(x=torch.ones(2, 1), edge_index=torch.tensor([[0], [1]]), mask=torch.tensor([True, False]))
    c = E.Data(x=torch.ones(3, 1), edge_index=torch.tensor([[2], [0]]), mask=torch.tensor([True, True, False]))
    b = E.Batch.from_data_list([a, c])
    assert b.mask.dtype == torch.bool and b.mask.tolist() == [True, False, True, True, False]
    assert b.edge_index.tolist() == [[0, 4], [1, 2]]
    assert b.batch.tolist() == [0, 0, 1, 1, 1]
    with pytest.raises(AssertionError):
        E.Batch.from_data_list([E.Data(x=torch.ones(1, 1), batch=torch.zeros(1, dtype=torch.long))])


def test_dataloader_batches():
    graphs, _, _ = load_collate("mixed4")
    datas = [_to_data(g) for g in graphs]
    out = list(E.DataLoader(datas, batch_size=3))
    assert [b.num_graphs for b in out] == [3, 1]


def test_to_does_not_revalidate_a_stale_plan():
    """Data.to() used to re-key the cached plan unconditionally: after edge dropout (a new edge_index) the stale plan was
    stamped valid and model(data) -- which calls data.to(device) before plan_of -- aggregated over the old E.  The
    validity is now judged BEFORE the move (reference call order: run_graphcount.py:134-135 `data.to`, then the layers)."""
    from esc_gnn_amd.plan import BatchPlan, plan_key, _sig
    graphs, _, _ = load_collate("count3")
    b = E.Batch.from_data_list([_to_data(g) for g in graphs])
    plan = BatchPlan(num_nodes=b.x.size(0), num_edges=b.edge_index.size(1), nnz=b.pos_enc.numel())
    plan._key = plan_key(b, plan.n_cols)
    plan.graph_ptr, plan.num_graphs, plan._batch_sig = torch.zeros(4, dtype=torch.int32), 3, _sig(b.batch)
    object.__setattr__(b, "_esc_plan", plan)
    b.to("cpu")                                           # unchanged tensors: the plan stays, re-keyed
    assert b.__dict__["_esc_plan"] is plan and plan._key == plan_key(b, plan.n_cols) and plan.graph_ptr is not None
    b.batch = b.batch.clone()                             # a re-assigned batch vector: graph bounds must be rebuilt
    b.to("cpu")
    assert b.__dict__["_esc_plan"] is plan and plan.graph_ptr is None and plan._batch_sig is None
    b.edge_index = b.edge_index[:, ::2].contiguous()      # edge dropout: a new edge_index
    assert plan._key != plan_key(b, plan.n_cols)
    b.to("cpu")
    assert "_esc_plan" not in b.__dict__                  # dropped, not stamped valid
    # in-place edits move the version counter and are caught the same way
    plan._key = plan_key(b, plan.n_cols)
    object.__setattr__(b, "_esc_plan", plan)
    b.pos_enc.add_(1)
    b.to("cpu")
    assert "_esc_plan" not in b.__dict__

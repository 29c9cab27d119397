"""HIP feature builder (create_subgraphs) against (i) every golden vector recorded from the
reference's own create_subgraphs and (ii) the oracle on fresh random graphs — bit-exact int64."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, require_gpu
import graph_sources as gs
import ref_features as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    require_gpu()
    import esc_gnn_amd
    return esc_gnn_amd


def _cmp(enc, want, name):
    ei, _, pe, pi, pb = enc
    assert ei.dtype == pe.dtype == pi.dtype == pb.dtype == torch.int64
    for got, key in ((ei[0], "out_src"), (ei[1], "out_dst"), (pe, "pos_enc"), (pi, "pos_index"), (pb, "pos_batch")):
        assert np.array_equal(got.numpy(), want[key]), (name, key)


def test_all_reference_goldens_batched(E, feature_cases):
    from esc_gnn_amd.utils_edge_efficient import encode_edge_lists
    groups = {}
    for c in feature_cases:
        groups.setdefault((c["h"], c["use_rd"], c["self_loop"]), []).append(c)
    assert len(groups) >= 6
    for (h, rd, sl), cases in groups.items():
        encs = encode_edge_lists([c["n"] for c in cases],
                                 [torch.tensor(np.stack([c["in_src"], c["in_dst"]])) for c in cases], h, rd, sl)
        for c, enc in zip(cases, encs):
            _cmp(enc, c, c["name"])


def test_single_graph_api_and_edge_attr(E):
    z = np.load(os.path.join(GOLDEN, "features_edge_attr.npz"))
    n = int(z["n"])
    for tag in ("ea1", "ea2"):
        d = E.Data(x=torch.ones(n, 3), edge_index=torch.tensor(np.stack([z["src"], z["dst"]])),
                   edge_attr=torch.tensor(z[tag + "_in"]), y=torch.zeros(1))
        o = E.create_subgraphs(d, 2, use_rd=True, self_loop=True)
        assert np.array_equal(o.edge_attr.numpy(), z[tag + "_out"])
        assert np.array_equal(o.edge_index.numpy(), np.stack([z["out_src"], z["out_dst"]]))
        assert sorted(o.keys) == ["edge_attr", "edge_index", "pos_batch", "pos_enc", "pos_index", "x", "y"]
        assert torch.equal(o.x, d.x)
    # h given as a list: only the last one survives (reference :41,:152)
    o2 = E.create_subgraphs(d, [1, 2], use_rd=True, self_loop=True)
    assert torch.equal(o2.pos_enc, o.pos_enc) and torch.equal(o2.pos_index, o.pos_index)


@pytest.mark.parametrize("cfg", [(3, True, True), (4, True, True), (3, True, False), (3, False, True), (2, True, True)])
def test_fresh_graphs_vs_oracle(E, cfg):
    from esc_gnn_amd.utils_edge_efficient import encode_edge_lists
    h, rd, sl = cfg
    graphs = [gs.count_shape_graph(g) for g in range(40, 52)] + [gs.molecule_like_graph(s) for s in range(50, 62)]
    graphs += [gs.random_directed_graph(s, 6 + s % 7, 12 + 3 * (s % 5)) for s in range(20, 30)]
    graphs.append((1, np.zeros(0, np.int64), np.zeros(0, np.int64)))            # single node, no edges
    graphs.append((3, np.zeros(0, np.int64), np.zeros(0, np.int64)))            # edgeless
    encs = encode_edge_lists([g[0] for g in graphs], [torch.tensor(np.stack([g[1], g[2]])) for g in graphs], h, rd, sl)
    for i, ((n, s, t), enc) in enumerate(zip(graphs, encs)):
        want = orc.encode_graph(s, t, n, h, rd, sl)
        _cmp(enc, dict(out_src=want["edge_src"], out_dst=want["edge_dst"], pos_enc=want["pos_enc"],
                       pos_index=want["pos_index"], pos_batch=want["pos_batch"]), "fresh%d" % i)


def test_larger_graph_and_errors(E):
    from esc_gnn_amd.utils_edge_efficient import encode_edge_lists
    # 70-node sparse graph (n > one wave of nodes), no rd
    n, s, t = gs.molecule_like_graph(7, 70, 70)
    enc = encode_edge_lists([n], [torch.tensor(np.stack([s, t]))], 3, False, True)[0]
    want = orc.encode_graph(s, t, n, 3, False, True)
    _cmp(enc, dict(out_src=want["edge_src"], out_dst=want["edge_dst"], pos_enc=want["pos_enc"],
                   pos_index=want["pos_index"], pos_batch=want["pos_batch"]), "mol70")
    enc = encode_edge_lists([n], [torch.tensor(np.stack([s, t]))], 3, True, True)[0]
    want = orc.encode_graph(s, t, n, 3, True, True)
    _cmp(enc, dict(out_src=want["edge_src"], out_dst=want["edge_dst"], pos_enc=want["pos_enc"],
                   pos_index=want["pos_index"], pos_batch=want["pos_batch"]), "mol70rd")
    # hub of degree 209 -> the reference's one_hot(num_classes=200) raises
    n = 210
    s = np.concatenate([np.zeros(n - 1, np.int64), np.arange(1, n)]); t = np.concatenate([np.arange(1, n), np.zeros(n - 1, np.int64)])
    with pytest.raises(RuntimeError):
        encode_edge_lists([n], [torch.tensor(np.stack([s, t]))], 1, False, False)
    with pytest.raises(RuntimeError):
        encode_edge_lists([2], [torch.tensor([[0, 5], [1, 0]])], 1, False, False)      # node id out of range
    with pytest.raises(RuntimeError):
        encode_edge_lists([2], [torch.tensor([[0], [1]])], 5, False, False)            # h > 4
    with pytest.raises(NotImplementedError):
        E.create_subgraphs(E.Data(x=torch.ones(2, 1), edge_index=torch.tensor([[0], [1]])), 1, max_nodes_per_hop=3)

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


def _hub_ring(n):
    """node 0 joined to everyone, the others on a ring: every 2-hop ego-net is the whole graph"""
    pairs = [(0, i) for i in range(1, n)] + [(i, i + 1) for i in range(1, n - 1)] + [(n - 1, 1)]
    s = np.array([a for a, b in pairs] + [b for a, b in pairs], np.int64)
    t = np.array([b for a, b in pairs] + [a for a, b in pairs], np.int64)
    return n, s, t


def _regular_like(n, seed):
    """ring + two random perfect matchings (degree <= 4): 3-hop ego-nets of 50..100 nodes, many distinct ones"""
    rng = np.random.default_rng(seed)
    pairs = {(i, (i + 1) % n) for i in range(n)}
    for _ in range(2):
        p = rng.permutation(n)
        pairs |= {(int(min(a, b)), int(max(a, b))) for a, b in zip(p[0::2], p[1::2]) if a != b}
    pairs = sorted((min(a, b), max(a, b)) for a, b in pairs)
    s = np.array([a for a, b in pairs] + [b for a, b in pairs], np.int64)
    t = np.array([b for a, b in pairs] + [a for a, b in pairs], np.int64)
    return n, s, t


@pytest.mark.parametrize("case", ["hub130", "regular104_h3", "regular104_h4", "mol150", "mixed_batch"])
def test_ego_nets_beyond_lds_with_rd(E, case):
    """use_rd on ego-nets of more than 96 nodes (ogbg-molhiv holds 222-atom molecules): the pseudo-inverse leaves LDS for
    a global-memory slab; graphs of > 96 nodes whose ego-nets stay small keep the LDS buckets.  Bit-exact vs the oracle."""
    from esc_gnn_amd.utils_edge_efficient import encode_edge_lists
    if case == "hub130":
        graphs, h, sl = [_hub_ring(130)], 2, True
    elif case == "regular104_h3":
        graphs, h, sl = [_regular_like(104, 1)], 3, False
    elif case == "regular104_h4":
        graphs, h, sl = [_regular_like(104, 2)], 4, True
    elif case == "mol150":
        graphs, h, sl = [gs.molecule_like_graph(3, 150, 150)], 4, True
    else:
        graphs, h, sl = [gs.count_shape_graph(41), _hub_ring(100), gs.molecule_like_graph(5), _regular_like(98, 3)], 3, True
    encs = encode_edge_lists([g[0] for g in graphs], [torch.tensor(np.stack([g[1], g[2]])) for g in graphs], h, True, sl)
    for i, ((n, s, t), enc) in enumerate(zip(graphs, encs)):
        want = orc.encode_graph(s, t, n, h, True, sl)
        _cmp(enc, dict(out_src=want["edge_src"], out_dst=want["edge_dst"], pos_enc=want["pos_enc"],
                       pos_index=want["pos_index"], pos_batch=want["pos_batch"]), "%s/%d" % (case, i))


def test_many_graphs_chunking_by_hop_table_budget(E):
    """create_subgraphs_many ends a chunk when the per-root hop tables (sum of n^2) would exceed the budget: same
    result whatever the chunking"""
    graphs = [gs.count_shape_graph(g) for g in range(60, 72)] + [gs.molecule_like_graph(s) for s in range(70, 76)]
    datas = [E.Data(x=torch.ones(n, 1), edge_index=torch.tensor(np.stack([s, t])), y=torch.zeros(1)) for n, s, t in graphs]
    whole = E.create_subgraphs_many(datas, 3, use_rd=True, self_loop=True)
    pieces = E.create_subgraphs_many(datas, 3, use_rd=True, self_loop=True, table_budget=900)     # ~1-2 graphs per chunk
    tiny = E.create_subgraphs_many(datas, 3, use_rd=True, self_loop=True, chunk=5)
    for a, b, c in zip(whole, pieces, tiny):
        for k in ("edge_index", "pos_enc", "pos_index", "pos_batch"):
            assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]), k

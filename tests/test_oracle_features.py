"""The oracle restatement (oracle/ref_features.py) against every golden vector produced by the
reference's own create_subgraphs (tests/golden/features_*.npz; generator oracle/make_golden.py)."""
import numpy as np
import pytest

import ref_features as orc


def test_oracle_matches_reference_goldens(feature_cases):
    assert len(feature_cases) >= 100
    for c in feature_cases:
        got = orc.encode_graph(c["in_src"], c["in_dst"], c["n"], c["h"], c["use_rd"], c["self_loop"])
        for mine, ref in (("edge_src", "out_src"), ("edge_dst", "out_dst"), ("pos_enc", "pos_enc"),
                          ("pos_index", "pos_index"), ("pos_batch", "pos_batch")):
            assert np.array_equal(got[mine], c[ref]), (c["name"], mine)


def test_width_and_offsets():
    # no-rd layout is 1700 wide with edge codes at 400 (reference :122-138)
    s = np.array([0, 1, 1, 2]); t = np.array([1, 0, 2, 1])
    a = orc.encode_graph(s, t, 3, 2, False, False)
    b = orc.encode_graph(s, t, 3, 2, True, False)
    assert a["pos_index"].max() < 1700 and b["pos_index"].max() < 1800
    assert (a["pos_index"][a["pos_index"] >= 400] + 100).tolist() == b["pos_index"][b["pos_index"] >= 500].tolist()


def test_phantom_root_on_self_loop_edge():
    # 6-path, edge (0,0): rd bins {0:5} — 4 reached nodes + the phantom copy
    s = np.array([0, 1, 1, 2, 2, 3, 3, 4, 4, 5]); t = np.array([1, 0, 2, 1, 3, 2, 4, 3, 5, 4])
    o = orc.encode_graph(s, t, 6, 3, True, True)
    k = int(np.flatnonzero((o["edge_src"] == 0) & (o["edge_dst"] == 0))[0])
    sel = o["pos_batch"] == k
    idx, val = o["pos_index"][sel], o["pos_enc"][sel]
    rd = {int(i) - 400: int(v) for i, v in zip(idx, val) if 400 <= i < 500}
    assert rd == {0: 5}


def test_overflow_raises():
    n = 210
    s = np.concatenate([np.zeros(n - 1, int), np.arange(1, n)]); t = np.concatenate([np.arange(1, n), np.zeros(n - 1, int)])
    with pytest.raises(RuntimeError):
        orc.encode_graph(s, t, n, 1, False, False)     # hub degree 209 >= 200

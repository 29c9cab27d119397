"""ORACLE (test infrastructure, not product code): CPU restatement of the
reference's edge-rooted h-hop ego-net structural encoding.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module; the product path (esc-gnn_amd/) never does.

Follows /root/reference/utils_edge_efficient.py:
  * create_subgraphs            :20-152   (driver, self-loop normalisation :33-36)
  * k_hop_subgraph              :201-294  (BFS target->source :210,:222-243; induced edge mask :283-285)
  * ego-net assembly            :52-67    (node order, union of two induced edge masks :55, hop pairs :56-61)
  * sub-degree                  :86       (out-degree over relabelled sub-edges, loops counted)
  * resistance distance         :92-107   (scipy laplacian -> dense -> scipy.linalg.pinv -> fp32)
  * sparse histogram encoding   :122-144

Pinned against outputs of the reference itself (tests/golden/*.npz, produced by
oracle/make_golden.py importing the unmodified reference under oracle/pyg_shim).

Plain numpy + scipy; python loops over edges -> use on small graphs only.
"""
import numpy as np
from scipy import linalg as _sla

Z_WIDTH_RD = 1800
Z_WIDTH_NO_RD = 1700


def normalise_self_loops(src, dst, n):
    """utils_edge_efficient.py:33-36 — drop every (a,a), append (i,i) i=0..n-1 at the END."""
    keep = src != dst
    loops = np.arange(n, dtype=np.int64)
    return (np.concatenate([src[keep], loops]), np.concatenate([dst[keep], loops]), keep)


def hop_table(src, dst, n, h):
    """D[r, x] = BFS depth of x from root r moving target->source (a step goes from t to s
    for every edge s->t), or h+1 if x is not reached within h hops.
    utils_edge_efficient.py:206-243 (col,row = edge_index; frontier mask on row, new = col[mask]).
    One BFS per root node suffices (observation i)."""
    far = h + 1
    D = np.full((n, n), far, dtype=np.int64)
    # step[t] = list of s with an edge s->t  (self loops are harmless: s already visited)
    step = [[] for _ in range(n)]
    for s, t in zip(src.tolist(), dst.tolist()):
        step[t].append(s)
    for r in range(n):
        D[r, r] = 0
        frontier = [r]
        for depth in range(1, h + 1):
            nxt = []
            for t in frontier:
                for s in step[t]:
                    if D[r, s] == far:
                        D[r, s] = depth
                        nxt.append(s)
            if not nxt:
                break
            frontier = nxt
    return D


def _pinv_rd(nodes_order, sub_src, sub_dst, phantom):
    """Resistance-distance column to local node 0 — utils_edge_efficient.py:92-107.
    Adjacency of ones over relabelled sub-edges (duplicates add up in coo->csr), scipy's
    csgraph.laplacian (diag = column sums minus diagonal, so self loops vanish), dense fp64
    scipy.linalg.pinv, rd = Lxx + Lyy - Lxy - Lyx, then fp32 rounding (:106 FloatTensor)."""
    m = len(nodes_order) + (1 if phantom else 0)
    local = {}
    base = 1 if phantom else 0  # phantom copy of the root is local index 0 (:52-54,:66)
    for i, v in enumerate(nodes_order):
        local[v] = i + base
    A = np.zeros((m, m), dtype=np.float64)
    for a, b in zip(sub_src, sub_dst):
        A[local[a], local[b]] += 1.0
    w = A.sum(axis=0) - np.diag(A)
    L = -A
    L[np.arange(m), np.arange(m)] = w
    P = _sla.pinv(L)
    d = np.diag(P)
    rd = P[0, 0] + d - P[0, :] - P[:, 0]
    return rd.astype(np.float32)


def encode_graph(src, dst, n, h, use_rd, self_loop):
    """Restatement of create_subgraphs for one graph.

    Returns dict(edge_src, edge_dst, kept (mask of surviving input edges or None),
                 pos_enc, pos_index, pos_batch) — all int64 like the reference."""
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    kept = None
    if self_loop:
        src, dst, kept = normalise_self_loops(src, dst, n)
    E = src.shape[0]
    width = Z_WIDTH_RD if use_rd else Z_WIDTH_NO_RD
    off_edge = 500 if use_rd else 400
    D = hop_table(src, dst, n, h)
    far = h + 1
    reach = D <= h
    vals, idxs, segs = [], [], []
    nonloop = src != dst
    for k in range(E):
        u, v = int(src[k]), int(dst[k])
        in_u, in_v = reach[u], reach[v]
        emask = (in_u[src] & in_u[dst]) | (in_v[src] & in_v[dst])       # :55 union of two induced sets
        s_src, s_dst = src[emask], dst[emask]
        nodes = np.flatnonzero(in_u | in_v)
        deg = np.bincount(s_src, minlength=n)                           # :86 out-degree, loops count
        z0, z1 = D[u], D[v]
        hist = np.zeros(width, dtype=np.int64)
        np.add.at(hist, deg[nodes], 1)                                  # :129 one_hot(sub_degree, 200)
        np.add.at(hist, 200 + z0[nodes], 1)
        np.add.at(hist, 300 + z1[nodes], 1)
        phantom = u == v
        if phantom:                                                     # duplicate root: isolated, z=(0,0)
            hist[0] += 1
            hist[200] += 1
            hist[300] += 1
        if use_rd:
            order = [u] + [x for x in nodes.tolist() if x != u]         # root first; rest any order (:287 quirk list)
            rd = _pinv_rd(order, s_src.tolist(), s_dst.tolist(), phantom)
            bins = rd.astype(np.int64)                                  # .long(): trunc toward zero (:131)
            if bins.min() < 0 or bins.max() >= 100:
                raise RuntimeError("one_hot overflow: resistance-distance bin outside [0,100)")
            np.add.at(hist, 400 + bins, 1)
        if deg[nodes].max(initial=0) >= 200:
            raise RuntimeError("one_hot overflow: sub-degree >= 200")
        nl = emask & nonloop                                            # :137 remove_self_loops
        a, b = src[nl], dst[nl]
        code = 216 * z0[a] + 36 * z1[a] + 6 * z0[b] + z1[b]
        if code.size and code.max() >= 1300:
            raise RuntimeError("one_hot overflow: edge code >= 1300 (needs h <= 4)")
        np.add.at(hist, off_edge + code, 1)
        nz = np.flatnonzero(hist)                                       # ascending (:140 torch.nonzero)
        vals.append(hist[nz])
        idxs.append(nz)
        segs.append(np.full(nz.shape[0], k, dtype=np.int64))
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, dtype=np.int64)
    return dict(edge_src=src, edge_dst=dst, kept=kept,
                pos_enc=cat(vals), pos_index=cat(idxs), pos_batch=cat(segs))

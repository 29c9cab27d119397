"""ORACLE / TEST INFRASTRUCTURE: deterministic graph sources shared by the golden
generator, the tests and bench.py's synthetic workload description.

No reference code here.  `count_shape_graph` is the synthetic "count_cycle shape"
generator of SURVEY.md §8(d); `read_g6`/`read_exp_txt` parse the two graph files the
reference ships (data/sr25/raw/sr251256.g6, data/EXP/GRAPHSAT.txt) — they are used only
inside the build container by make_golden.py (the GPU box has no /root/reference).
"""
import numpy as np

COUNT_SHAPE_MIX = ((10, 6), (15, 6), (20, 5), (30, 5))


def _undirected_to_directed(pairs, n):
    """Both directions, sorted row-major, duplicates removed (the order np.where(A==1) gives,
    GraphCountDataset.py:72, and also what PyG's to_undirected+coalesce yields, SRDataset.py:37)."""
    A = np.zeros((n, n), dtype=bool)
    for a, b in pairs:
        A[a, b] = True
        A[b, a] = True
    s, d = np.where(A)
    return s.astype(np.int64), d.astype(np.int64)


def count_shape_graph(g):
    """Graph g of the synthetic count_cycle-shaped dataset: random d-regular on n nodes,
    (n,d) cycling through COUNT_SHAPE_MIX, networkx seed = g."""
    import networkx as nx
    n, d = COUNT_SHAPE_MIX[g % 4]
    G = nx.random_regular_graph(d, n, seed=g)
    s, t = _undirected_to_directed(list(G.edges()), n)
    return n, s, t


def triangle_counts(n, s, t):
    """Per-node triangle count diag(A^3)/2 — the synthetic regression target."""
    A = np.zeros((n, n), dtype=np.int64)
    A[s, t] = 1
    return (np.diagonal(A @ A @ A) // 2).astype(np.float32)


def molecule_like_graph(seed, n_lo=18, n_hi=30):
    """Random tree on n in [n_lo,n_hi] nodes plus 1-3 ring-closing edges (ZINC / molhiv-like
    topology)."""
    rng = np.random.RandomState(seed)
    n = int(rng.randint(n_lo, n_hi + 1))
    pairs = [(i, int(rng.randint(0, i))) for i in range(1, n)]
    have = set(map(tuple, map(sorted, pairs)))
    for _ in range(int(rng.randint(1, 4))):
        for _try in range(20):
            a, b = map(int, rng.randint(0, n, size=2))
            if a != b and tuple(sorted((a, b))) not in have:
                have.add(tuple(sorted((a, b))))
                pairs.append((a, b))
                break
    s, t = _undirected_to_directed(pairs, n)
    return n, s, t


def random_directed_graph(seed, n, m, loops=True):
    """Asymmetric edge list with optional pre-existing self loops and duplicates allowed off."""
    rng = np.random.RandomState(seed)
    seen, s, t = set(), [], []
    while len(s) < m:
        a, b = map(int, rng.randint(0, n, size=2))
        if (a == b and not loops) or (a, b) in seen:
            continue
        seen.add((a, b))
        s.append(a)
        t.append(b)
    return n, np.array(s, dtype=np.int64), np.array(t, dtype=np.int64)


def hand_cases():
    """Small named edge cases: single edge, triangle, path, star, isolated node,
    pre-existing self loops, directed cycle."""
    und = _undirected_to_directed
    out = {}
    out["single_edge"] = (2,) + und([(0, 1)], 2)
    out["triangle"] = (3,) + und([(0, 1), (1, 2), (0, 2)], 3)
    out["path6"] = (6,) + und([(i, i + 1) for i in range(5)], 6)
    out["star5"] = (6,) + und([(0, i) for i in range(1, 6)], 6)
    out["isolated"] = (5,) + und([(0, 1), (1, 2), (2, 0)], 5)          # nodes 3,4 isolated
    s, t = und([(0, 1), (1, 2), (2, 3)], 4)
    out["with_loops"] = (4, np.concatenate([s, [1, 3]]), np.concatenate([t, [1, 3]]))
    out["dicycle"] = (5, np.arange(5, dtype=np.int64), (np.arange(5, dtype=np.int64) + 1) % 5)
    out["two_triangles"] = (5,) + und([(0, 1), (1, 2), (0, 2), (2, 3), (3, 4), (2, 4)], 5)
    return out


def read_g6(path):
    import networkx as nx
    out = []
    for G in nx.read_graph6(path):
        n = G.number_of_nodes()
        out.append((n,) + _undirected_to_directed(list(G.edges()), n))
    return out


def read_exp_txt(path, limit):
    """EXP text format: line 1 = #graphs; per graph 'n label', then n lines
    'node_label deg nbr_1 ... nbr_deg' (SURVEY §8c)."""
    out = []
    with open(path) as f:
        total = int(f.readline())
        for _ in range(min(total, limit)):
            n, _label = map(int, f.readline().split())
            pairs = []
            for i in range(n):
                parts = list(map(int, f.readline().split()))
                for nb in parts[2:2 + parts[1]]:
                    pairs.append((i, nb))
            out.append((n,) + _undirected_to_directed(pairs, n))
    return out

"""Datasets for the counting benchmark.

The reference reads data/count_cycle/raw/data.mat through GraphCountDataset.py:97-120 and applies
create_subgraphs as pre_transform.  The raw .mat files are not shipped (.MISSING_LARGE_BLOBS), so the
default here is the deterministic synthetic "count_cycle shape" generator of SURVEY.md §8(d):
graph g is a random d-regular graph on n nodes, (n,d) = [(10,6),(15,6),(20,5),(30,5)][g % 4],
networkx seed g; edges in np.where(A==1) order (GraphCountDataset.py:72); x = ones[n,10] (:84);
y = per-node triangle counts.  `load_count_mat` reads a real data.mat when one is present.
"""
import numpy as np
import torch

from .data import Data
from .utils_edge_efficient import create_subgraphs_many

COUNT_SHAPE_MIX = ((10, 6), (15, 6), (20, 5), (30, 5))


def count_shape_adjacency(g):
    import networkx as nx
    n, d = COUNT_SHAPE_MIX[g % 4]
    G = nx.random_regular_graph(d, n, seed=g)
    A = np.zeros((n, n), dtype=np.float32)
    for a, b in G.edges():
        A[a, b] = A[b, a] = 1.0
    return A


def adjacency_to_data(A, y):
    """GraphCountDataset.adj2data (:69-84): edge order of np.where(A == 1), x = ones[n,10]."""
    begin, end = np.where(A == 1.0)
    n = A.shape[0]
    return Data(x=torch.ones(n, 10), edge_index=torch.tensor(np.stack([begin, end]).astype(np.int64)),
                y=torch.as_tensor(y), num_nodes=n)


def synthetic_count_graphs(first, count):
    out = []
    for g in range(first, first + count):
        A = count_shape_adjacency(g)
        tri = (np.diagonal(A @ A @ A) / 2.0).astype(np.float32)
        out.append(adjacency_to_data(A, tri))
    return out


def build_count_dataset(first, count, h=3, use_rd=True, self_loop=True):
    """Synthetic graphs + ESC features (HIP feature builder), as run_graphcount.py:404-408 configures it."""
    raw = synthetic_count_graphs(first, count)
    done = create_subgraphs_many(raw, h, use_rd=use_rd, self_loop=self_loop)
    for d in done:
        d.num_nodes = None            # like the reference's new Data: num_nodes is inferred from x
    return done


def load_count_mat(path, split="train", target=0):
    """Read the benchmark's data.mat (reference GraphCountDataset.process :97-111)."""
    import scipy.io as scio
    raw = scio.loadmat(path)
    idx = {"train": "train_idx", "val": "val_idx", "test": "test_idx"}[split]
    if raw["F"].shape[0] == 1:
        ids = raw[idx][0]
        pairs = [(raw["A"][0][i], raw["F"][0][i]) for i in ids]
    else:
        pairs = list(zip(raw["A"][0][raw[idx]][0], raw["F"][raw[idx]][0]))
    out = []
    for A, y in pairs:
        y = np.asarray(y)
        if y.ndim == 1:
            y = y.reshape(1, -1)
        d = adjacency_to_data(np.asarray(A, dtype=np.float32), y)
        out.append(d)
    return out


# ---------------------------------------------------------------------------------------------------------
# Molecule-shaped synthetic sets: ZINC.pkl and the OGB downloads are absent offline, so the ZINC /
# OGB drivers default to seeded trees-with-rings of the same size and feature layout.
# ---------------------------------------------------------------------------------------------------------
def molecule_like_edges(seed, n_lo=18, n_hi=30):
    """Random tree on n in [n_lo, n_hi] nodes + 1-3 ring-closing edges; both directions, sorted by (src, dst)."""
    rng = np.random.RandomState(seed)
    n = int(rng.randint(n_lo, n_hi + 1))
    und = {(int(rng.randint(0, i)), i) for i in range(1, n)}
    rings = 0
    for _ in range(int(rng.randint(1, 4))):
        for _try in range(20):
            a, b = sorted(map(int, rng.randint(0, n, size=2)))
            if a != b and (a, b) not in und:
                und.add((a, b))
                rings += 1
                break
    both = sorted(und | {(b, a) for a, b in und})
    ei = np.array(both, dtype=np.int64).T
    return n, ei, rings, rng


def synthetic_zinc_graphs(first, count):
    """ZINC layout (dataset_zinc.py:56-72): x int64[n] atom type in [0,28), edge_attr int64[E] bond type in [0,4),
    y float[1] (a smooth function of the topology so that training has signal)."""
    out = []
    for g in range(first, first + count):
        n, ei, rings, rng = molecule_like_edges(1000 + g)
        x = torch.tensor(rng.randint(0, 28, size=n))
        bond = rng.randint(0, 4, size=ei.shape[1])
        key = {}
        for k in range(ei.shape[1]):                       # same bond type in both directions
            a, b = int(ei[0, k]), int(ei[1, k])
            bond[k] = key.setdefault((min(a, b), max(a, b)), bond[k])
        deg = np.bincount(ei[0], minlength=n)
        y = float(rings) + 0.25 * float((deg >= 3).sum()) + 0.05 * float(x.float().mean())
        out.append(Data(x=x, edge_index=torch.tensor(ei), edge_attr=torch.tensor(bond), y=torch.tensor([y]),
                        num_nodes=n))
    return out


def synthetic_ogbmol_graphs(first, count, num_tasks=1, nan_ratio=0.0):
    """ogbg-mol* layout: x int64[n,9] (AtomEncoder columns), edge_attr int64[E,3] (BondEncoder columns),
    y float[1,num_tasks] in {0,1} with optional NaN (unlabeled) entries as in ogbg-molpcba."""
    from .ogb_mol_gnn import ATOM_FEATURE_DIMS, BOND_FEATURE_DIMS
    out = []
    for g in range(first, first + count):
        n, ei, rings, rng = molecule_like_edges(5000 + g, 12, 40)
        x = np.stack([rng.randint(0, d, size=n) for d in ATOM_FEATURE_DIMS], axis=1)
        ea = np.stack([rng.randint(0, d, size=ei.shape[1]) for d in BOND_FEATURE_DIMS], axis=1)
        score = rings + (x[:, 0] % 7 == 0).sum() * 0.5
        y = np.array([[float((score + t) % 3 >= 1.5) for t in range(num_tasks)]], dtype=np.float32)
        if nan_ratio > 0:
            y[0, rng.rand(num_tasks) < nan_ratio] = np.nan
        out.append(Data(x=torch.tensor(x), edge_index=torch.tensor(ei), edge_attr=torch.tensor(ea), y=torch.tensor(y),
                        num_nodes=n))
    return out


def build_feature_dataset(raw, h, use_rd=True, self_loop=False):
    done = create_subgraphs_many(raw, h, use_rd=use_rd, self_loop=self_loop)
    for d in done:
        d.num_nodes = None
    return done

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

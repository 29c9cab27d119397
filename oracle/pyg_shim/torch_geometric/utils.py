"""Shim `torch_geometric.utils` helpers (own restatement of PyG semantics)."""
import numpy as np
import scipy.sparse as ssp
import torch


def remove_self_loops(edge_index, edge_attr=None):
    keep = edge_index[0] != edge_index[1]
    edge_index = edge_index[:, keep]
    if edge_attr is not None:
        edge_attr = edge_attr[keep]
    return edge_index, edge_attr


def add_self_loops(edge_index, edge_attr=None, fill_value=1.0, num_nodes=None):
    n = int(edge_index.max()) + 1 if num_nodes is None else int(num_nodes)
    loops = torch.arange(n, dtype=edge_index.dtype, device=edge_index.device)
    loops = loops.unsqueeze(0).repeat(2, 1)
    if edge_attr is not None:
        fill = edge_attr.new_full((n,) + tuple(edge_attr.shape[1:]), fill_value)
        edge_attr = torch.cat([edge_attr, fill], dim=0)
    return torch.cat([edge_index, loops], dim=1), edge_attr


def add_remaining_self_loops(edge_index, edge_attr=None, fill_value=1.0, num_nodes=None):
    raise NotImplementedError("unused by the efficient path")


def degree(index, num_nodes=None, dtype=None):
    n = int(index.max()) + 1 if num_nodes is None else int(num_nodes)
    out = torch.zeros(n, dtype=dtype if dtype is not None else torch.float)
    return out.scatter_add_(0, index, torch.ones_like(index, dtype=out.dtype))


def to_scipy_sparse_matrix(edge_index, edge_attr=None, num_nodes=None):
    row, col = edge_index.cpu().numpy()
    n = int(edge_index.max()) + 1 if num_nodes is None else int(num_nodes)
    vals = np.ones(row.shape[0]) if edge_attr is None else edge_attr.cpu().numpy()
    return ssp.coo_matrix((vals, (row, col)), shape=(n, n))


def to_dense_batch(*a, **k):
    raise NotImplementedError("unused by the efficient path")

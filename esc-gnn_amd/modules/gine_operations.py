"""GINE+ multi-hop convolutions — the MI355X twin of the aggregate primitive of
/root/reference/modules/gine_operations.py:306-362 (NAIVEGINEPLUS, GINEPLUS):

    result = (1 + eps[0]) * x_0 + sum_{d=1..k} (1 + eps[d]) * sum_{edges with distance d} relu(x_j (+ e if d == 1))

Each per-distance neighbour sum runs through the HIP segmented gather-reduce (csrc/aggregate.hip, no self
term, optional edge term).  The multi-hop edge construction (make_multihop_edges :256-303, torch_sparse SpGEMM)
is out of scope: `multihop_edge_index` and `distance` are inputs here, as in the
reference's forward."""
import torch
from torch import nn

from .. import ops
from ..plan import BatchPlan


def _plans(multihop_edge_index, distance, k, num_nodes):
    """One execution plan per distance class (esc_plan_csr, csrc/plan.hip).  They depend on the batch only, while every
    GINE+ layer of the model asks for them: cached on the edge tensor, keyed on the identity (address, shape, in-place
    version) of both inputs, so that L layers x forward/backward build them once per batch."""
    from ..plan import _sig
    key = (_sig(multihop_edge_index), _sig(distance), int(k), int(num_nodes))
    cached = getattr(multihop_edge_index, "_esc_gineplus_plans", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    out = []
    for d in range(1, k + 1):
        ei = multihop_edge_index[:, distance == d]
        out.append(BatchPlan.from_tensors(ei, num_nodes))
    multihop_edge_index._esc_gineplus_plans = (key, out)
    return out


class NAIVEGINEPLUS(nn.Module):
    def __init__(self, fun, dim, k=4, **kwargs):
        super().__init__()
        self.k = k
        self.nn = fun
        self.eps = nn.Parameter(torch.zeros(k + 1, dim), requires_grad=True)

    def forward(self, x, multihop_edge_index, distance, edge_attr):
        assert x.size(-1) == edge_attr.size(-1)
        plans = _plans(multihop_edge_index, distance, self.k, x.size(0))
        result = (1 + self.eps[0]) * x
        for i in range(self.k):
            out = ops.neighbour_sum(x, edge_attr if i == 0 else None, plans[i])
            result = result + (1 + self.eps[i + 1]) * out
        return self.nn(result)

    def __repr__(self):
        return "{}(nn={}, k={})".format(self.__class__.__name__, self.nn, self.eps.size(0))


class GINEPLUS(nn.Module):
    def __init__(self, fun, dim, k=4, **kwargs):
        super().__init__()
        self.k = k
        self.nn = fun
        self.eps = nn.Parameter(torch.zeros(k + 1, dim), requires_grad=True)

    def forward(self, XX, multihop_edge_index, distance, edge_attr):
        """XX is the list of previous xs, XX[0] being the last layer's (reference :343)."""
        assert len(XX) >= self.k
        assert XX[-1].size(-1) == edge_attr.size(-1)
        plans = _plans(multihop_edge_index, distance, self.k, XX[0].size(0))
        result = (1 + self.eps[0]) * XX[0]
        for i, x in enumerate(XX):
            if i >= self.k:
                break
            out = ops.neighbour_sum(x, edge_attr if i == 0 else None, plans[i])
            result = result + (1 + self.eps[i + 1]) * out
        return [self.nn(result)] + XX

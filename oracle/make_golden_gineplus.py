"""ORACLE / TEST INFRASTRUCTURE — golden for SURVEY §8 row a-12 (runs ONLY in the build container).

The class bodies of the reference's NAIVEGINEPLUS and GINEPLUS (/root/reference/modules/gine_operations.py:306-362) are
extracted with ast and exec'd over a stand-in `MessagePassing` whose propagate() is the PyG 2.0.4 semantics the
reference relies on (aggr='add'): x_j = x.index_select(0, edge_index[0]); message(x_j, edge_attr); scatter-add at
edge_index[1] with dim_size = x.size(0) (sequential index_add_ in edge order).  The reference's own forward / message
code then runs on seeded inputs; inputs, outputs and every gradient go to tests/golden/model_gineplus.npz.
Only data is written — no reference source."""
import ast
import os
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_FILE = "/root/reference/modules/gine_operations.py"
OUT = os.path.join(ROOT, "tests", "golden", "model_gineplus.npz")


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", **kw):
        super().__init__()
        assert aggr == "add"

    def propagate(self, edge_index, x, edge_attr=None):
        x_j = x.index_select(0, edge_index[0])
        msg = self.message(x_j, edge_attr)
        return torch.zeros_like(x).index_add_(0, edge_index[1], msg)


def reference_classes():
    tree = ast.parse(open(REF_FILE).read())
    nodes = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in ("NAIVEGINEPLUS", "GINEPLUS")]
    ns = dict(torch=torch, nn=torch.nn, F=F, nng=types.SimpleNamespace(MessagePassing=MessagePassing))
    exec(compile(ast.Module(body=nodes, type_ignores=[]), REF_FILE, "exec"), ns)
    return ns["NAIVEGINEPLUS"], ns["GINEPLUS"]


def main():
    torch.set_num_threads(1)
    torch.manual_seed(77)
    Naive, Plus = reference_classes()
    n, dim, k = 37, 24, 3
    # a multi-hop edge list: random directed pairs with a distance class 1..k (+ one class beyond k that must be ignored)
    m = 260
    src, dst = torch.randint(0, n, (m,)), torch.randint(0, n, (m,))
    distance = torch.randint(1, k + 2, (m,))
    mei = torch.stack([src, dst])
    m1 = int((distance == 1).sum())
    out = dict(multihop_edge_index=mei.numpy(), distance=distance.numpy(), k=np.int64(k))

    def mlp():
        return torch.nn.Sequential(torch.nn.Linear(dim, dim), torch.nn.ReLU(), torch.nn.Linear(dim, dim))

    for name, Cls in (("naive", Naive), ("plus", Plus)):
        conv = Cls(mlp(), dim, k=k)
        with torch.no_grad():
            conv.eps.copy_(0.3 * torch.randn(k + 1, dim))
        edge_attr = torch.randn(m1, dim, requires_grad=True)
        if name == "naive":
            x = torch.randn(n, dim, requires_grad=True)
            res = conv(x, mei, distance, edge_attr)
            xs = [x]
        else:
            xs = [torch.randn(n, dim, requires_grad=True) for _ in range(k + 1)]     # one more than k: must be passed through
            ret = conv(list(xs), mei, distance, edge_attr)
            assert len(ret) == len(xs) + 1 and all(a is b for a, b in zip(ret[1:], xs))
            res = ret[0]
        w = torch.randn(n, dim)
        (res * w).sum().backward()
        out[name + "_w"] = w.numpy()
        out[name + "_edge_attr"] = edge_attr.detach().numpy()
        out[name + "_out"] = res.detach().numpy()
        out[name + "_d_edge_attr"] = edge_attr.grad.numpy()
        for i, x in enumerate(xs):
            out["%s_x%d" % (name, i)] = x.detach().numpy()
            out["%s_dx%d" % (name, i)] = (x.grad if x.grad is not None else torch.zeros_like(x)).numpy()
        for pn, p in conv.named_parameters():
            out["%s_param_%s" % (name, pn)] = p.detach().numpy()
            out["%s_grad_%s" % (name, pn)] = p.grad.numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main()

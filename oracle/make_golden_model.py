"""ORACLE / TEST INFRASTRUCTURE — model-composition golden (runs ONLY in the build container).

run_graphcount.py cannot be imported (its module body parses argv, loads datasets and trains),
so — as SURVEY.md §8(c) describes — the *class body* of the reference's NestedGIN_eff
(run_graphcount.py:39-194) is extracted with ast and exec'd in a namespace whose PyG primitives
(GINEConv, global_add_pool, global_mean_pool) are oracle/ref_model.py's restatements.  The
reference's own layer composition then runs on a reference-collated batch; we record
  * the state_dict key/shape list,
  * the parameters (seeded init; hidden=16, layers=3 keeps the fixture small),
  * train-mode and eval-mode predictions, the L1 loss and every parameter gradient,
and check oracle/ref_model.NestedGINEffRef reproduces them bit-for-bit.
Only data is written to tests/golden/model_count.npz — no reference source.
"""
import ast
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [HERE]
import ref_model as rm  # noqa: E402

REF_FILE = "/root/reference/run_graphcount.py"
OUT = os.path.join(ROOT, "tests", "golden", "model_count.npz")


def reference_class(ref_file=REF_FILE):
    src = open(ref_file).read()
    tree = ast.parse(src)
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "NestedGIN_eff")
    ns = dict(torch=torch, F=F, ELU=torch.nn.ELU, Linear=torch.nn.Linear, Sequential=torch.nn.Sequential, ReLU=torch.nn.ReLU,
              BN=torch.nn.BatchNorm1d, Dropout=torch.nn.Dropout, GINEConv=rm.GINEConv,
              global_add_pool=rm.global_add_pool, global_mean_pool=rm.global_mean_pool)
    exec(compile(ast.Module(body=[node], type_ignores=[]), ref_file, "exec"), ns)
    return ns["NestedGIN_eff"]


class Bag(object):
    """attribute bag standing in for a PyG Batch inside the reference forward."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def to(self, device):
        return self

    def __contains__(self, key):
        return key in self.__dict__


def main():
    torch.set_num_threads(1)  # CPU index/scatter backward is run-to-run nondeterministic when threaded
    g = np.load(os.path.join(ROOT, "tests", "golden", "collate_count3.npz"))
    b = {k[len("batch_"):]: torch.tensor(g[k]) for k in g.files if k.startswith("batch_")}
    L, H = 3, 16
    torch.manual_seed(1234)
    Ref = reference_class()
    ref = Ref(None, L, H, use_rd=True, graph_pred=False, dropout=0, edge_nest=True, use_cycle=True)
    # randomise BN affine + eps so that every parameter matters in the comparison
    with torch.no_grad():
        for name, p in ref.named_parameters():
            if p.dim() == 1 and "bias" not in name:
                p.add_(0.1 * torch.randn_like(p))
    sd0 = {k: v.clone() for k, v in ref.state_dict().items()}

    mine = rm.NestedGINEffRef(L, H, graph_pred=False, dropout=0.0, use_cycle=True)
    assert list(mine.state_dict().keys()) == list(sd0.keys()), "state_dict key order differs"
    mine.load_state_dict(sd0)

    y = b["y"].view(-1, 1)
    y = (y - y.mean()) / y.std()
    out = {}

    def fwd_ref(m):
        return m(Bag(x=b["x"], edge_index=b["edge_index"], batch=b["batch"], pos_enc=b["pos_enc"],
                     pos_index=b["pos_index"], pos_batch=b["pos_batch"]))

    def fwd_mine(m):
        return m(b["x"], b["edge_index"], b["pos_enc"], b["pos_index"], b["pos_batch"], b["batch"])

    results = []
    for m, fwd in ((ref, fwd_ref), (mine, fwd_mine)):
        m.train()
        pred = fwd(m)
        loss = F.l1_loss(pred, y)
        loss.backward()
        grads = {k: p.grad.clone() for k, p in m.named_parameters()}
        sd_after = {k: v.clone() for k, v in m.state_dict().items()}
        m.eval()
        with torch.no_grad():
            pred_eval = fwd(m)
        results.append((pred.detach(), loss.detach(), grads, pred_eval, sd_after))
    (p0, l0, g0, e0, s0), (p1, l1, g1, e1, s1) = results
    assert torch.equal(p0, p1) and torch.equal(l0, l1) and torch.equal(e0, e1), "composition mismatch"
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k

    out["keys"] = np.array(list(sd0.keys()))
    for k, v in sd0.items():
        out["param/" + k] = v.numpy()
    for k, v in g0.items():
        out["grad/" + k] = v.numpy()
    for k, v in s0.items():
        if "running" in k or "num_batches" in k:
            out["after/" + k] = v.numpy()
    out["y"] = y.numpy()
    out["pred_train"] = p0.numpy()
    out["pred_eval"] = e0.numpy()
    out["loss"] = l0.numpy()
    out["layers"], out["hidden"] = np.int64(L), np.int64(H)
    np.savez_compressed(OUT, **out)
    n_par = sum(v.numel() for k, v in sd0.items())
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB;", len(sd0), "tensors,", n_par, "elements; loss", float(l0))

    # also record the full-size key/shape list (L=4, H=256) the checkpoint format must keep
    big = Ref(None, 4, 256, use_rd=True, graph_pred=False, dropout=0, edge_nest=True, use_cycle=True)
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_L4_H256.txt"), "w") as f:
        for k, v in big.state_dict().items():
            f.write("%s %s\n" % (k, "x".join(map(str, v.shape)) or "scalar"))
    print("params L4 H256:", sum(p.numel() for p in big.parameters()))


def main_sr():
    """kernel/gin.py:200-379 (the run_sr.py / run_exp.py model): graph-level readout, log_softmax head."""
    torch.set_num_threads(1)
    g = np.load(os.path.join(ROOT, "tests", "golden", "collate_mixed4.npz"))
    b = {k[len("batch_"):]: torch.tensor(g[k]) for k in g.files if k.startswith("batch_")}
    L, H, NC = 3, 16, 3

    class DS(object):
        num_features, num_classes = 10, NC
    torch.manual_seed(4321)
    Ref = reference_class("/root/reference/kernel/gin.py")
    ref = Ref(DS, L, H, use_rd=False, graph_pred=True, dropout=0, use_cycle=False)
    with torch.no_grad():
        for name, p in ref.named_parameters():
            if p.dim() == 1 and "bias" not in name:
                p.add_(0.1 * torch.randn_like(p))
    sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
    mine = rm.NestedGINEffSRRef(10, NC, L, H, graph_pred=True, dropout=0.0, use_cycle=False)
    assert list(mine.state_dict().keys()) == list(sd0.keys())
    mine.load_state_dict(sd0)
    x = torch.randn(b["x"].shape[0], 10)                 # non-constant node features
    label = torch.tensor([0, 2, 1, 1])
    res = []
    for m, call in ((ref, lambda m: m(Bag(x=x, edge_index=b["edge_index"], batch=b["batch"], pos_enc=b["pos_enc"],
                                         pos_index=b["pos_index"], pos_batch=b["pos_batch"]))),
                    (mine, lambda m: m(x, b["edge_index"], b["pos_enc"], b["pos_index"], b["pos_batch"], b["batch"]))):
        m.train()
        out = call(m)
        loss = F.nll_loss(out, label)
        loss.backward()
        res.append((out.detach(), loss.detach(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k
    out = {"keys": np.array(list(sd0.keys())), "x": x.numpy(), "label": label.numpy(), "logp": res[0][0].numpy(),
           "loss": res[0][1].numpy(), "layers": np.int64(L), "hidden": np.int64(H), "classes": np.int64(NC)}
    for k, v in sd0.items():
        out["param/" + k] = v.numpy()
    for k, v in res[0][2].items():
        out["grad/" + k] = v.numpy()
    path = os.path.join(ROOT, "tests", "golden", "model_sr.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; loss", float(res[0][1]))


def main_zinc():
    """zinc_models.py:504-611 (hidden is hard-wired to 256 there): 2 layers on the 3-graph ZINC-like batch."""
    torch.set_num_threads(1)
    g = np.load(os.path.join(ROOT, "tests", "golden", "collate_zinc3.npz"))
    b = {k[len("batch_"):]: torch.tensor(g[k]) for k in g.files if k.startswith("batch_")}
    L = 2
    torch.manual_seed(777)
    Ref = reference_class("/root/reference/zinc_models.py")
    ref = Ref(None, L)
    with torch.no_grad():
        for name, p in ref.named_parameters():
            if p.dim() == 1 and "bias" not in name:
                p.add_(0.1 * torch.randn_like(p))
    sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
    mine = rm.NestedGINEffZincRef(L)
    assert list(mine.state_dict().keys()) == list(sd0.keys()), (list(mine.state_dict().keys())[:8], list(sd0.keys())[:8])
    mine.load_state_dict(sd0)
    res = []
    for m, call in ((ref, lambda m: m(Bag(x=b["x"], edge_index=b["edge_index"], edge_attr=b["edge_attr"], batch=b["batch"],
                                         pos_enc=b["pos_enc"], pos_index=b["pos_index"], pos_batch=b["pos_batch"]))),
                    (mine, lambda m: m(b["x"], b["edge_index"], b["edge_attr"], b["pos_enc"], b["pos_index"],
                                       b["pos_batch"], b["batch"]))):
        m.train()
        out = call(m)
        loss = F.l1_loss(out, b["y"].view(-1, 1))
        loss.backward()
        res.append((out.detach(), loss.detach(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k
    # parameters are 1.1 M floats at hidden=256: store the seed recipe instead of the tensors, plus outputs and a
    # digest of every gradient (sum, abs-sum) for the comparison
    out = {"keys": np.array(list(sd0.keys())), "pred": res[0][0].numpy(), "loss": res[0][1].numpy(), "layers": np.int64(L),
           "seed": np.int64(777)}
    for k, v in res[0][2].items():
        out["gsum/" + k] = np.array([float(v.double().sum()), float(v.double().abs().sum())])
    path = os.path.join(ROOT, "tests", "golden", "model_zinc.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; loss", float(res[0][1]))


def zinc_reference_init(L=2, seed=777):
    """Deterministic parameter recipe shared by the golden writer and the tests (torch CPU RNG)."""
    torch.manual_seed(seed)
    m = rm.NestedGINEffZincRef(L)
    return m


def reference_ogb_classes():
    """exec AtomEncoder, GINConv_eff, GNN_node_efficient and GNN of /root/reference/ogb_mol_gnn.py on shim primitives."""
    path = "/root/reference/ogb_mol_gnn.py"
    tree = ast.parse(open(path).read())
    want = ("AtomEncoder", "GINConv_eff", "GNN_node_efficient", "GNN")
    nodes = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in want]
    ns = dict(torch=torch, F=F, Linear=torch.nn.Linear, Sequential=torch.nn.Sequential, ReLU=torch.nn.ReLU,
              Dropout=torch.nn.Dropout, MessagePassing=rm.MessagePassing, BondEncoder=rm.BondEncoder,
              get_atom_feature_dims=lambda: list(rm.ATOM_FEATURE_DIMS), global_add_pool=rm.global_add_pool,
              global_mean_pool=rm.global_mean_pool, GNN_node=None)
    exec(compile(ast.Module(body=nodes, type_ignores=[]), path, "exec"), ns)
    return ns["GNN"]


def main_ogb():
    torch.set_num_threads(1)
    g = np.load(os.path.join(ROOT, "tests", "golden", "collate_molhiv4.npz"))
    b = {k[len("batch_"):]: torch.tensor(g[k]) for k in g.files if k.startswith("batch_")}
    L, H = 3, 32
    torch.manual_seed(2024)
    Ref = reference_ogb_classes()
    ref = Ref("ogbg-molhiv", 1, num_layer=L, emb_dim=H, gnn_type="gin_eff", virtual_node=True, residual=True,
              drop_ratio=0.0, JK="last", graph_pooling="mean")
    with torch.no_grad():
        for name, p in ref.named_parameters():
            if p.dim() == 1 and "bias" not in name:
                p.add_(0.1 * torch.randn_like(p))
        ref.gnn_node.virtualnode_embedding.weight.add_(0.1 * torch.randn(1, H))
    sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
    mine = rm.GNNEffRef(1, L, H, virtual_node=True, residual=True, drop_ratio=0.0, JK="last", graph_pooling="mean")
    assert list(mine.state_dict().keys()) == list(sd0.keys()), [a for a, c in zip(mine.state_dict().keys(), sd0.keys()) if a != c][:5]
    mine.load_state_dict(sd0)
    y = b["y"].float().view(-1, 1)
    res = []
    for m, call in ((ref, lambda m: m(Bag(x=b["x"], edge_index=b["edge_index"], edge_attr=b["edge_attr"], batch=b["batch"],
                                         pos_enc=b["pos_enc"], pos_index=b["pos_index"], pos_batch=b["pos_batch"]))),
                    (mine, lambda m: m(b["x"], b["edge_index"], b["edge_attr"], b["batch"], b["pos_enc"], b["pos_index"],
                                       b["pos_batch"]))):
        m.train()
        out = call(m)
        loss = F.binary_cross_entropy_with_logits(out, y)         # run_ogb_mol.py:65-72 (cls criterion)
        loss.backward()
        res.append((out.detach(), loss.detach(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k
    out = {"keys": np.array(list(sd0.keys())), "logit": res[0][0].numpy(), "loss": res[0][1].numpy(), "layers": np.int64(L),
           "hidden": np.int64(H)}
    for k, v in sd0.items():
        out["param/" + k] = v.numpy()
    for k, v in res[0][2].items():
        out["grad/" + k] = v.numpy()
    path = os.path.join(ROOT, "tests", "golden", "model_ogb.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; loss", float(res[0][1]))


if __name__ == "__main__":
    if "--only-ogb" in sys.argv:
        main_ogb()
    else:
        main()
        main_sr()
        main_zinc()
        main_ogb()

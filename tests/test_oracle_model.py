"""Oracle model (oracle/ref_model.py) against the composition golden produced by exec'ing the
reference's own NestedGIN_eff class body (tests/golden/model_count.npz), plus an independent fp64
per-edge loop for the two sparse primitives."""
import os

import numpy as np
import torch

from conftest import GOLDEN, load_collate
import ref_model as rm


def _golden_model():
    z = np.load(os.path.join(GOLDEN, "model_count.npz"))
    m = rm.NestedGINEffRef(int(z["layers"]), int(z["hidden"]))
    keys = [str(k) for k in z["keys"]]
    assert list(m.state_dict().keys()) == keys
    m.load_state_dict({k: torch.tensor(z["param/" + k]) for k in keys})
    return m, z


def test_state_dict_layout_full_size():
    want = [l.split() for l in open(os.path.join(GOLDEN, "state_dict_L4_H256.txt"))]
    m = rm.NestedGINEffRef(4, 256)
    got = [(k, "x".join(map(str, v.shape)) or "scalar") for k, v in m.state_dict().items()]
    assert got == [tuple(w) for w in want]
    assert sum(p.numel() for p in m.parameters()) == 1593359


def test_forward_backward_match_reference_composition():
    torch.set_num_threads(1)
    m, z = _golden_model()
    _, b, _ = load_collate("count3")
    b = {k: torch.tensor(v) for k, v in b.items()}
    m.train()
    pred = m(b["x"], b["edge_index"], b["pos_enc"], b["pos_index"], b["pos_batch"], b["batch"])
    loss = torch.nn.functional.l1_loss(pred, torch.tensor(z["y"]))
    loss.backward()
    assert torch.equal(pred.detach(), torch.tensor(z["pred_train"]))
    assert torch.equal(loss.detach(), torch.tensor(z["loss"]))
    for k, p in m.named_parameters():
        assert torch.allclose(p.grad, torch.tensor(z["grad/" + k]), rtol=1e-5, atol=1e-7), k
    m.eval()
    with torch.no_grad():
        pe = m(b["x"], b["edge_index"], b["pos_enc"], b["pos_index"], b["pos_batch"], b["batch"])
    assert torch.equal(pe, torch.tensor(z["pred_eval"]))


def test_primitives_against_fp64_loops():
    torch.manual_seed(0)
    _, b, _ = load_collate("count3")
    ei = torch.tensor(b["edge_index"]); N = b["x"].shape[0]; E = ei.shape[1]
    W = torch.randn(1800, 8)
    bag = rm.global_add_pool(W[torch.tensor(b["pos_index"])] * torch.tensor(b["pos_enc"]).view(-1, 1),
                             torch.tensor(b["pos_batch"]))
    ref = np.zeros((E, 8))
    for v, i, k in zip(b["pos_enc"], b["pos_index"], b["pos_batch"]):
        ref[k] += float(v) * W[i].double().numpy()
    assert np.allclose(bag.numpy(), ref, rtol=1e-5, atol=1e-5)
    conv = rm.GINEConv(torch.nn.Sequential(torch.nn.Linear(8, 8)), train_eps=True, edge_dim=8)
    conv.nn[0].weight.data = torch.eye(8); conv.nn[0].bias.data.zero_(); conv.eps.data.fill_(0.25)
    x = torch.randn(N, 8); ea = torch.randn(E, 8)
    out = conv(x, ei, ea).detach()
    e = conv.lin(ea).detach().double().numpy()
    ref = 1.25 * x.double().numpy()
    for k in range(E):
        ref[int(ei[1, k])] += np.maximum(x[int(ei[0, k])].double().numpy() + e[k], 0)
    assert np.allclose(out.numpy(), ref, rtol=1e-5, atol=1e-5)


def test_sr_variant_matches_reference_composition():
    """kernel/gin.py:200-379 (run_sr / run_exp model) golden produced from the reference class body."""
    torch.set_num_threads(1)
    z = np.load(os.path.join(GOLDEN, "model_sr.npz"))
    m = rm.NestedGINEffSRRef(10, int(z["classes"]), int(z["layers"]), int(z["hidden"]))
    keys = [str(k) for k in z["keys"]]
    assert list(m.state_dict().keys()) == keys
    m.load_state_dict({k: torch.tensor(z["param/" + k]) for k in keys})
    _, b, _ = load_collate("mixed4")
    b = {k: torch.tensor(v) for k, v in b.items()}
    m.train()
    out = m(torch.tensor(z["x"]), b["edge_index"], b["pos_enc"], b["pos_index"], b["pos_batch"], b["batch"])
    loss = torch.nn.functional.nll_loss(out, torch.tensor(z["label"]))
    assert torch.equal(out.detach(), torch.tensor(z["logp"])) and torch.equal(loss.detach(), torch.tensor(z["loss"]))


def zinc_oracle_from_recipe(z):
    """Rebuild the parameters of tests/golden/model_zinc.npz from its seed recipe (the fixture keeps outputs and
    gradient digests, not the 1.1 M parameters)."""
    torch.manual_seed(int(z["seed"]))
    m = rm.NestedGINEffZincRef(int(z["layers"]))
    with torch.no_grad():
        for name, p in m.named_parameters():
            if p.dim() == 1 and "bias" not in name:
                p.add_(0.1 * torch.randn_like(p))
    return m


def test_zinc_variant_matches_reference_composition():
    """zinc_models.py:504-611 golden produced from the reference class body (ELU, type embeddings, add-pool)."""
    torch.set_num_threads(1)
    z = np.load(os.path.join(GOLDEN, "model_zinc.npz"))
    m = zinc_oracle_from_recipe(z)
    assert list(m.state_dict().keys()) == [str(k) for k in z["keys"]]
    _, b, _ = load_collate("zinc3")
    b = {k: torch.tensor(v) for k, v in b.items()}
    m.train()
    out = m(b["x"], b["edge_index"], b["edge_attr"], b["pos_enc"], b["pos_index"], b["pos_batch"], b["batch"])
    loss = torch.nn.functional.l1_loss(out, b["y"].view(-1, 1))
    loss.backward()
    assert torch.equal(out.detach(), torch.tensor(z["pred"])) and torch.equal(loss.detach(), torch.tensor(z["loss"]))
    for n, p in m.named_parameters():
        s = z["gsum/" + n]
        assert abs(float(p.grad.double().sum()) - s[0]) <= 1e-6 * max(1.0, s[1]), n


def test_ogb_variant_matches_reference_composition():
    """ogb_mol_gnn.py gin_eff route (GNN / GNN_node_efficient / GINConv_eff / AtomEncoder class bodies exec'd on the
    oracle primitives) -> tests/golden/model_ogb.npz."""
    torch.set_num_threads(1)
    z = np.load(os.path.join(GOLDEN, "model_ogb.npz"))
    m = rm.GNNEffRef(1, int(z["layers"]), int(z["hidden"]), virtual_node=True, residual=True, drop_ratio=0.0)
    keys = [str(k) for k in z["keys"]]
    assert list(m.state_dict().keys()) == keys
    m.load_state_dict({k: torch.tensor(z["param/" + k]) for k in keys})
    _, b, _ = load_collate("molhiv4")
    b = {k: torch.tensor(v) for k, v in b.items()}
    m.train()
    out = m(b["x"], b["edge_index"], b["edge_attr"], b["batch"], b["pos_enc"], b["pos_index"], b["pos_batch"])
    loss = torch.nn.functional.binary_cross_entropy_with_logits(out, b["y"].float().view(-1, 1))
    assert torch.equal(out.detach(), torch.tensor(z["logit"])) and torch.equal(loss.detach(), torch.tensor(z["loss"]))

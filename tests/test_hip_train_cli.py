"""End-to-end: the run_graphcount harness trains on the synthetic count dataset (HIP feature build ->
HBM store -> device collate -> NestedGIN_eff -> FlatAdam), writes the reference's artefacts, and the
training loss goes down."""
import os

import pytest
import torch

from conftest import require_gpu

pytestmark = pytest.mark.gpu


def test_cli_two_epochs(tmp_path, monkeypatch, capsys):
    require_gpu()
    import esc_gnn_amd.run_graphcount as rg
    monkeypatch.chdir(tmp_path)
    rg.main("--epochs 2 --synthetic_graphs 40 --batch_size 8 --layers 2 --h 2 --lr 0.01 --save_appendix t".split())
    out = capsys.readouterr().out
    assert "Epoch: 001" in out and "Validation MAE" in out and "Test MAE norm" in out
    res = os.path.join(tmp_path, "results", "count_cycle_t")
    assert sorted(os.listdir(res)) == ["cmd_input.txt", "log.txt", "model_checkpoint2.pth", "run_graphcount.py",
                                       "utils_edge_efficient.py"]
    sd = torch.load(os.path.join(res, "model_checkpoint2.pth"), map_location="cpu")
    assert "z_initial.weight" in sd and sd["z_initial.weight"].shape == (1800, 256)
    # --load_model + --eval round trip
    rg.main(("--eval 1 --synthetic_graphs 40 --batch_size 8 --layers 2 --h 2 --save_appendix t --load_model "
             + os.path.join(res, "model_checkpoint2.pth")).split())
    assert "Test MAE" in capsys.readouterr().out


def test_loss_decreases():
    require_gpu()
    import esc_gnn_amd as E
    from esc_gnn_amd.datasets import build_count_dataset
    torch.manual_seed(0)
    graphs = build_count_dataset(0, 64, h=3)
    y = torch.cat([g.y for g in graphs]); mean, std = y.mean(), y.std()
    for g in graphs:
        g.y = (g.y - mean) / std
    store = E.DeviceGraphStore(graphs, "cuda:0")
    model = E.NestedGIN_eff(None, 3, 64, use_rd=True, graph_pred=False, dropout=0, edge_nest=True, use_cycle=True).to("cuda:0")
    opt = E.optim.FlatAdam(model.parameters(), lr=5e-3)
    losses = []
    for it in range(40):
        b = store.collate(list(range((it % 4) * 16, (it % 4) * 16 + 16)))
        opt.zero_grad()
        loss = E.ops.l1_loss(model(b), b.y)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert sum(losses[-4:]) < 0.7 * sum(losses[:4]), losses


def test_bce_logits_masked_matches_torch():
    require_gpu()
    import esc_gnn_amd as E
    torch.manual_seed(3)
    pred = (torch.randn(37, 5) * 3).cuda().requires_grad_(True)
    y = (torch.rand(37, 5) > 0.5).float()
    y[torch.rand(37, 5) < 0.3] = float("nan")
    loss = E.ops.bce_with_logits_loss(pred, y.cuda())
    loss.backward()
    p64 = pred.detach().cpu().double().requires_grad_(True)
    lab = y == y
    ref = torch.nn.BCEWithLogitsLoss()(p64[lab], y.double()[lab])
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-6 * max(1.0, abs(float(ref)))
    assert torch.allclose(pred.grad.cpu().double(), p64.grad, atol=1e-7)


def test_zinc_cli_two_epochs(tmp_path, monkeypatch, capsys):
    require_gpu()
    import esc_gnn_amd.run_zinc as rz
    monkeypatch.chdir(tmp_path)
    rz.main("--epochs 2 --synthetic_graphs 60 --batch_size 10 --layers 2 --h 2 --lr 0.005 --save_appendix _t".split())
    out = capsys.readouterr().out
    assert "Epoch: 001" in out and "Validation MAE" in out
    res = os.path.join(tmp_path, "results", "zinc_NestedGIN_eff_t")
    assert "model_checkpoint2.pth" in os.listdir(res) and "log.txt" in os.listdir(res)
    sd = torch.load(os.path.join(res, "model_checkpoint2.pth"), map_location="cpu")
    assert sd["node_type_embedding.weight"].shape == (100, 32) and sd["convs.0.lin.weight"].shape == (256, 288)


def test_ogb_cli_runs_checkpoints_and_ensemble(tmp_path, monkeypatch, capsys):
    require_gpu()
    import esc_gnn_amd.run_ogb_mol as ro
    monkeypatch.chdir(tmp_path)
    base = ("--dataset ogbg-molpcba --gnn gin_eff --edge_nest True --efficient True --self_loop True --h 2 --runs 1 "
            "--num_layer 2 --emb_dim 64 --batch_size 16 --synthetic_graphs 80 --log_steps 1 --save_appendix _t ")
    ro.main((base + "--epochs 2 --ensemble --ensemble_lookback 1 --ensemble_interval 1").split())
    out = capsys.readouterr().out
    assert "Best validation score" in out and "Ensemble test score" in out and "Final Test" in out
    res = os.path.join(tmp_path, "results", "ogbg-molpcba_t")
    for f in ("run1_best_model.pth", "run1_model_checkpoint2.pth", "run1_optimizer_checkpoint2.pth", "log.txt"):
        assert f in os.listdir(res), f
    ro.main((base + "--epochs 3 --continue_from 2").split())          # resume: trains epoch 3 only
    out = capsys.readouterr().out
    assert "epoch 3" in out and "epoch 2," not in out


def test_reference_style_loop_with_torch_loss_and_optimizer():
    """The loop of the reference (run_graphcount.py:494-505) verbatim — torch.nn.L1Loss, torch.optim.Adam,
    optimizer.zero_grad / loss.backward / optimizer.step — on the drop-in module: the training-mode forward runs as one
    autograd node on the whole-step engine, and it learns."""
    require_gpu()
    import esc_gnn_amd as E
    from esc_gnn_amd.datasets import build_count_dataset
    torch.manual_seed(0)
    graphs = build_count_dataset(0, 64, h=3)
    y = torch.cat([g.y for g in graphs]); mean, std = y.mean(), y.std()
    for g in graphs:
        g.y = (g.y - mean) / std
    loader = E.DataLoader(graphs, batch_size=16, shuffle=False)            # the reference's loader call; batches are collated on the device from a pinned copy
    model = E.NestedGIN_eff(None, 3, 64, use_rd=True, graph_pred=False, dropout=0, edge_nest=True, use_cycle=True).to("cuda:0")
    optimizer = torch.optim.Adam(model.parameters(), lr=5e-3)
    losses = []
    for epoch in range(10):
        model.train()
        for data in loader:
            data = data.to("cuda:0")
            optimizer.zero_grad()
            yy = data.y.view([data.y.size(0), 1])
            out = model(data)
            assert type(out.grad_fn).__name__.startswith("_EngineNode")
            loss = torch.nn.L1Loss()(out, yy)
            loss.backward()
            optimizer.step()
            losses.append(float(loss.detach()))
    assert sum(losses[-4:]) < 0.7 * sum(losses[:4]), losses

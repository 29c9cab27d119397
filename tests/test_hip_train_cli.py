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

"""CLI surface of run_graphcount (reference :315-358): same flags, types and defaults."""
import esc_gnn_amd.run_graphcount as rg


def test_reference_flags_and_defaults():
    a = rg.build_parser().parse_args([])
    want = dict(model="NestedGIN_eff", target=3, ab=False, layers=5, h=3, max_nodes_per_hop=None, node_label="hop",
                epochs=2000, batch_size=256, lr=1e-3, lr_decay_factor=0.9, patience=10, normalize_x=False,
                not_normalize_dist=False, RNI=False, use_relative_pos=False, seed=0, save_appendix="",
                keep_old=False, dataset="count_cycle", load_model=None, eval=0, train_only=0)
    for k, v in want.items():
        assert getattr(a, k) == v, k
    b = rg.build_parser().parse_args("--h 4 --layers 4 --batch_size 128 --dataset count_graphlet --target 1".split())
    assert (b.h, b.layers, b.batch_size, b.dataset, b.target) == (4, 4, 128, "count_graphlet", 1)


def test_scheduler_matches_torch():
    import torch
    from esc_gnn_amd.optim import ReduceLROnPlateau

    class Opt(object):
        param_groups = [dict(lr=1e-3)]
    p = torch.nn.Parameter(torch.zeros(1))
    ref_opt = torch.optim.Adam([p], lr=1e-3)
    ref = torch.optim.lr_scheduler.ReduceLROnPlateau(ref_opt, mode="min", factor=0.9, patience=2, min_lr=1e-5)
    mine = ReduceLROnPlateau(Opt, mode="min", factor=0.9, patience=2, min_lr=1e-5)
    seq = [1.0, 0.9, 0.95, 0.96, 0.97, 0.98, 0.5, 0.6, 0.6, 0.6, 0.6, 0.6, 0.6]
    for v in seq:
        ref.step(v)
        mine.step(v)
        assert abs(ref_opt.param_groups[0]["lr"] - Opt.param_groups[0]["lr"]) < 1e-12


def test_zinc_flags_and_defaults():
    import esc_gnn_amd.run_zinc as rz
    a = rz.build_parser().parse_args([])
    want = dict(target=0, filter=False, convert="post", layers=6, h=3, max_nodes_per_hop=None, node_label="spd",
                use_rd=True, epochs=1000, batch_size=256, lr=1e-3, lr_decay_factor=0.95, patience=10, drop_ratio=0.0,
                self_loop=False, seed=1, save_appendix="", dataset="zinc", load_model=None, eval=0, train_only=0)
    for k, v in want.items():
        assert getattr(a, k) == v, k


def test_ogb_flags_and_defaults():
    import esc_gnn_amd.run_ogb_mol as ro
    a = ro.build_parser().parse_args([])
    want = dict(dataset="ogbg-molhiv", runs=10, gnn="gin", virtual_node=True, residual=True, drop_ratio=0.65,
                num_layer=5, emb_dim=300, h=None, graph_pooling="mean", use_rd=True, edge_nest=False, self_loop=False,
                efficient=False, batch_size=32, epochs=100, lr=2e-4, lr_decay_factor=0.5, ensemble=False,
                ensemble_lookback=70, ensemble_interval=10, scheduler=False, log_steps=10, continue_from=None,
                run_from=1, save_appendix="_h4_l6_spd_rd_gin_edge_eff")
    for k, v in want.items():
        assert getattr(a, k) == v, k
    b = ro.build_parser().parse_args("--edge_nest False --efficient True".split())
    assert b.edge_nest is True and b.efficient is True        # the reference's type=bool quirk: any non-empty string


def test_ogb_metrics_match_sklearn():
    import numpy as np
    from sklearn.metrics import average_precision_score, roc_auc_score
    from esc_gnn_amd.metrics import Evaluator
    rng = np.random.RandomState(0)
    y = (rng.rand(300, 5) > 0.7).astype(np.float32)
    y[rng.rand(300, 5) < 0.3] = np.nan
    y[:, 4] = np.where(np.isnan(y[:, 4]), np.nan, 0.0)        # a task without positives is skipped
    s = np.round(rng.randn(300, 5), 1)                        # ties
    for name, fn in (("ogbg-molhiv", roc_auc_score), ("ogbg-molpcba", average_precision_score)):
        want = np.mean([fn(y[~np.isnan(y[:, t]), t], s[~np.isnan(y[:, t]), t]) for t in range(4)])
        got = Evaluator(name).eval({"y_true": y, "y_pred": s})
        assert abs(list(got.values())[0] - want) < 1e-12


def test_step_lr_matches_torch():
    import torch
    from esc_gnn_amd.run_ogb_mol import StepLR

    class Opt(object):
        param_groups = [dict(lr=2e-4)]
    ref_opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=2e-4)
    ref = torch.optim.lr_scheduler.StepLR(ref_opt, step_size=20, gamma=0.5)
    mine = StepLR(Opt, 20, 0.5)
    for _ in range(65):
        ref_opt.step()
        ref.step()
        mine.step()
        assert abs(ref_opt.param_groups[0]["lr"] - Opt.param_groups[0]["lr"]) < 1e-15

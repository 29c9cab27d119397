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

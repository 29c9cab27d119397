"""NestedGIN_eff for the substructure-counting benchmark + its training harness — the MI355X-native
twin of /root/reference/run_graphcount.py (model :39-194, CLI :315-358, data wiring :393-455,
train/test/loop :483-613).

Same constructor, forward contract and state_dict key layout as the reference class, so a
checkpoint written by either loads into the other (`--load_model`, reference :472-474).
All device arithmetic of forward/backward goes through libescgnn_hip.so.
"""
import torch
import torch.nn.functional as F
from torch.nn import Dropout, Sequential

from . import ops
from .nn import AbsorbedReLU, BatchNorm1d, GINEConv, Linear, global_mean_pool
from .plan import plan_of

Z_TABLE_ROWS = 1800  # reference :51 — always 1800, even for the 1700-wide no-rd layout


def _bn_relu(hidden):
    return BatchNorm1d(hidden, fuse_relu=True), AbsorbedReLU()


def _mlp(n_in, hidden, p):
    return Sequential(Linear(n_in, hidden), Dropout(p), *_bn_relu(hidden),
                      Linear(hidden, hidden), Dropout(p), *_bn_relu(hidden))


class NestedGIN_eff(torch.nn.Module):
    def __init__(self, dataset, num_layers, hidden, use_z=False, use_rd=False, use_cycle=False, graph_pred=True,
                 use_id=None, dropout=0.2, multi_layer=False, edge_nest=False):
        super().__init__()
        if use_id is not None:
            raise NotImplementedError("use_id: the identity-aware baseline is outside the ESC hot path")
        # stored-but-unused flags are kept for interface parity (reference :43-50)
        self.use_rd, self.use_z, self.graph_pred, self.use_cycle = use_rd, True, graph_pred, use_cycle
        self.use_id, self.dropout, self.multi_layer, self.edge_nest = use_id, dropout, multi_layer, edge_nest
        input_dim = 10
        self.z_initial = torch.nn.Embedding(Z_TABLE_ROWS, hidden)
        self.z_embedding = Sequential(Dropout(dropout), *_bn_relu(hidden), Linear(hidden, hidden),
                                      Dropout(dropout), *_bn_relu(hidden))
        self.x_embedding = _mlp(input_dim, hidden, dropout)
        self.conv1 = GINEConv(_mlp(input_dim, hidden, dropout), train_eps=True, edge_dim=hidden)
        self.convs = torch.nn.ModuleList(
            GINEConv(_mlp(hidden, hidden, dropout), train_eps=True, edge_dim=hidden)
            for _ in range(num_layers - 1))
        self.lin1 = Linear(num_layers * hidden + hidden, hidden)
        self.bn_lin1 = BatchNorm1d(hidden, eps=1e-5, momentum=0.1, fuse_relu=True)
        self.lin2 = Linear(hidden, 1 if use_cycle else dataset.num_classes)
        self.engine_forward = True       # training-mode forward through the whole-step engine when it covers the config

    def reset_parameters(self):
        for layer in self.z_embedding.children():
            if hasattr(layer, "reset_parameters"):
                layer.reset_parameters()
        self.conv1.reset_parameters()
        for conv in self.convs:
            conv.reset_parameters()
        self.lin1.reset_parameters()
        self.bn_lin1.reset_parameters()
        self.lin2.reset_parameters()

    def forward(self, data, return_embeddings=False):
        data.to(self.lin1.weight.device)
        x, edge_index, batch = data.x, data.edge_index, data.batch
        if (self.training and torch.is_grad_enabled() and not return_embeddings and self.engine_forward
                and "edge_pos" not in data and x.is_floating_point() and x.dim() == 2 and x.size(0) >= 2
                and edge_index.size(1) >= 2 and "pos_batch" in data):       # the engine's own preconditions (esc::check)
            from .engine import MAX_LAYERS, engine_forward, engine_supports, _node_cache
            cache = _node_cache(self) if self.lin1.weight.device.type == "cuda" and 1 + len(self.convs) <= MAX_LAYERS else None
            if engine_supports(self, cache) and x.size(1) == self.x_embedding[0].in_features:
                return engine_forward(self, data, cache)       # the whole step as one autograd node (engine.hip)
        if (not self.training and not torch.is_grad_enabled() and not return_embeddings and self.engine_forward
                and "edge_pos" not in data and x.is_floating_point() and x.dim() == 2 and x.size(0) >= 2
                and edge_index.size(1) >= 2 and "pos_batch" in data):
            from .engine import engine_predict, engine_supports
            if engine_supports(self) and x.size(1) == self.x_embedding[0].in_features:
                return engine_predict(self, data)       # eval-mode forward as one call (esc_engine_predict)
        plan = plan_of(data, Z_TABLE_ROWS)
        if "edge_pos" in data:                       # dense layout of the slow variant (reference :142-145)
            z = ops.linear(data.edge_pos.float(), self.z_initial.weight.t().contiguous())
        else:
            z = ops.esc_bag(self.z_initial.weight, plan)
        z = self.z_embedding(z)
        h = self.conv1(x, edge_index, z, plan)
        xs = [self.x_embedding(x), h]
        for conv in self.convs:
            h = conv(h, edge_index, z, plan)
            xs.append(h)
        cat = torch.cat(xs, dim=1)
        if self.graph_pred:
            cat = global_mean_pool(cat, batch)
        o = self.lin1(cat)
        o = self.bn_lin1(o) if o.size(0) > 1 else F.relu(o)      # bn_lin1 carries the ReLU of reference :186
        o = F.dropout(o, p=self.dropout, training=self.training)
        o = self.lin2(o)
        if not self.use_cycle:
            o = F.log_softmax(o, dim=-1)
        return (o, cat) if return_embeddings else o


# =====================================================================================================
# Training harness — `python -m esc_gnn_amd.run_graphcount ...` (flags of reference :315-358)
# =====================================================================================================
_FLAGS = [  # (name, kwargs) — same names, types and defaults as the reference CLI
    ("--model", dict(default="NestedGIN_eff", type=str, help="NestedGIN_eff (PPGN_eff: dense 3-WL baseline, out of scope)")),
    ("--target", dict(default=3, type=int)),
    ("--ab", dict(action="store_true", default=False)),
    ("--layers", dict(type=int, default=5)),
    ("--h", dict(type=int, default=3, help="hop of enclosing subgraph")),
    ("--max_nodes_per_hop", dict(type=int, default=None)),
    ("--node_label", dict(type=str, default="hop")),
    ("--epochs", dict(type=int, default=2000)),
    ("--batch_size", dict(type=int, default=256)),
    ("--lr", dict(type=float, default=1e-3)),
    ("--lr_decay_factor", dict(type=float, default=0.9)),
    ("--patience", dict(type=int, default=10)),
    ("--normalize_x", dict(action="store_true", default=False)),
    ("--not_normalize_dist", dict(action="store_true", default=False)),
    ("--RNI", dict(action="store_true", default=False)),
    ("--use_relative_pos", dict(action="store_true", default=False)),
    ("--seed", dict(type=int, default=0)),
    ("--save_appendix", dict(default="")),
    ("--keep_old", dict(action="store_true", default=False)),
    ("--dataset", dict(default="count_cycle", help="count_cycle/count_graphlet")),
    ("--load_model", dict(default=None)),
    ("--eval", dict(default=0, type=int)),
    ("--train_only", dict(default=0, type=int)),
    # additions (not in the reference): synthetic data when data/<dataset>/raw/data.mat is absent
    ("--synthetic_graphs", dict(type=int, default=5000, help="size of the synthetic count_cycle-shaped dataset")),
    ("--data_root", dict(default="data")),
]


def build_parser():
    import argparse
    ap = argparse.ArgumentParser(description="NestedGNN for counting experiments (MI355X hot path).")
    for name, kw in _FLAGS:
        ap.add_argument(name, **kw)
    return ap


def _load_splits(args):
    """train/val/test lists of pre-transformed Data (reference :404-430)."""
    import os
    from .datasets import build_count_dataset, load_count_mat
    from .utils_edge_efficient import create_subgraphs_many
    mat = os.path.join(args.data_root, args.dataset, "raw", "data.mat")
    if os.path.exists(mat):
        splits = []
        for name in ("train", "val", "test"):
            raw = load_count_mat(mat, name)
            splits.append(create_subgraphs_many(raw, args.h, use_rd=True, self_loop=True))
        return splits, True
    G = args.synthetic_graphs
    n_tr, n_val = int(0.3 * G), int(0.2 * G)              # 30/20/50 split by index
    alld = build_count_dataset(0, G, h=args.h, use_rd=True, self_loop=True)
    return [alld[:n_tr], alld[n_tr:n_tr + n_val], alld[n_tr + n_val:]], False


def main(argv=None):
    import os
    import random
    import shutil
    import sys
    import time

    import numpy as np
    import torch.distributed as dist

    from . import ops
    from .engine import StepEngine
    from .optim import FlatAdam, ReduceLROnPlateau
    from .parallel import broadcast_buffers, broadcast_parameters, edge_pipeline_parameters, shard_slice
    from .store import DeviceGraphStore

    args = build_parser().parse_args(argv)
    if args.model != "NestedGIN_eff":
        print("Model not implemented")
        raise NotImplementedError
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("run_graphcount: needs a HIP device (the hot path has no CPU fallback)")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:                                         # RCCL over xGMI; ESC_DIST_BACKEND=gloo only for rehearsing on one GPU
        backend = os.environ.get("ESC_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))

    torch.manual_seed(args.seed)                          # reference :361-366
    torch.cuda.manual_seed_all(args.seed)
    random.seed(args.seed)
    np.random.seed(args.seed)

    if args.save_appendix == "":
        args.save_appendix = "_" + time.strftime("%Y%m%d%H%M%S")
    args.res_dir = "results/" + args.dataset + "_" + args.save_appendix
    cmd_input = "python " + " ".join(sys.argv) + "\n"
    if rank == 0:
        print("Results will be saved in " + args.res_dir)
        os.makedirs(args.res_dir, exist_ok=True)
        here = os.path.dirname(os.path.abspath(__file__))
        for f in ("run_graphcount.py", "utils_edge_efficient.py"):     # reference :381-383 backs its sources up
            shutil.copy(os.path.join(here, f), args.res_dir)
        with open(os.path.join(args.res_dir, "cmd_input.txt"), "a") as fh:
            fh.write(cmd_input)
        print("Command line input: " + cmd_input + " is saved.")
    target = int(args.target)
    if rank == 0:
        print("---- Target: {} ----".format(target))

    (tr, va, te), real = _load_splits(args)

    def column(d):                                        # MyTransform (reference :35-37)
        y = d.y
        return y[:, target] if (real and y.dim() == 2) else y.reshape(-1)
    for part in (tr, va, te):
        for d in part:
            d.y = column(d).float()
    y_train_val = torch.cat([d.y for d in tr + va])       # reference :441-447
    mean, std = y_train_val.mean(), y_train_val.std()
    for part in (tr, va, te):
        for d in part:
            d.y = (d.y - mean) / std
    if rank == 0:
        print("Mean = %.3f, Std = %.3f" % (float(mean), float(std)))
    stores = [DeviceGraphStore(part, device) for part in (tr, va, te)]
    n_train_targets = sum(d.y.numel() for d in tr)

    model = NestedGIN_eff(None, args.layers, 256, use_rd=True, graph_pred=False, dropout=0, edge_nest=True,
                          use_cycle=True)                # reference :465
    if args.load_model is not None:
        model.load_state_dict(torch.load(args.load_model, map_location="cpu"))
    if rank == 0:
        print("Using " + model.__class__.__name__ + " model")
    model = model.to(device)
    broadcast_parameters(model, 0)
    # world > 1: two gradient buckets — node-pipeline parameters first, the edge pipeline's (whose gradients are final only
    # after the backward tail) behind them
    optimizer = FlatAdam(model.parameters(), lr=args.lr, late=edge_pipeline_parameters(model))
    scheduler = ReduceLROnPlateau(optimizer, mode="min", factor=args.lr_decay_factor, patience=args.patience,
                                  min_lr=0.00001)
    engine = StepEngine(model)         # one native call per step; same parameters / .grad slots / BN buffers
    gen = torch.Generator().manual_seed(args.seed)

    def shards(store, shuffle):
        G = len(store)
        order = torch.randperm(G, generator=gen) if shuffle else torch.arange(G)
        for i in range(0, G, args.batch_size):
            ids = order[i:i + args.batch_size]
            if ids.numel() < world:       # fewer graphs than ranks: some ranks would miss the step's collectives —
                if not shuffle and rank == 0:                        # training drops it, evaluation gives it to rank 0
                    yield ids
                continue
            lo, hi = shard_slice(ids.numel(), rank, world)           # shard the global batch by graph
            yield ids[lo:hi]

    def batches(store, shuffle):
        for ids in shards(store, shuffle):
            yield store.collate(ids)

    def train(epoch):
        model.train()
        loss_all = torch.zeros((), device=device)
        todo = shards(stores[0], True)
        first = next(todo, None)
        data = None if first is None else stores[0].collate(first)
        while data is not None:
            n_local = data.y.size(0)
            # forward + L1Loss + backward (reference :494-503); the next batch is collated between the two halves of
            # the step, while the edge pipeline finishes.  world > 1: sum-gradients, one all-reduce of grad ++ [n_local],
            # division inside the Adam launch
            loss = engine.begin_step(data, loss_denom=1 if world > 1 else None)
            ids = next(todo, None)
            upcoming = None if ids is None else stores[0].collate(ids)
            if world > 1:
                optimizer.all_reduce_early()        # overlaps the edge pipeline's backward tail
            engine.end_step()
            if world > 1:
                loss_all += loss
                optimizer.step(grad_denom=optimizer.all_reduce_late(n_local))
            else:
                loss_all += loss * n_local
                optimizer.step()
            data = upcoming
        if world > 1:
            dist.all_reduce(loss_all)
        return float(loss_all) / n_train_targets

    def test(store):
        # BatchNorm running statistics were updated from rank-local shards: evaluate (and later checkpoint) rank 0's on
        # every rank, so that the logged MAE is the one the saved model reproduces
        broadcast_buffers(model, 0)
        model.eval()
        err, num = torch.zeros((), device=device), 0
        with torch.no_grad():
            for data in batches(store, False):
                y_hat = engine.predict(data)[:, 0]
                err += torch.sum(torch.abs(y_hat - data.y))
                num += data.y.size(0)
        tot = torch.stack([err, torch.tensor(float(num), device=device)])
        if world > 1:
            dist.all_reduce(tot)
        return float(tot[0] / tot[1]) * float(std)

    if args.eval:
        print("Test MAE: %.7f" % test(stores[2]))
        return
    best_val_error, count, log = None, 0, ""
    for epoch in range(1, args.epochs + 1):
        lr = optimizer.param_groups[0]["lr"]
        loss = train(epoch)
        val_error = test(stores[1])
        scheduler.step(val_error)
        count += 1
        if best_val_error is None:
            best_val_error = val_error
        if val_error <= best_val_error or count == 10:    # reference :595-598
            test_error = test(stores[2])
            best_val_error, count = val_error, 0
            log = ("Epoch: {:03d}, LR: {:7f}, Loss: {:.7f}, Validation MAE: {:.7f}, "
                   "Test MAE: {:.7f}, Test MAE norm: {:.7f}").format(epoch, lr, loss, val_error, test_error,
                                                                     test_error / float(std))
            if rank == 0:
                print("\n" + log + "\n")
                with open(os.path.join(args.res_dir, "log.txt"), "a") as fh:
                    fh.write(log + "\n")
    if rank == 0:
        torch.save(model.state_dict(), os.path.join(args.res_dir, "model_checkpoint{}.pth".format(args.epochs)))
        print(cmd_input[:-1])
        print(log)
        with open(os.path.join(args.res_dir, "log.txt"), "a") as fh:
            fh.write(log + "\n")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

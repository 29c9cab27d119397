"""ORACLE (test infrastructure, not product code): pure-PyTorch CPU restatement of the
NestedGIN_eff message-passing path, op-for-op what PyG 2.0.4 dispatches to on CPU
(index_select gather -> relu(x_j + e) -> scatter-add; global_add_pool = scatter-add).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this.

Follows:
  * GINEConv semantics         /root/reference/GraphGPS/graphgps/layer/gine_conv_layer.py:18-84
                               (minus its r_ij factor) and the call sites
                               /root/reference/run_graphcount.py:77-89,97-109,161,169
  * ESC bag (global_add_pool)  /root/reference/run_graphcount.py:155
  * model composition          /root/reference/run_graphcount.py:39-194
  * loss / optimiser           /root/reference/run_graphcount.py:478-505

Parity status: the three PyG primitives (GINEConv, global_add_pool, global_mean_pool) live in
the un-vendored dependency torch-geometric==2.0.4 (requirements.txt:94); no reference test pins
them => "parity unpinned" at that boundary.  The *composition* is pinned: make_golden.py execs
the reference's own NestedGIN_eff class body on top of these primitives and the result is
compared with NestedGINEffRef below (tests/golden/model_count.npz).
"""
import torch
import torch.nn.functional as F
from torch.nn import BatchNorm1d, Dropout, Linear, ReLU, Sequential


def global_add_pool(x, batch, size=None):
    """PyG: scatter(x, batch, dim=0, dim_size=batch.max()+1, reduce='add')."""
    size = int(batch.max()) + 1 if size is None else size
    out = x.new_zeros((size,) + tuple(x.shape[1:]))
    return out.index_add_(0, batch, x)


def global_mean_pool(x, batch, size=None):
    size = int(batch.max()) + 1 if size is None else size
    s = global_add_pool(x, batch, size)
    cnt = torch.zeros(size, dtype=x.dtype).index_add_(0, batch, torch.ones_like(batch, dtype=x.dtype))
    return s / cnt.clamp(min=1).view(-1, *([1] * (x.dim() - 1)))


class GINEConv(torch.nn.Module):
    """out = nn( sum_{k: dst_k = i} relu(x[src_k] + lin(edge_attr_k)) + (1+eps) * x_i )."""

    def __init__(self, nn, eps=0.0, train_eps=False, edge_dim=None):
        super().__init__()
        self.nn = nn
        self.initial_eps = eps
        if train_eps:
            self.eps = torch.nn.Parameter(torch.Tensor([eps]))
        else:
            self.register_buffer("eps", torch.Tensor([eps]))
        self.lin = Linear(edge_dim, nn[0].in_features) if edge_dim is not None else None

    def reset_parameters(self):
        for m in self.nn:
            if hasattr(m, "reset_parameters"):
                m.reset_parameters()
        self.eps.data.fill_(self.initial_eps)
        if self.lin is not None:
            self.lin.reset_parameters()

    def forward(self, x, edge_index, edge_attr):
        e = self.lin(edge_attr) if self.lin is not None else edge_attr
        msg = (x.index_select(0, edge_index[0]) + e).relu()
        out = torch.zeros_like(x).index_add_(0, edge_index[1], msg)
        out = out + (1 + self.eps) * x
        return self.nn(out)


def _mlp(i, h, p):
    return Sequential(Linear(i, h), Dropout(p), BatchNorm1d(h), ReLU(),
                      Linear(h, h), Dropout(p), BatchNorm1d(h), ReLU())


class NestedGINEffRef(torch.nn.Module):
    """Same module tree / state_dict keys as run_graphcount.py:39-121, forward :134-194."""

    def __init__(self, num_layers, hidden, graph_pred=False, dropout=0.0, use_cycle=True,
                 num_classes=1, input_dim=10, z_in=1800):
        super().__init__()
        self.graph_pred, self.dropout, self.use_cycle = graph_pred, dropout, use_cycle
        self.z_initial = torch.nn.Embedding(z_in, hidden)
        self.z_embedding = Sequential(Dropout(dropout), BatchNorm1d(hidden), ReLU(), Linear(hidden, hidden),
                                      Dropout(dropout), BatchNorm1d(hidden), ReLU())
        self.x_embedding = _mlp(input_dim, hidden, dropout)
        self.conv1 = GINEConv(_mlp(input_dim, hidden, dropout), train_eps=True, edge_dim=hidden)
        self.convs = torch.nn.ModuleList(
            [GINEConv(_mlp(hidden, hidden, dropout), train_eps=True, edge_dim=hidden)
             for _ in range(num_layers - 1)])
        self.lin1 = Linear(num_layers * hidden + hidden, hidden)
        self.bn_lin1 = BatchNorm1d(hidden, eps=1e-5, momentum=0.1)
        self.lin2 = Linear(hidden, 1 if use_cycle else num_classes)

    def bag(self, pos_enc, pos_index, pos_batch, num_edges=None):
        return global_add_pool(self.z_initial.weight[pos_index] * pos_enc.view(-1, 1), pos_batch, num_edges)

    def forward(self, x, edge_index, pos_enc, pos_index, pos_batch, batch=None, return_embeddings=False):
        z = self.z_embedding(self.bag(pos_enc, pos_index, pos_batch))
        h = self.conv1(x, edge_index, z)
        xs = [self.x_embedding(x), h]
        for conv in self.convs:
            h = conv(h, edge_index, z)
            xs.append(h)
        cat = torch.cat(xs, dim=1)
        if self.graph_pred:
            cat = global_mean_pool(cat, batch)
        o = self.lin1(cat)
        if o.size(0) > 1:
            o = self.bn_lin1(o)
        o = F.dropout(F.relu(o), p=self.dropout, training=self.training)
        o = self.lin2(o)
        if not self.use_cycle:
            o = F.log_softmax(o, dim=-1)
        return (o, cat) if return_embeddings else o


class NestedGINEffSRRef(torch.nn.Module):
    """kernel/gin.py:200-379 composition (run_sr.py / run_exp.py): no x_embedding, lin1 over L*H, dropout before
    the last ReLU, mean-pool readout, log_softmax head unless use_cycle.  Same state_dict keys."""

    def __init__(self, num_features, num_classes, num_layers, hidden, graph_pred=True, dropout=0.0, use_cycle=False):
        super().__init__()
        self.graph_pred, self.dropout, self.use_cycle = graph_pred, dropout, use_cycle
        self.z_initial = torch.nn.Embedding(1800, hidden)
        self.z_embedding = Sequential(Dropout(dropout), BatchNorm1d(hidden), ReLU(), Linear(hidden, hidden),
                                      Dropout(dropout), BatchNorm1d(hidden), ReLU())
        self.conv1 = GINEConv(_mlp(num_features, hidden, dropout), train_eps=True, edge_dim=hidden)
        self.convs = torch.nn.ModuleList(
            [GINEConv(_mlp(hidden, hidden, dropout), train_eps=True, edge_dim=hidden) for _ in range(num_layers - 1)])
        self.lin1 = Linear(num_layers * hidden, hidden)
        self.bn_lin1 = BatchNorm1d(hidden, eps=1e-5, momentum=0.1)
        self.lin2 = Linear(hidden, 1 if use_cycle else num_classes)

    def forward(self, x, edge_index, pos_enc, pos_index, pos_batch, batch):
        z = global_add_pool(self.z_initial.weight[pos_index] * pos_enc.view(-1, 1), pos_batch)
        z = self.z_embedding(z)
        h = self.conv1(x, edge_index, z)
        xs = [h]
        for conv in self.convs:
            h = conv(h, edge_index, z)
            xs.append(h)
        o = torch.cat(xs, dim=1)
        if self.graph_pred:
            o = global_mean_pool(o, batch)
        o = self.lin1(o)
        if o.size(0) > 1:
            o = self.bn_lin1(o)
        o = F.relu(F.dropout(o, p=self.dropout, training=self.training))
        o = self.lin2(o)
        return o if self.use_cycle else F.log_softmax(o, dim=-1)


class NestedGINEffZincRef(torch.nn.Module):
    """zinc_models.py:504-611 composition: ELU, node/edge type embeddings, edge term [z_emb | edge_type_emb],
    add-pool readout.  Same state_dict keys."""

    def __init__(self, num_layers, hidden=256):
        super().__init__()
        from torch.nn import ELU

        def mlp(i):
            return Sequential(Linear(i, hidden), Dropout(0.0), BatchNorm1d(hidden), ELU(),
                              Linear(hidden, hidden), Dropout(0.0), BatchNorm1d(hidden), ELU())
        self.z_initial = torch.nn.Embedding(1800, hidden)
        self.z_embedding = Sequential(Dropout(0.0), BatchNorm1d(hidden), ELU(), Linear(hidden, hidden), Dropout(0.0),
                                      BatchNorm1d(hidden), ELU())
        self.conv1 = GINEConv(mlp(32), train_eps=True, edge_dim=hidden + 32)
        self.convs = torch.nn.ModuleList([GINEConv(mlp(hidden), train_eps=True, edge_dim=hidden + 32)
                                          for _ in range(num_layers - 1)])
        self.lin1 = Linear(num_layers * hidden, hidden)
        self.bn_lin1 = BatchNorm1d(hidden, eps=1e-5, momentum=0.1)
        self.lin2 = Linear(hidden, 1)
        self.node_type_embedding = torch.nn.Embedding(100, 32)
        self.edge_type_embedding = torch.nn.Embedding(100, 32)

    def forward(self, x, edge_index, edge_attr, pos_enc, pos_index, pos_batch, batch):
        h = self.node_type_embedding(x)
        z = global_add_pool(self.z_initial.weight[pos_index] * pos_enc.view(-1, 1), pos_batch)
        z = torch.cat((self.z_embedding(z), self.edge_type_embedding(edge_attr)), dim=-1)
        h = self.conv1(h, edge_index, z)
        xs = [h]
        for conv in self.convs:
            h = conv(h, edge_index, z)
            xs.append(h)
        o = global_add_pool(torch.cat(xs, dim=1), batch)
        o = self.lin1(o)
        if o.size(0) > 1:
            o = self.bn_lin1(o)
        return self.lin2(F.elu(o))


# ---- OGB gin_eff route (ogb_mol_gnn.py:264-282, 323-358, 614-792, 66-261) -------------------------------------
ATOM_FEATURE_DIMS = (119, 5, 12, 12, 10, 6, 6, 2, 2)   # ogb==1.3.3 get_atom_feature_dims(), recalled (un-vendored)
BOND_FEATURE_DIMS = (5, 6, 2)                          # ogb==1.3.3 get_bond_feature_dims()


class MessagePassing(torch.nn.Module):
    """PyG MessagePassing(aggr='add') reduced to what GINConv_eff uses: x_j = x[edge_index[0]], message(),
    scatter-add at edge_index[1], update()."""

    def __init__(self, aggr="add"):
        super().__init__()
        assert aggr == "add"

    def propagate(self, edge_index, x, edge_attr):
        msg = self.message(x_j=x.index_select(0, edge_index[0]), edge_attr=edge_attr)
        return self.update(torch.zeros_like(x).index_add_(0, edge_index[1], msg))


class BondEncoder(torch.nn.Module):
    """ogb.graphproppred.mol_encoder.BondEncoder restated (sum of xavier-initialised embeddings)."""

    def __init__(self, emb_dim):
        super().__init__()
        self.bond_embedding_list = torch.nn.ModuleList()
        for d in BOND_FEATURE_DIMS:
            emb = torch.nn.Embedding(d, emb_dim)
            torch.nn.init.xavier_uniform_(emb.weight.data)
            self.bond_embedding_list.append(emb)

    def forward(self, edge_attr):
        out = 0
        for i in range(edge_attr.shape[1]):
            out = out + self.bond_embedding_list[i](edge_attr[:, i])
        return out


class AtomEncoderRef(torch.nn.Module):
    def __init__(self, emb_dim):
        super().__init__()
        self.atom_embedding_list = torch.nn.ModuleList()
        for d in ATOM_FEATURE_DIMS:
            emb = torch.nn.Embedding(d, emb_dim)
            torch.nn.init.xavier_uniform_(emb.weight.data)
            self.atom_embedding_list.append(emb)

    def forward(self, x):
        out = 0
        for i in range(x.shape[1]):
            out = out + self.atom_embedding_list[i](x[:, i])
        return out


class GINConvEffRef(MessagePassing):
    def __init__(self, emb_dim):
        super().__init__("add")
        self.mlp = Sequential(Linear(emb_dim, 2 * emb_dim), BatchNorm1d(2 * emb_dim), ReLU(), Linear(2 * emb_dim, emb_dim))
        self.eps = torch.nn.Parameter(torch.Tensor([0]))
        self.edge_encoder = BondEncoder(emb_dim)
        self.edge_encoder_pos = Linear(emb_dim, emb_dim)

    def forward(self, x, edge_index, edge_attr, edge_pos):
        e = self.edge_encoder(edge_attr) + self.edge_encoder_pos(edge_pos)
        return self.mlp((1 + self.eps) * x + self.propagate(edge_index, x=x, edge_attr=e))

    def message(self, x_j, edge_attr):
        return F.relu(x_j + edge_attr)

    def update(self, aggr_out):
        return aggr_out


class GNNEffRef(torch.nn.Module):
    """GNN(gnn_type='gin_eff') with sum/mean pooling: same state_dict keys as the reference wrapper."""

    class Node(torch.nn.Module):
        def __init__(self, num_layer, emb_dim, drop_ratio, JK, residual, virtual_node):
            super().__init__()
            self.num_layer, self.drop_ratio, self.JK, self.residual, self.virtual_node = num_layer, drop_ratio, JK, residual, virtual_node
            self.z_initial = torch.nn.Embedding(1800, emb_dim)
            self.z_embedding = Sequential(Dropout(drop_ratio), BatchNorm1d(emb_dim), ReLU(), Linear(emb_dim, emb_dim),
                                          Dropout(drop_ratio), BatchNorm1d(emb_dim), ReLU())
            self.node_encoder = AtomEncoderRef(emb_dim)
            if virtual_node:
                self.virtualnode_embedding = torch.nn.Embedding(1, emb_dim)
                torch.nn.init.constant_(self.virtualnode_embedding.weight.data, 0)
            self.convs = torch.nn.ModuleList()
            self.batch_norms = torch.nn.ModuleList()
            for _ in range(num_layer):
                self.convs.append(GINConvEffRef(emb_dim))
                self.batch_norms.append(BatchNorm1d(emb_dim))
            if virtual_node:
                self.mlp_virtualnode_list = torch.nn.ModuleList()
                for _ in range(num_layer - 1):
                    self.mlp_virtualnode_list.append(Sequential(
                        Linear(emb_dim, 2 * emb_dim), BatchNorm1d(2 * emb_dim), ReLU(),
                        Linear(2 * emb_dim, emb_dim), BatchNorm1d(emb_dim), ReLU()))

        def forward(self, x, edge_index, edge_attr, batch, pos_enc, pos_index, pos_batch):
            B = int(batch[-1]) + 1
            vn = self.virtualnode_embedding(torch.zeros(B, dtype=torch.long)) if self.virtual_node else None
            h_list = [self.node_encoder(x)]
            z = self.z_embedding(global_add_pool(self.z_initial.weight[pos_index] * pos_enc.view(-1, 1), pos_batch))
            for layer in range(self.num_layer):
                if self.virtual_node:
                    h_list[layer] = h_list[layer] + vn[batch]
                h = self.batch_norms[layer](self.convs[layer](h_list[layer], edge_index, edge_attr, z))
                if layer == self.num_layer - 1:
                    h = F.dropout(h, self.drop_ratio, training=self.training)
                else:
                    h = F.dropout(F.relu(h), self.drop_ratio, training=self.training)
                if self.residual:
                    h = h + h_list[layer]
                h_list.append(h)
                if self.virtual_node and layer < self.num_layer - 1:
                    tmp = global_add_pool(h_list[layer], batch, B) + vn
                    upd = F.dropout(self.mlp_virtualnode_list[layer](tmp), self.drop_ratio, training=self.training)
                    vn = vn + upd if self.residual else upd
            if self.JK == "last":
                return h_list[-1]
            out = 0
            for layer in range(self.num_layer):
                out = out + h_list[layer]
            return out

    def __init__(self, num_tasks, num_layer, emb_dim, virtual_node=True, residual=False, drop_ratio=0.0, JK="last",
                 graph_pooling="mean"):
        super().__init__()
        self.gnn_node = GNNEffRef.Node(num_layer, emb_dim, drop_ratio, JK, residual, virtual_node)
        self.pool = global_add_pool if graph_pooling == "sum" else global_mean_pool
        self.graph_pred_linear = Linear(emb_dim, num_tasks)

    def forward(self, x, edge_index, edge_attr, batch, pos_enc, pos_index, pos_batch):
        h = self.gnn_node(x, edge_index, edge_attr, batch, pos_enc, pos_index, pos_batch)
        return self.graph_pred_linear(self.pool(h, batch))


def train_step(model, optimizer, b):
    """One optimisation step as run_graphcount.py:494-505 (L1 loss, mean over nodes)."""
    optimizer.zero_grad()
    pred = model(b["x"], b["edge_index"], b["pos_enc"], b["pos_index"], b["pos_batch"], b.get("batch"))
    loss = F.l1_loss(pred, b["y"].view(-1, 1))
    loss.backward()
    optimizer.step()
    return loss.detach()

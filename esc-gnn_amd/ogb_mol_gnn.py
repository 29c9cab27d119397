"""OGB molecule models on the ESC hot path — the MI355X twin of the `gin_eff` route of
/root/reference/ogb_mol_gnn.py: AtomEncoder (:264-282), GINConv_eff (:323-358), GNN_node_efficient (:614-792)
and the GNN wrapper (:66-261) with sum/mean graph pooling (BASELINE config 5; selected by `--gnn gin_eff`,
run_ogb_mol.py:403-404).  Same state_dict key layout as the reference classes.

Embedding tables sizes come from the un-vendored `ogb==1.3.3` (requirements.txt:54):
`get_atom_feature_dims()` / `get_bond_feature_dims()` — recalled, unverifiable offline (parity unpinned there).
The aggregate is the same GINE primitive (csrc/aggregate.hip); BatchNorm/ReLU, Linear, embeddings and the
virtual-node pooling / broadcast run through the HIP kernels.  Dropout (drop_ratio, default 0.5) uses torch's
device RNG like the reference; plain elementwise adds (virtual-node / residual sums) are torch ops.
"""
import torch
import torch.nn.functional as F
from torch.nn import Dropout, Sequential

from . import ops
from .nn import AbsorbedReLU, BatchNorm1d, Embedding, Linear, global_add_pool, global_mean_pool
from .plan import plan_of
from .run_graphcount import Z_TABLE_ROWS, _bn_relu

ATOM_FEATURE_DIMS = (119, 5, 12, 12, 10, 6, 6, 2, 2)     # ogb.utils.features.get_atom_feature_dims() @1.3.3
BOND_FEATURE_DIMS = (5, 6, 2)                            # ogb.utils.features.get_bond_feature_dims() @1.3.3


class _SumOfEmbeddings(torch.nn.Module):
    def _build(self, dims, emb_dim, attr):
        tables = torch.nn.ModuleList()
        for d in dims:
            emb = Embedding(d, emb_dim)
            torch.nn.init.xavier_uniform_(emb.weight.data)
            tables.append(emb)
        setattr(self, attr, tables)
        self._tables = attr

    def forward(self, x):
        tables = getattr(self, self._tables)
        if x.dim() != 2 or x.shape[1] > len(tables):
            raise ValueError("expected integer features [n, <=%d]" % len(tables))
        return ops.embedding_sum([t.weight for t in tables[:x.shape[1]]], x)       # one launch for all columns


class AtomEncoder(_SumOfEmbeddings):
    def __init__(self, emb_dim):
        super().__init__()
        self._build(ATOM_FEATURE_DIMS, emb_dim, "atom_embedding_list")


class BondEncoder(_SumOfEmbeddings):
    def __init__(self, emb_dim):
        super().__init__()
        self._build(BOND_FEATURE_DIMS, emb_dim, "bond_embedding_list")


class GINConv_eff(torch.nn.Module):
    """out = mlp((1+eps) x + sum_j relu(x_j + bond(edge_attr) + edge_encoder_pos(z_emb)))  (reference :346-358)"""

    def __init__(self, dataset, emb_dim):
        super().__init__()
        self.mlp = Sequential(Linear(emb_dim, 2 * emb_dim), BatchNorm1d(2 * emb_dim, fuse_relu=True), AbsorbedReLU(),
                              Linear(2 * emb_dim, emb_dim))
        self.eps = torch.nn.Parameter(torch.Tensor([0]))
        if dataset.startswith("ogbg-mol"):
            self.edge_encoder = BondEncoder(emb_dim=emb_dim)
        elif dataset.startswith("ogbg-ppa"):
            self.edge_encoder = Linear(7, emb_dim)
        self.edge_encoder_pos = Linear(emb_dim, emb_dim)

    def forward(self, x, edge_index, edge_attr, edge_pos, plan=None):
        if plan is None:
            from .plan import BatchPlan
            plan = BatchPlan.from_tensors(edge_index, x.size(0))
        e = self.edge_encoder(edge_attr) + self.edge_encoder_pos(edge_pos)
        return self.mlp(ops.gine_aggregate(x, e, self.eps, plan))


class GNN_node_efficient(torch.nn.Module):
    def __init__(self, dataset, num_layer, emb_dim, drop_ratio=0.5, JK="last", residual=False, gnn_type="gin",
                 virtual_node=True, use_rd=False, adj_dropout=0, skip_node_encoder=False, use_rp=None,
                 center_pool_virtual=False, RNI=False):
        super().__init__()
        if center_pool_virtual or RNI or gnn_type not in ("gin", "gin_eff"):
            raise NotImplementedError("only the gin_eff route without center pooling / RNI is on the ESC hot path")
        self.num_layer, self.drop_ratio, self.JK, self.residual = num_layer, drop_ratio, JK, residual
        self.virtual_node, self.use_rd, self.use_rp, self.adj_dropout = virtual_node, use_rd, use_rp, adj_dropout
        self.center_pool_virtual, self.RNI, self.skip_node_encoder = center_pool_virtual, RNI, skip_node_encoder
        dropout = drop_ratio
        self.z_initial = torch.nn.Embedding(Z_TABLE_ROWS, emb_dim)
        self.z_embedding = Sequential(Dropout(dropout), *_bn_relu(emb_dim), Linear(emb_dim, emb_dim), Dropout(dropout),
                                      *_bn_relu(emb_dim))
        if not skip_node_encoder:
            if dataset.startswith("ogbg-mol"):
                self.node_encoder = AtomEncoder(emb_dim)
            elif dataset.startswith("ogbg-ppa"):
                self.node_encoder = Embedding(1, emb_dim)
        if virtual_node:
            self.virtualnode_embedding = Embedding(1, emb_dim)
            torch.nn.init.constant_(self.virtualnode_embedding.weight.data, 0)
        self.convs = torch.nn.ModuleList(GINConv_eff(dataset, emb_dim) for _ in range(num_layer))
        self.batch_norms = torch.nn.ModuleList(
            BatchNorm1d(emb_dim, fuse_relu=(layer != num_layer - 1)) for layer in range(num_layer))   # no relu on the last
        if virtual_node:
            self.mlp_virtualnode_list = torch.nn.ModuleList(
                Sequential(Linear(emb_dim, 2 * emb_dim), BatchNorm1d(2 * emb_dim, fuse_relu=True), AbsorbedReLU(),
                           Linear(2 * emb_dim, emb_dim), BatchNorm1d(emb_dim, fuse_relu=True), AbsorbedReLU())
                for _ in range(num_layer - 1))

    def forward(self, batched_data, x=None, edge_index=None, edge_attr=None, batch=None, perturb=None):
        if batched_data is not None:
            x, edge_index, edge_attr, batch = (batched_data.x, batched_data.edge_index, batched_data.edge_attr,
                                               batched_data.batch)
        dev = self.z_initial.weight.device
        if edge_index.device != dev:     # the reference expects batch.to(device) from its loop (run_ogb_mol.py:58)
            batched_data.to(dev)
            x, edge_index, edge_attr, batch = (batched_data.x, batched_data.edge_index, batched_data.edge_attr,
                                               batched_data.batch)
        plan = plan_of(batched_data, Z_TABLE_ROWS)
        ng = batched_data.__dict__.get("_num_graphs") if batched_data is not None else None    # device collate: known on the host
        num_graphs = ng if ng is not None else int(batch[-1].item()) + 1
        if self.virtual_node:
            vn = self.virtualnode_embedding(torch.zeros(num_graphs, dtype=edge_index.dtype, device=edge_index.device))
        h0 = x if self.skip_node_encoder else self.node_encoder(x)
        if "edge_pos" in batched_data:
            z = ops.linear(batched_data.edge_pos.float(), self.z_initial.weight.t().contiguous())
        else:
            z = ops.esc_bag(self.z_initial.weight, plan)
        z = self.z_embedding(z)
        h_list = [h0]
        if perturb is not None:
            h_list[0] = h_list[0] + perturb
        for layer in range(self.num_layer):
            if self.virtual_node:
                h_list[layer] = h_list[layer] + ops.segment_broadcast(vn, batch, num_graphs)    # vn[batch]
            h = self.convs[layer](h_list[layer], edge_index, edge_attr, z, plan)
            h = self.batch_norms[layer](h)                 # ReLU fused except on the last layer (reference :747-752)
            h = F.dropout(h, self.drop_ratio, training=self.training)
            if self.residual:
                h = h + h_list[layer]
            h_list.append(h)
            if self.virtual_node and layer < self.num_layer - 1:
                tmp = global_add_pool(h_list[layer], batch, num_graphs) + vn
                upd = F.dropout(self.mlp_virtualnode_list[layer](tmp), self.drop_ratio, training=self.training)
                vn = vn + upd if self.residual else upd
        if self.JK == "last":
            return h_list[-1]
        out = 0
        for layer in range(self.num_layer):             # reference "sum" JK skips the last representation (:785-788)
            out = out + h_list[layer]
        return out


class GNN(torch.nn.Module):
    def __init__(self, dataset, num_tasks, num_layer=5, emb_dim=300, gnn_type="gin", virtual_node=True, residual=False,
                 drop_ratio=0.5, JK="last", graph_pooling="mean", subgraph_pooling="mean", use_rd=False, use_rp=None,
                 RNI=False, deg_sub=None, deg_graph=None, **kwargs):
        super().__init__()
        if gnn_type != "gin_eff":
            raise NotImplementedError("only --gnn gin_eff is the ESC path (run_ogb_mol.py:403-404); other types are baselines")
        self.num_layer, self.drop_ratio, self.JK, self.emb_dim, self.num_tasks = num_layer, drop_ratio, JK, emb_dim, num_tasks
        self.graph_pooling, self.subgraph_pooling = graph_pooling, subgraph_pooling
        self.gnn_node = GNN_node_efficient(dataset, num_layer, emb_dim, JK=JK, drop_ratio=drop_ratio, residual=residual,
                                           gnn_type=gnn_type, virtual_node=virtual_node, use_rd=use_rd, use_rp=use_rp,
                                           RNI=RNI)
        if graph_pooling == "sum":
            self.pool = global_add_pool
        elif graph_pooling == "mean":
            self.pool = global_mean_pool
        else:
            raise NotImplementedError("graph_pooling %r: only sum / mean are on the ESC hot path" % graph_pooling)
        self.graph_pred_linear = Linear(emb_dim, num_tasks)
        self.step_engine = True       # training-mode forward through the whole-step engine when the batch allows it

    def forward(self, data, perturb=None):
        if self.training and torch.is_grad_enabled() and self.step_engine and perturb is None:
            from .engine import ogb_engine_forward, ogb_engine_ready
            if data.edge_index.device != self.graph_pred_linear.weight.device:
                data.to(self.graph_pred_linear.weight.device)
            if ogb_engine_ready(self, data):
                return ogb_engine_forward(self, data)      # one autograd node (csrc/engine.hip esc_ogb_*)
        if not self.training and not torch.is_grad_enabled() and self.step_engine and perturb is None:
            from .engine import ogb_engine_predict, ogb_engine_ready
            if data.edge_index.device != self.graph_pred_linear.weight.device:
                data.to(self.graph_pred_linear.weight.device)
            if ogb_engine_ready(self, data):
                return ogb_engine_predict(self, data)      # eval-mode forward as one call (esc_ogb_predict)
        x = self.gnn_node(data, perturb=perturb)
        return self.graph_pred_linear(self.pool(x, data.batch))

"""NestedGIN_eff as used by the expressiveness drivers (run_sr.py:22, run_exp.py:21) — the MI355X twin of
/root/reference/kernel/gin.py:200-379.  Differs from the counting model (run_graphcount.py) by: node input
width = dataset.num_features, no x_embedding, lin1 over num_layers*hidden, dropout applied BEFORE the final
ReLU (:368-374), graph-level readout by global_mean_pool by default, log_softmax head unless use_cycle.
Same state_dict key layout as the reference class."""
import torch
import torch.nn.functional as F
from torch.nn import Dropout, Sequential

from . import ops
from .nn import AbsorbedReLU, BatchNorm1d, GINEConv, Linear, global_mean_pool
from .plan import plan_of
from .run_graphcount import Z_TABLE_ROWS, _bn_relu, _mlp


class NestedGIN_eff(torch.nn.Module):
    def __init__(self, dataset, num_layers, hidden, use_z=False, use_rd=False, use_cycle=False, graph_pred=True,
                 use_id=None, dropout=0.2, multi_layer=False, edge_nest=False):
        super().__init__()
        if use_id is not None:
            raise NotImplementedError("use_id (GINIDConvLayer) is the identity-aware baseline, outside the ESC hot path")
        self.use_rd, self.use_z, self.graph_pred, self.use_cycle = use_rd, True, graph_pred, use_cycle
        self.use_id, self.dropout, self.multi_layer, self.edge_nest = use_id, dropout, multi_layer, edge_nest
        self.z_initial = torch.nn.Embedding(Z_TABLE_ROWS, hidden)
        self.z_embedding = Sequential(Dropout(dropout), *_bn_relu(hidden), Linear(hidden, hidden),
                                      Dropout(dropout), *_bn_relu(hidden))
        input_dim = dataset.num_features
        self.conv1 = GINEConv(_mlp(input_dim, hidden, dropout), train_eps=True, edge_dim=hidden)
        self.convs = torch.nn.ModuleList(
            GINEConv(_mlp(hidden, hidden, dropout), train_eps=True, edge_dim=hidden) for _ in range(num_layers - 1))
        self.lin1 = Linear(num_layers * hidden, hidden)
        self.bn_lin1 = BatchNorm1d(hidden, eps=1e-5, momentum=0.1)
        self.lin2 = Linear(hidden, 1 if use_cycle else dataset.num_classes)

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

    def forward(self, data):
        data.to(self.lin1.weight.device)
        x, edge_index, batch = data.x, data.edge_index, data.batch
        plan = plan_of(data, Z_TABLE_ROWS)
        if "edge_pos" in data:
            z = ops.linear(data.edge_pos.float(), self.z_initial.weight.t().contiguous())
        else:
            z = ops.esc_bag(self.z_initial.weight, plan)
        z = self.z_embedding(z)
        h = self.conv1(x.float(), edge_index, z, plan)
        xs = [h]
        for conv in self.convs:
            h = conv(h, edge_index, z, plan)
            xs.append(h)
        o = torch.cat(xs, dim=1)
        if self.graph_pred:
            o = global_mean_pool(o, batch)
        o = self.lin1(o)
        if o.size(0) > 1:
            o = self.bn_lin1(o)
        o = F.relu(F.dropout(o, p=self.dropout, training=self.training))     # dropout BEFORE relu (:371-372)
        o = self.lin2(o)
        return o if self.use_cycle else F.log_softmax(o, dim=-1)

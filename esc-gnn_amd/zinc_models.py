"""ZINC NestedGIN_eff — the MI355X twin of /root/reference/zinc_models.py:504-611 (BASELINE config 4).
ELU activations (fused into the HIP BatchNorm kernels), 32-wide node/edge type embeddings (looked up through
the ESC bag kernels), edge term = [z_emb | edge_type_embedding] (edge_dim = 256 + 32), global_add_pool
readout (HIP segment-pool), dropout 0.  Same constructor and state_dict key layout as the reference class."""
import torch
import torch.nn.functional as F
from torch.nn import Dropout, Sequential

from . import ops
from .nn import AbsorbedELU, BatchNorm1d, Embedding, GINEConv, Linear, global_add_pool
from .plan import plan_of
from .run_graphcount import Z_TABLE_ROWS


def _bn_elu(hidden):
    return BatchNorm1d(hidden, fuse_relu="elu"), AbsorbedELU()


def _mlp_elu(n_in, hidden, p):
    return Sequential(Linear(n_in, hidden), Dropout(p), *_bn_elu(hidden), Linear(hidden, hidden), Dropout(p), *_bn_elu(hidden))


class NestedGIN_eff(torch.nn.Module):
    def __init__(self, dataset, num_layers, concat=False, use_pos=False, use_max_dist=False, RNI=False, **kwargs):
        super().__init__()
        self.use_z = True
        self.step_engine = True       # training-mode forward through the whole-step engine when the batch allows it
        hidden, dropout = 256, 0.0
        self.dropout = dropout
        self.z_initial = torch.nn.Embedding(Z_TABLE_ROWS, hidden)
        self.z_embedding = Sequential(Dropout(dropout), *_bn_elu(hidden), Linear(hidden, hidden), Dropout(dropout),
                                      *_bn_elu(hidden))
        input_dim, edge_attr_dim = 32, 32
        self.conv1 = GINEConv(_mlp_elu(input_dim, hidden, dropout), train_eps=True, edge_dim=hidden + edge_attr_dim)
        self.convs = torch.nn.ModuleList(
            GINEConv(_mlp_elu(hidden, hidden, dropout), train_eps=True, edge_dim=hidden + edge_attr_dim)
            for _ in range(num_layers - 1))
        self.lin1 = Linear(num_layers * hidden, hidden)
        self.bn_lin1 = BatchNorm1d(hidden, eps=1e-5, momentum=0.1, fuse_relu="elu")   # dropout is 0 => ELU follows BN
        self.lin2 = Linear(hidden, 1)
        self.node_type_embedding = Embedding(100, 32)
        self.edge_type_embedding = Embedding(100, 32)

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
        self.node_type_embedding.reset_parameters()
        self.edge_type_embedding.reset_parameters()

    def forward(self, data):
        data.to(self.lin1.weight.device)
        if self.training and torch.is_grad_enabled() and self.step_engine:
            from .engine import zinc_engine_forward, zinc_engine_ready
            if zinc_engine_ready(self, data):
                return zinc_engine_forward(self, data)     # the whole step as one autograd node (csrc/engine.hip esc_zinc_*)
        if not self.training and not torch.is_grad_enabled() and self.step_engine:
            from .engine import zinc_engine_predict, zinc_engine_ready
            if zinc_engine_ready(self, data):
                return zinc_engine_predict(self, data)     # eval-mode forward as one call (esc_zinc_predict)
        x, edge_index, batch = self.node_type_embedding(data.x.view(-1)), data.edge_index, data.batch
        plan = plan_of(data, Z_TABLE_ROWS)
        if "edge_pos" in data:
            z = ops.linear(data.edge_pos.float(), self.z_initial.weight.t().contiguous())
        else:
            z = ops.esc_bag(self.z_initial.weight, plan)
        z = self.z_embedding(z)
        z = torch.cat((z, self.edge_type_embedding(data.edge_attr.view(-1))), dim=-1)
        h = self.conv1(x, edge_index, z, plan)
        xs = [h]
        for conv in self.convs:
            h = conv(h, edge_index, z, plan)
            xs.append(h)
        o = global_add_pool(torch.cat(xs, dim=1), batch)
        o = self.lin1(o)
        o = self.bn_lin1(o) if o.size(0) > 1 else F.elu(o)      # reference :606-609 (dropout p = 0)
        return self.lin2(o)

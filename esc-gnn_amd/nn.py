"""torch.nn-compatible layers whose arithmetic runs in libescgnn_hip.so.

`Linear` keeps torch.nn.Linear's parameters / init / state_dict keys but multiplies on the fp32
matrix cores; `GINEConv` mirrors the constructor and state_dict layout of PyG 2.0.4's GINEConv as
the reference instantiates it (run_graphcount.py:77-89,97-109: nn=Sequential(...), train_eps=True,
edge_dim=hidden -> parameters `eps`, `nn.*`, `lin.{weight,bias}`).
"""
import torch

from . import ops
from .plan import plan_of


class Linear(torch.nn.Linear):
    def forward(self, x):
        lead = x.shape[:-1]
        y = ops.linear(x.reshape(-1, x.shape[-1]), self.weight, self.bias)
        return y.view(*lead, self.out_features)


class BatchNorm1d(torch.nn.BatchNorm1d):
    """torch.nn.BatchNorm1d (same parameters/buffers/state_dict) computed by the HIP norm kernels.
    `fuse_relu=True` also applies the ReLU that follows it in the reference's Sequential (the
    ReLU module stays in place as `AbsorbedReLU` so the child indices / checkpoint keys match)."""

    ACT = {None: 0, False: 0, "none": 0, True: 1, "relu": 1, "elu": 2}

    def __init__(self, num_features, eps=1e-5, momentum=0.1, fuse_relu=False):
        """fuse_relu: False/None | True/'relu' | 'elu' — the activation module that follows in the reference."""
        super().__init__(num_features, eps=eps, momentum=momentum)
        self.fuse_relu = self.ACT[fuse_relu]
        self.sync_group = False        # False: per-rank statistics; None / a ProcessGroup: statistics over that group

    @staticmethod
    def convert_sync(module, group=None):
        """Switch every esc BatchNorm1d under `module` to statistics over all ranks of `group` (default group if None)
        — the counterpart of torch.nn.SyncBatchNorm.convert_sync_batchnorm for graph-sharded data parallelism."""
        for m in module.modules():
            if isinstance(m, BatchNorm1d):
                m.sync_group = group
        return module

    def forward(self, x):
        if x.dim() != 2:
            raise ValueError("expected 2D input (got {}D input)".format(x.dim()))
        if self.training:
            if self.num_batches_tracked is not None:
                self.num_batches_tracked.add_(1)
            if self.sync_group is not False:
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.sync_group) > 1:
                    return ops.sync_batch_norm_act(x, self.weight, self.bias, self.running_mean, self.running_var,
                                                   self.eps, self.momentum, self.fuse_relu, self.sync_group)
            return ops.batch_norm_act(x, self.weight, self.bias, self.running_mean, self.running_var,
                                      self.eps, self.momentum, self.fuse_relu)
        return ops.bn_eval_act(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                               self.fuse_relu)


class AbsorbedReLU(torch.nn.ReLU):
    """Placeholder for a ReLU whose work is fused into the preceding BatchNorm1d(fuse_relu=True)."""

    def forward(self, x):
        return x


class AbsorbedELU(torch.nn.ELU):
    """Placeholder for an ELU fused into the preceding BatchNorm1d(fuse_relu='elu')."""

    def forward(self, x):
        return x


class Embedding(torch.nn.Embedding):
    """torch.nn.Embedding whose lookup and gradient run through the ESC bag kernels (one entry of weight 1 per row)."""

    def forward(self, index):
        return ops.embedding(self.weight, index)


class GINEConv(torch.nn.Module):
    """out = nn( sum_{j->i} relu(x_j + lin(e_ji)) + (1 + eps) * x_i )"""

    def __init__(self, nn, eps=0.0, train_eps=False, edge_dim=None):
        super().__init__()
        self.nn = nn
        self.initial_eps = eps
        if train_eps:
            self.eps = torch.nn.Parameter(torch.Tensor([eps]))
        else:
            self.register_buffer("eps", torch.Tensor([eps]))
        first = nn[0]
        in_channels = first.in_features if hasattr(first, "in_features") else first.in_channels
        self.lin = Linear(edge_dim, in_channels) if edge_dim is not None else None

    def reset_parameters(self):
        for m in self.nn.modules():
            if m is not self.nn and hasattr(m, "reset_parameters"):
                m.reset_parameters()
        self.eps.data.fill_(self.initial_eps)
        if self.lin is not None:
            self.lin.reset_parameters()

    def forward(self, x, edge_index, edge_attr=None, plan=None):
        if plan is None:
            from .plan import BatchPlan
            plan = BatchPlan.from_tensors(edge_index, x.size(0))
        if self.lin is None and x.size(-1) != edge_attr.size(-1):
            raise ValueError("Node and edge feature dimensionalities do not match. "
                             "Consider setting the 'edge_dim' attribute of 'GINEConv'")
        e = self.lin(edge_attr) if self.lin is not None else edge_attr
        return self.nn(ops.gine_aggregate(x, e, self.eps, plan))

    def __repr__(self):
        return "{}(nn={})".format(self.__class__.__name__, self.nn)


def global_add_pool(x, batch, size=None):
    """PyG global_add_pool: segment sum by graph id (HIP segment_pool kernels)."""
    return ops.segment_pool(x, batch, size, mean=False)


def global_mean_pool(x, batch, size=None):
    """PyG global_mean_pool: segment sum / max(count, 1)."""
    return ops.segment_pool(x, batch, size, mean=True)

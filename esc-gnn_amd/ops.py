"""Autograd bindings of the HIP hot-path kernels (ctypes -> libescgnn_hip.so, C ABI in
include/escgnn_hip.h).  Each Function names the reference call site it replaces.  There is no
CPU or PyTorch fallback: a CPU tensor raises.
"""
import ctypes

import torch
from torch.autograd import Function

from . import _native as nv


def _dev(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("esc_gnn_amd: the hot path runs on the HIP device only (got a %s tensor); "
                               "there is no CPU fallback" % t.device)
        if t.dtype != torch.float32:
            raise TypeError("esc_gnn_amd: expected float32, got %s" % t.dtype)


def _on(dev, *tensors):
    """Every pointer handed to a kernel must be a DEVICE pointer on the same GPU: an index tensor left on the host
    would be dereferenced by the GPU and fault.  Raise instead."""
    for t in tensors:
        if t is not None and t.device != dev:
            raise RuntimeError("esc_gnn_amd: tensor on %s but the kernel runs on %s — move the batch to the device "
                               "first (data.to(device))" % (t.device, dev))


def _plan_on(dev, plan, *fields):
    for f in fields:
        _on(dev, getattr(plan, f))


def _rows(t):
    """2-D, unit inner stride view + its leading dimension."""
    if t.dim() != 2:
        raise ValueError("expected a 2-D tensor, got shape %s" % (tuple(t.shape),))
    if t.stride(1) != 1 or (t.size(0) > 1 and t.stride(0) < t.size(1)):
        t = t.contiguous()
    return t, (t.stride(0) if t.size(0) > 1 else t.size(1))


class _Bag(Function):
    """z_emb = global_add_pool(z_initial.weight[pos_index] * pos_enc[:,None], pos_batch)
    (run_graphcount.py:155) as a CSR SpMM; backward = deterministic CSC segmented sum."""

    @staticmethod
    def forward(ctx, table, plan):
        _dev(table)
        table = table.contiguous()
        if plan.row_ptr is None:
            raise ValueError("batch has no pos_enc/pos_index/pos_batch")
        _plan_on(table.device, plan, "row_ptr", "bag_idx", "bag_val", "col_ptr", "col_row", "col_val", "col_col")
        E, H = plan.num_edges, table.size(1)
        out = torch.empty((E, H), dtype=torch.float32, device=table.device)
        nv.call("esc_bag_fwd_rows", nv.ptr(table), table.size(0), H, nv.ptr(plan.row_ptr), nv.ptr(plan.bag_idx),
                nv.ptr(plan.bag_val), E, nv.ptr(out), H, 0, None, nv.stream())
        ctx.plan, ctx.shape = plan, tuple(table.shape)
        return out

    @staticmethod
    def backward(ctx, dz):
        plan = ctx.plan
        rows, H = ctx.shape
        dz, ld = _rows(dz)
        dtable = torch.empty((rows, H), dtype=torch.float32, device=dz.device)
        scratch = torch.empty(max(1, nv.lib().esc_bag_bwd_scratch(plan.nnz, H)), dtype=torch.float32, device=dz.device)
        nv.call("esc_bag_bwd_table_rows", nv.ptr(dz), ld, H, nv.ptr(plan.col_ptr), nv.ptr(plan.col_row),
                nv.ptr(plan.col_val), nv.ptr(plan.col_col), plan.nnz, rows, plan.num_edges, 0, nv.ptr(dtable),
                nv.ptr(scratch), nv.stream())
        return dtable, None


def esc_bag(table, plan):
    return _Bag.apply(table, plan)


class _GineAggregate(Function):
    """out = sum_{k: dst_k=i} relu(x[src_k] + e_k) + (1+eps) x_i — PyG GINEConv propagate + self term
    (run_graphcount.py:161,169; gine_conv_layer.py:56-84)."""

    @staticmethod
    def forward(ctx, x, e, eps, plan):
        _dev(x, e, eps)
        _on(x.device, e, eps)
        _plan_on(x.device, plan, "in_ptr", "in_edge", "in_src", "out_ptr", "out_edge", "out_dst")
        x, ldx = _rows(x)
        e, lde = _rows(e)
        N, C = x.shape
        if e.shape != (plan.num_edges, C) or N != plan.num_nodes:
            raise ValueError("aggregate: x %s / e %s do not match the batch (N=%d, E=%d)"
                             % (tuple(x.shape), tuple(e.shape), plan.num_nodes, plan.num_edges))
        out = torch.empty((N, C), dtype=torch.float32, device=x.device)
        nv.call("esc_gine_aggregate_fwd", nv.ptr(x), ldx, nv.ptr(e), lde, nv.ptr(plan.in_ptr),
                nv.ptr(plan.in_edge), nv.ptr(plan.in_src), nv.ptr(eps), N, C, nv.ptr(out), C, nv.stream())
        ctx.save_for_backward(x, e, eps)
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, g):
        x, e, eps = ctx.saved_tensors
        plan = ctx.plan
        g, ldg = _rows(g)
        N, C = x.shape
        need_dx, need_de, need_eps = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        d_e = torch.empty_like(e)
        dx = torch.empty((N, C), dtype=torch.float32, device=x.device) if need_dx else None
        slots = int(nv.lib().esc_gine_aggregate_bwd_deps_slots(C))       # partial dot products per row (see the header)
        part = torch.empty(N * slots, dtype=torch.float32, device=x.device) if need_eps else None
        nv.call("esc_gine_aggregate_bwd", nv.ptr(x), x.stride(0), nv.ptr(e), e.stride(0), nv.ptr(g), ldg,
                nv.ptr(plan.out_ptr), nv.ptr(plan.out_edge), nv.ptr(plan.out_dst), nv.ptr(eps), N, C,
                nv.ptr(d_e), d_e.stride(0), nv.ptr(dx), C, 0, nv.ptr(part), nv.stream())
        deps = None
        if need_eps:
            deps = torch.empty(1, dtype=torch.float32, device=x.device)
            nv.call("esc_reduce_sum", nv.ptr(part), N * slots, nv.ptr(deps), nv.stream())
        return dx, (d_e if need_de else None), deps, None


def gine_aggregate(x, e, eps, plan):
    return _GineAggregate.apply(x, e, eps, plan)


class _Linear(Function):
    """torch.nn.Linear on the fp32 matrix cores (every Linear of run_graphcount.py:54-121,183-189)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _dev(x, weight, bias)
        _on(x.device, weight, bias)
        x, ldx = _rows(x)
        weight = weight.contiguous()
        M, K = x.shape
        N = weight.size(0)
        if weight.size(1) != K:
            raise ValueError("linear: x %s vs weight %s" % (tuple(x.shape), tuple(weight.shape)))
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        nv.call("esc_linear_fwd", nv.ptr(x), ldx, nv.ptr(weight), K, nv.ptr(bias), None, None, M, N, K,
                nv.ptr(y), N, None, nv.stream())
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy, ldy = _rows(dy)
        M, K = x.shape
        N = weight.size(0)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=torch.float32, device=x.device)
            nv.call("esc_linear_bwd_input", nv.ptr(dy), ldy, nv.ptr(weight), K, M, N, K, nv.ptr(dx), K, 0,
                    nv.stream())
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty((N, K), dtype=torch.float32, device=x.device)
            db = torch.empty(N, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            if M == 0:
                dw.zero_()
                if db is not None:
                    db.zero_()
            else:
                slabs = torch.empty(nv.lib().esc_linear_bwd_weight_scratch(M, N, K), dtype=torch.float32,
                                    device=x.device)
                nv.call("esc_linear_bwd_weight", nv.ptr(dy), ldy, nv.ptr(x), x.stride(0), None, None, M, N, K,
                        nv.ptr(dw), K, nv.ptr(db), nv.ptr(slabs), nv.stream())
        return dx, dw, db


def linear(x, weight, bias=None):
    return _Linear.apply(x, weight, bias)


class _BatchNormAct(Function):
    """Training-mode BatchNorm1d (+ optional fused ReLU): batch statistics, running-stat update and
    normalisation (run_graphcount.py:55-60,66-72,80-87,115)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, relu):
        _dev(x, gamma, beta)
        _on(x.device, gamma, beta, running_mean, running_var)
        x, ldx = _rows(x)
        M, C = x.shape
        if M <= 1:
            raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (tuple(x.shape),))
        dev = x.device
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        invstd = torch.empty(C, dtype=torch.float32, device=dev)
        scratch = torch.empty(nv.lib().esc_bn_scratch(C), dtype=torch.float32, device=dev)
        y = torch.empty((M, C), dtype=torch.float32, device=dev)
        s = nv.stream()
        nv.call("esc_bn_stats", nv.ptr(x), ldx, M, C, float(eps), float(momentum), nv.ptr(mean), nv.ptr(invstd),
                nv.ptr(running_mean), nv.ptr(running_var), None, None, None, None, nv.ptr(scratch), s)
        nv.call("esc_bn_apply", nv.ptr(x), ldx, M, C, nv.ptr(mean), nv.ptr(invstd), nv.ptr(gamma), nv.ptr(beta),
                int(relu), nv.ptr(y), C, s)
        ctx.save_for_backward(x, y if relu else None, gamma, mean, invstd)
        ctx.relu, ctx.scratch = int(relu), scratch        # 0 none, 1 relu, 2 elu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mean, invstd = ctx.saved_tensors
        dy, ldg = _rows(dy)
        M, C = x.shape
        dx = torch.empty((M, C), dtype=torch.float32, device=x.device)
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
        nv.call("esc_bn_bwd", nv.ptr(x), x.stride(0), nv.ptr(y), C, nv.ptr(dy), ldg, M, C, nv.ptr(mean),
                nv.ptr(invstd), nv.ptr(gamma), None, int(ctx.relu), nv.ptr(dx), C, nv.ptr(dgamma), nv.ptr(dbeta),
                nv.ptr(ctx.scratch), nv.stream())
        return dx, (dgamma if gamma is not None else None), (dbeta if gamma is not None else None), None, None, None, None, None


def batch_norm_act(x, gamma, beta, running_mean, running_var, eps, momentum, relu):
    return _BatchNormAct.apply(x, gamma, beta, running_mean, running_var, eps, momentum, relu)


class _SyncBatchNormAct(Function):
    """BatchNorm1d(train) whose batch statistics span all ranks of a process group (what makes graph-sharded
    data parallelism reproduce the single-device forward).  Forward: local (n, mean, M2) from esc_bn_stats, one
    all_gather of 2C+1 doubles, Chan merge, esc_bn_apply with the global statistics.  Backward: local column sums
    (esc_bn_bwd_sums), one all-reduce of 2C floats, esc_bn_bwd_apply with sums / N_global.  The incoming gradient must
    belong to ONE objective shared by the ranks (sum-form loss, `l1_loss(..., denom=1)` + `FlatBucket.all_reduce_sum`)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, relu, group):
        import torch.distributed as dist
        _dev(x, gamma, beta)
        _on(x.device, gamma, beta, running_mean, running_var)
        x, ldx = _rows(x)
        M, C = x.shape
        dev = x.device
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        invstd = torch.empty(C, dtype=torch.float32, device=dev)
        scratch = torch.empty(nv.lib().esc_bn_scratch(C), dtype=torch.float32, device=dev)
        s = nv.stream()
        pack = torch.zeros(2 * C + 1, dtype=torch.float64, device=dev)
        pack[0] = M
        if M > 1:
            nv.call("esc_bn_stats", nv.ptr(x), ldx, M, C, float(eps), float(momentum), nv.ptr(mean), nv.ptr(invstd),
                    None, None, None, None, None, None, nv.ptr(scratch), s)
            pack[1:C + 1] = mean.double()
            pack[C + 1:] = (1.0 / invstd.double().pow(2) - float(eps)).clamp_min(0.0) * M        # M2 = var * n
        elif M == 1:
            pack[1:C + 1] = x[0].double()
        world = dist.get_world_size(group)
        gathered = [torch.empty_like(pack) for _ in range(world)]
        dist.all_gather(gathered, pack, group=group)
        n, mu, m2 = gathered[0][0].clone(), gathered[0][1:C + 1].clone(), gathered[0][C + 1:].clone()
        for g in gathered[1:]:                                   # Chan merge in rank order (identical on every rank)
            nb, mub, m2b = g[0], g[1:C + 1], g[C + 1:]
            tot = n + nb
            delta = mub - mu
            mu = mu + delta * (nb / tot)
            m2 = m2 + m2b + delta * delta * (n * nb / tot)
            n = tot
        if float(n) <= 1:
            raise ValueError("Expected more than 1 value per channel when training (over all ranks)")
        var = m2 / n
        mean = mu.float()
        invstd = torch.rsqrt(var + float(eps)).float()
        if running_mean is not None:
            running_mean.mul_(1.0 - momentum).add_(mean, alpha=momentum)
            running_var.mul_(1.0 - momentum).add_((m2 / (n - 1)).float(), alpha=momentum)
        y = torch.empty((M, C), dtype=torch.float32, device=dev)
        if M > 0:
            nv.call("esc_bn_apply", nv.ptr(x), ldx, M, C, nv.ptr(mean), nv.ptr(invstd), nv.ptr(gamma), nv.ptr(beta),
                    int(relu), nv.ptr(y), C, s)
        ctx.save_for_backward(x, y if relu else None, gamma, mean, invstd)
        ctx.relu, ctx.scratch, ctx.group, ctx.n_global = int(relu), scratch, group, float(n)
        return y

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        x, y, gamma, mean, invstd = ctx.saved_tensors
        dy, ldg = _rows(dy)
        M, C = x.shape
        dev = x.device
        sums = torch.zeros(2 * C, dtype=torch.float32, device=dev)
        dgamma = torch.zeros(C, dtype=torch.float32, device=dev)
        dbeta = torch.zeros(C, dtype=torch.float32, device=dev)
        s = nv.stream()
        if M > 0:
            nv.call("esc_bn_bwd_sums", nv.ptr(x), x.stride(0), nv.ptr(y), C, nv.ptr(dy), ldg, M, C, nv.ptr(mean),
                    nv.ptr(invstd), nv.ptr(gamma), None, int(ctx.relu), nv.ptr(sums), nv.ptr(dgamma), nv.ptr(dbeta),
                    nv.ptr(ctx.scratch), s)
        dist.all_reduce(sums, group=ctx.group)
        coef = sums / ctx.n_global
        dx = torch.empty((M, C), dtype=torch.float32, device=dev)
        if M > 0:
            nv.call("esc_bn_bwd_apply", nv.ptr(x), x.stride(0), nv.ptr(y), C, nv.ptr(dy), ldg, M, C, nv.ptr(mean),
                    nv.ptr(invstd), nv.ptr(gamma), None, int(ctx.relu), nv.ptr(coef), nv.ptr(dx), C, s)
        return dx, dgamma, dbeta, None, None, None, None, None, None


def sync_batch_norm_act(x, gamma, beta, running_mean, running_var, eps, momentum, relu, group=None):
    return _SyncBatchNormAct.apply(x, gamma, beta, running_mean, running_var, eps, momentum, relu, group)


class _BnEvalAct(Function):
    """Inference-mode BatchNorm (+ fused activation) on the RUNNING statistics, differentiable: y = act(x*scale + shift)
    with scale = gamma / sqrt(running_var + eps), shift = beta - running_mean*scale (model.eval() with gradients enabled:
    fine-tuning with frozen statistics, input-gradient probes).  Backward on the BatchNorm kernels with the batch terms
    switched off: column sums give d(gamma), d(beta); dx = scale * dy * act'(.)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, relu):
        _dev(x)
        _on(x.device, gamma, beta, running_mean, running_var)
        x, ldx = _rows(x)
        M, C = x.shape
        dev = x.device
        scale, shift = torch.empty(C, dtype=torch.float32, device=dev), torch.empty(C, dtype=torch.float32, device=dev)
        s = nv.stream()
        nv.call("esc_bn_eval_coef", nv.ptr(running_mean), nv.ptr(running_var), nv.ptr(gamma), nv.ptr(beta), float(eps), C,
                nv.ptr(scale), nv.ptr(shift), s)
        y = torch.empty((M, C), dtype=torch.float32, device=dev)
        if M > 0:
            nv.call("esc_affine_act", nv.ptr(x), ldx, M, C, nv.ptr(scale), nv.ptr(shift), int(relu), nv.ptr(y), C, s)
        ctx.save_for_backward(x, y if relu else None, gamma, beta, running_mean, running_var)
        ctx.relu, ctx.eps = int(relu), float(eps)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, beta, rm, rv = ctx.saved_tensors
        dy, ldg = _rows(dy)
        M, C = x.shape
        dev = x.device
        s = nv.stream()
        invstd, unused = torch.empty(C, dtype=torch.float32, device=dev), torch.empty(C, dtype=torch.float32, device=dev)
        nv.call("esc_bn_eval_coef", nv.ptr(rm), nv.ptr(rv), None, None, ctx.eps, C, nv.ptr(invstd), nv.ptr(unused), s)
        dx = torch.empty((M, C), dtype=torch.float32, device=dev)
        dgamma, dbeta = torch.zeros(C, dtype=torch.float32, device=dev), torch.zeros(C, dtype=torch.float32, device=dev)
        if M > 0:
            scratch = torch.empty(nv.lib().esc_bn_scratch(C), dtype=torch.float32, device=dev)
            sums = torch.empty(2 * C, dtype=torch.float32, device=dev)
            nv.call("esc_bn_bwd_sums", nv.ptr(x), x.stride(0), nv.ptr(y), C, nv.ptr(dy), ldg, M, C, nv.ptr(rm), nv.ptr(invstd),
                    nv.ptr(gamma), nv.ptr(beta), ctx.relu, nv.ptr(sums), nv.ptr(dgamma), nv.ptr(dbeta), nv.ptr(scratch), s)
            zero = torch.zeros(2 * C, dtype=torch.float32, device=dev)       # no batch-statistics terms in eval mode
            nv.call("esc_bn_bwd_apply", nv.ptr(x), x.stride(0), nv.ptr(y), C, nv.ptr(dy), ldg, M, C, nv.ptr(rm), nv.ptr(invstd),
                    nv.ptr(gamma), nv.ptr(beta), ctx.relu, nv.ptr(zero), nv.ptr(dx), C, s)
        return dx, (dgamma if gamma is not None else None), (dbeta if beta is not None else None), None, None, None, None


def bn_eval_act(x, gamma, beta, running_mean, running_var, eps, relu):
    """Inference-mode BatchNorm (+activation) with running statistics; differentiable (see _BnEvalAct)."""
    return _BnEvalAct.apply(x, gamma, beta, running_mean, running_var, eps, relu)


class _L1Loss(Function):
    """torch.nn.L1Loss()(pred, y) (run_graphcount.py:500-501); `denom` overrides the mean's divisor
    (global node count under graph-sharded data parallelism)."""

    @staticmethod
    def forward(ctx, pred, y, denom):
        _dev(pred, y)
        _on(pred.device, y)
        pred = pred.contiguous().view(-1)
        y = y.contiguous().view(-1)
        if pred.numel() != y.numel():
            raise ValueError("l1_loss: %d predictions vs %d targets" % (pred.numel(), y.numel()))
        M = pred.numel()
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        dpred = torch.empty(M, dtype=torch.float32, device=pred.device)
        nv.call("esc_l1_loss", nv.ptr(pred), nv.ptr(y), M, int(denom or M), 1.0, nv.ptr(loss), nv.ptr(dpred), nv.stream())
        ctx.save_for_backward(dpred)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return (dpred * g).view(-1, 1), None, None


def l1_loss(pred, y, denom=None):
    return _L1Loss.apply(pred, y, denom)


class _BceLogits(Function):
    """BCEWithLogitsLoss()(pred[is_labeled], y[is_labeled]), is_labeled = (y == y) (run_ogb_mol.py:65-72);
    `denom` overrides the divisor (global labeled count under graph sharding)."""

    @staticmethod
    def forward(ctx, pred, y, denom):
        _dev(pred, y)
        _on(pred.device, y)
        shape = pred.shape
        pred = pred.contiguous().view(-1)
        y = y.contiguous().view(-1).float()
        if pred.numel() != y.numel():
            raise ValueError("bce_with_logits_loss: %d predictions vs %d targets" % (pred.numel(), y.numel()))
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        dpred = torch.empty(pred.numel(), dtype=torch.float32, device=pred.device)
        nv.call("esc_bce_logits_loss", nv.ptr(pred), nv.ptr(y), pred.numel(), int(denom or 0), nv.ptr(loss), nv.ptr(dpred),
                nv.stream())
        ctx.save_for_backward(dpred)
        ctx.shape = shape
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return (dpred * g).view(ctx.shape), None, None


def bce_with_logits_loss(pred, y, denom=None):
    return _BceLogits.apply(pred, y, denom)


class _SegmentPool(Function):
    """global_add_pool / global_mean_pool (run_graphcount.py:179, zinc_models.py:602) over a sorted batch vector."""

    @staticmethod
    def forward(ctx, x, seg_ptr, mean):
        _dev(x)
        _on(x.device, seg_ptr)
        x, ldx = _rows(x)
        G, C = seg_ptr.numel() - 1, x.size(1)
        out = torch.empty((G, C), dtype=torch.float32, device=x.device)
        nv.call("esc_segment_pool_fwd", nv.ptr(x), ldx, nv.ptr(seg_ptr), G, C, int(mean), nv.ptr(out), C, nv.stream())
        ctx.seg_ptr, ctx.mean, ctx.n = seg_ptr, bool(mean), x.size(0)
        return out

    @staticmethod
    def backward(ctx, g):
        g, ldg = _rows(g)
        G, C = g.shape
        dx = torch.empty((ctx.n, C), dtype=torch.float32, device=g.device)
        nv.call("esc_segment_pool_bwd", nv.ptr(g), ldg, nv.ptr(ctx.seg_ptr), G, C, int(ctx.mean), nv.ptr(dx), C,
                nv.stream())
        return dx, None, None


def _seg_ptr(batch, size=None):
    """int32 [G+1] segment pointers of a non-decreasing batch vector (esc_plan_csr: integer histogram + scan), built
    once per batch tensor and cached on it (the device collate pre-fills the cache: it knows the node ranges)"""
    from .plan import _csr
    cache = getattr(batch, "_esc_seg", None)
    key = (batch._version, size)
    if cache is not None and key in cache:
        return cache[key]
    n_seg = int(batch[-1].item()) + 1 if (size is None and batch.numel()) else int(size or 0)
    if batch.numel() > 1 and not bool((batch[1:] >= batch[:-1]).all()):
        raise ValueError("segment_pool: batch vector must be sorted")
    seg_ptr = _csr(batch, max(n_seg, 1), want_perm=False)[0][:n_seg + 1]
    batch._esc_seg = {key: seg_ptr}
    return seg_ptr


def segment_pool(x, batch, size=None, mean=False):
    """`batch` must be non-decreasing (Batch.from_data_list / the device collate produce it that way)."""
    _dev(x)
    _on(x.device, batch)
    return _SegmentPool.apply(x, _seg_ptr(batch, size), mean)


class _Embedding(Function):
    """weight[index] via esc_bag_fwd with unit counts (exact: 0 + w*1); gradient via the CSC segmented sum."""

    @staticmethod
    def forward(ctx, weight, index):
        _dev(weight)
        _on(weight.device, index)
        weight = weight.contiguous()
        idx = index.reshape(-1)
        n, H = idx.numel(), weight.size(1)
        if n and (int(idx.min()) < 0 or int(idx.max()) >= weight.size(0)):
            raise IndexError("embedding index out of range")
        dev = weight.device
        row_ptr = torch.arange(n + 1, dtype=torch.int32, device=dev)
        idx32 = idx.to(torch.int32)
        ones = torch.ones(n, dtype=torch.int32, device=dev)
        out = torch.empty((n, H), dtype=torch.float32, device=dev)
        nv.call("esc_bag_fwd", nv.ptr(weight), H, nv.ptr(row_ptr), nv.ptr(idx32), nv.ptr(ones), n, nv.ptr(out), H, nv.stream())
        ctx.save_for_backward(idx)
        ctx.rows, ctx.H = weight.size(0), H
        return out.view(*index.shape, H)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g, ld = _rows(g.reshape(-1, ctx.H))
        n, dev = idx.numel(), g.device
        from .plan import _csr
        col_ptr, c_row = _csr(idx, ctx.rows)                 # stable grouping of the positions by table row (csrc/plan.hip)
        c_val = torch.ones(n, dtype=torch.int32, device=dev)
        c_col = idx[c_row.long()].to(torch.int32)
        dw = torch.empty((ctx.rows, ctx.H), dtype=torch.float32, device=dev)
        scratch = torch.empty(max(1, nv.lib().esc_bag_bwd_scratch(n, ctx.H)), dtype=torch.float32, device=dev)
        nv.call("esc_bag_bwd_table", nv.ptr(g), ld, ctx.H, nv.ptr(col_ptr), nv.ptr(c_row), nv.ptr(c_val), nv.ptr(c_col),
                n, ctx.rows, nv.ptr(dw), nv.ptr(scratch), nv.stream())
        return dw, None


def embedding(weight, index):
    return _Embedding.apply(weight, index)


def embed_plan(index, dims, known_range=None):
    """bag plan of a sum-of-embeddings lookup over the concatenated tables: CSR by output row (idx32, row_ptr, ones) and
    CSC by table row (col_ptr, c_row, c_col); depends only on the index tensor — built once (esc_plan_csr) and cached
    on it (the bond features of a batch are looked up by every layer).
    known_range = (per-column minima, per-column maxima) of the DATASET the rows were gathered from (the device store
    computes them once): when they lie inside `dims` the per-batch range check — a device read-back that drains the
    stream — is skipped."""
    cache = getattr(index, "_esc_embed", None)
    if cache is not None and cache[0] == (dims, index._version):
        return cache[1]
    if index.device.type != "cuda":
        raise RuntimeError("esc_gnn_amd: embedding plans are built on the GPU (got a %s tensor)" % index.device.type)
    n, k = index.shape
    dev = index.device
    if k != len(dims) or k > 16:
        raise ValueError("embed_plan: index has %d columns for %d tables (at most 16)" % (k, len(dims)))
    if known_range is None:                              # the device store leaves the dataset-wide range on the tensors it collates
        tag = getattr(index, "_esc_known_range", None)
        if tag is not None and tag[1] == index._version:
            known_range = tag[0]
    trusted = (known_range is not None and len(known_range[0]) == k == len(dims) and
               all(lo >= 0 and hi < d for lo, hi, d in zip(known_range[0], known_range[1], dims)))
    idx = index if (index.dtype == torch.int64 and index.is_contiguous()) else index.to(torch.int64).contiguous()
    rows, total = int(sum(dims)), n * k
    # ONE int32 slab: idx32 | ones | c_row | c_col | row_ptr | col_ptr | flag | scratch (8-byte aligned start)
    sizes = [total, total, total, total, n + 1, rows + 1, 2]
    pad = (-sum(sizes)) % 2
    need = int(nv.lib().esc_embed_plan_scratch(n, k, rows))
    slab = torch.empty(sum(sizes) + pad + need, dtype=torch.int32, device=dev)
    parts, o = [], 0
    for sz in sizes:
        parts.append(slab[o:o + sz])
        o += sz
    idx32, ones, c_row, c_col, row_ptr, col_ptr, flag = parts
    scratch = slab[o + pad:]
    dims_h = (ctypes.c_int64 * k)(*[int(d) for d in dims])
    if total == 0:                                       # nothing to look up: empty entry arrays, all-zero pointers
        row_ptr.zero_()
        col_ptr.zero_()
        flag.zero_()
    else:
        nv.call("esc_embed_plan", nv.ptr(idx), n, k, ctypes.addressof(dims_h), nv.ptr(idx32), nv.ptr(ones), nv.ptr(row_ptr),
                nv.ptr(col_ptr), nv.ptr(c_row), nv.ptr(c_col), nv.ptr(scratch), nv.ptr(flag), nv.stream())
    if not trusted and total and int(flag[0].item()):    # one check per index tensor, not per lookup (ids known in range: no read-back)
        raise IndexError("embedding index out of range")
    plan = dict(idx32=idx32, row_ptr=row_ptr, ones=ones, col_ptr=col_ptr, c_row=c_row, c_col=c_col, entries=total, rows=rows,
                _slab=slab)
    index._esc_embed = ((dims, index._version), plan)
    return plan


class _EmbeddingSum(Function):
    """out[r] = sum_j table[index[r, j] + offset_j]: the sum-of-embeddings encoders of the OGB models (AtomEncoder,
    BondEncoder: ogb_mol_gnn.py:264-282 and ogb's BondEncoder) as ONE bag launch over the concatenated tables instead
    of one lookup per feature column.  Entries of a row are added in column order starting from 0, i.e. exactly the
    reference's `out = 0; for i: out = out + emb_i(x[:, i])`.  The CSC plan of the gradient depends only on the index
    tensor: it is built once and cached on it (the bond features of a batch are looked up by every layer)."""

    @staticmethod
    def forward(ctx, table, index, dims):
        _dev(table)
        _on(table.device, index)
        table = table.contiguous()
        n, k = index.shape
        H, dev = table.size(1), table.device
        plan = embed_plan(index, dims)
        out = torch.empty((n, H), dtype=torch.float32, device=dev)
        nv.call("esc_bag_fwd_rows", nv.ptr(table), table.size(0), H, nv.ptr(plan["row_ptr"]), nv.ptr(plan["idx32"]), nv.ptr(plan["ones"]), n,
                nv.ptr(out), H, 0, None, nv.stream())
        ctx.plan, ctx.rows, ctx.H, ctx.nk = plan, table.size(0), H, n * k
        return out

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan
        g, ld = _rows(g.reshape(-1, ctx.H))
        dw = torch.empty((ctx.rows, ctx.H), dtype=torch.float32, device=g.device)
        scratch = torch.empty(max(1, nv.lib().esc_bag_bwd_scratch(ctx.nk, ctx.H)), dtype=torch.float32, device=g.device)
        nv.call("esc_bag_bwd_table", nv.ptr(g), ld, ctx.H, nv.ptr(plan["col_ptr"]), nv.ptr(plan["c_row"]),
                nv.ptr(plan["ones"]), nv.ptr(plan["c_col"]), ctx.nk, ctx.rows, nv.ptr(dw), nv.ptr(scratch), nv.stream())
        return dw, None, None


def embedding_sum(weights, index):
    """sum_j weights[j][index[:, j]] (index: LongTensor [n, len(weights)])."""
    if index.dim() != 2 or index.size(1) != len(weights):
        raise ValueError("embedding_sum: index must be [n, %d]" % len(weights))
    dims = tuple(int(w.size(0)) for w in weights)
    return _EmbeddingSum.apply(torch.cat(list(weights), dim=0), index, dims)


class _NeighbourSum(Function):
    """out[i] = sum_{k: dst_k=i} relu(x[src_k] (+ e_k)) — GINE aggregate without the self term and with an
    optional edge term: the per-distance message sum of GINEPLUS / NAIVEGINEPLUS
    (modules/gine_operations.py:306-362)."""

    @staticmethod
    def forward(ctx, x, e, plan):
        _dev(x, e)
        _on(x.device, e)
        _plan_on(x.device, plan, "in_ptr", "in_edge", "in_src", "out_ptr", "out_edge", "out_dst")
        x, ldx = _rows(x)
        N, C = x.shape
        lde = 0
        if e is not None:
            e, lde = _rows(e)
            if e.shape != (plan.num_edges, C):
                raise ValueError("neighbour_sum: edge term %s does not match %d edges x %d" % (tuple(e.shape), plan.num_edges, C))
        out = torch.empty((N, C), dtype=torch.float32, device=x.device)
        nv.call("esc_gine_aggregate_fwd", nv.ptr(x), ldx, nv.ptr(e), lde, nv.ptr(plan.in_ptr), nv.ptr(plan.in_edge),
                nv.ptr(plan.in_src), None, N, C, nv.ptr(out), C, nv.stream())
        ctx.save_for_backward(x, e)
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, g):
        x, e = ctx.saved_tensors
        plan = ctx.plan
        g, ldg = _rows(g)
        N, C = x.shape
        d_e = torch.empty_like(e) if e is not None else None
        dx = torch.empty((N, C), dtype=torch.float32, device=x.device)
        nv.call("esc_gine_aggregate_bwd", nv.ptr(x), x.stride(0), nv.ptr(e), e.stride(0) if e is not None else 0,
                nv.ptr(g), ldg, nv.ptr(plan.out_ptr), nv.ptr(plan.out_edge), nv.ptr(plan.out_dst), None, N, C,
                nv.ptr(d_e), C if d_e is not None else 0, nv.ptr(dx), C, 0, None, nv.stream())
        return dx, d_e, None


def neighbour_sum(x, e, plan):
    return _NeighbourSum.apply(x, e, plan)


class _SegmentBroadcast(Function):
    """rows[batch] for a sorted batch vector (virtual-node embedding -> nodes, ogb_mol_gnn.py:744): the transpose of
    global_add_pool, run by the segment-pool kernels."""

    @staticmethod
    def forward(ctx, rows, seg_ptr, n):
        _dev(rows)
        _on(rows.device, seg_ptr)
        rows, ld = _rows(rows)
        G, C = rows.shape
        out = torch.empty((n, C), dtype=torch.float32, device=rows.device)
        nv.call("esc_segment_pool_bwd", nv.ptr(rows), ld, nv.ptr(seg_ptr), G, C, 0, nv.ptr(out), C, nv.stream())
        ctx.seg_ptr, ctx.G = seg_ptr, G
        return out

    @staticmethod
    def backward(ctx, g):
        g, ld = _rows(g)
        C = g.size(1)
        d = torch.empty((ctx.G, C), dtype=torch.float32, device=g.device)
        nv.call("esc_segment_pool_fwd", nv.ptr(g), ld, nv.ptr(ctx.seg_ptr), ctx.G, C, 0, nv.ptr(d), C, nv.stream())
        return d, None, None


def segment_broadcast(rows, batch, size=None):
    size = rows.size(0) if size is None else int(size)
    return _SegmentBroadcast.apply(rows, _seg_ptr(batch, size), batch.numel())

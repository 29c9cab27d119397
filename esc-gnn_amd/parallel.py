"""Graph-sharded data parallelism for NestedGIN_eff.

The reference has no distributed path (its only hook, kernel/train_eval.py:44-58, is never
initialised).  Graphs are independent, so a global batch is sharded BY GRAPH: rank r of W takes the
contiguous slice [r*B/W, (r+1)*B/W) (concatenating rank outputs reproduces single-device order), runs
the hot path on its shard with no data-path collective, and the step ends with ONE all-reduce of the
flat fp32 gradient bucket (RCCL over xGMI on MI355X; gloo in the CPU tests).

Exact global-mean loss: the reference's L1Loss averages over all N nodes of the batch.  Each rank
back-propagates its LOCAL mean; the bucket carries `grad * n_local` plus one extra slot holding
`n_local`, so after a single SUM all-reduce `bucket[:-1] / bucket[-1]` is the gradient of
sum_r sum_i |err| / N_global — no second collective for the denominator.

BatchNorm uses each shard's own batch statistics (plain DP).  Matching the single-device statistics
would need a (sum, sumsq, count) all-reduce per BN layer (SyncBN) — not enabled in this revision.
"""
import weakref

import torch
import torch.distributed as dist

_BUCKETS = weakref.WeakValueDictionary()      # gradient-store address -> FlatBucket (engine.py: direct gradient writes)


def shard_slice(batch_size, rank, world):
    """Contiguous graph slice of rank `rank` (sizes differ by at most one)."""
    base, rem = divmod(batch_size, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_slice_balanced(weights, rank, world):
    """Contiguous graph slice of rank `rank` that balances the SUM of `weights` (per-graph edge counts: the step's cost is
    edge-sized work, SURVEY 8e "optionally balance by sum E") instead of the graph count: the cut points are where the
    running sum crosses k/world of the total, every rank keeps at least one graph while there are enough of them.
    Deterministic and identical on every rank (host arithmetic on the same numbers)."""
    w = [float(x) for x in weights]
    n = len(w)
    if n <= world:
        return shard_slice(n, rank, world)
    total, run, cuts, k = sum(w), 0.0, [0], 1
    for i, x in enumerate(w):
        # close slice k-1 in front of graph i when adding it would overshoot the target by more than stopping short undershoots
        while k < world and len(cuts) == k and i > cuts[-1] and (n - i) >= (world - k):
            target = total * k / world
            if run + x - target > target - run or (n - i) == (world - k):
                cuts.append(i)
                k += 1
            else:
                break
        run += x
    while len(cuts) < world:                             # (degenerate weights: fall back to the remaining graphs one by one)
        cuts.append(min(n - (world - len(cuts)), max(cuts[-1] + 1, n - (world - len(cuts)))))
    cuts.append(n)
    return cuts[rank], cuts[rank + 1]


class FlatBucket(object):
    """Re-homes parameters and gradients of a model as views into two flat fp32 buffers
    (+1 trailing slot in the gradient buffer for the node-count piggyback)."""

    ALIGN = 16      # floats: every parameter (and its gradient) starts on a 64-byte boundary, so the kernels'
                    # float4 paths apply to all of them (1-element tensors such as GINEConv.eps would otherwise
                    # push everything behind them off 16-byte alignment)

    def __init__(self, params, late=None):
        """late: parameters whose gradients are complete only at the very end of a step — for NestedGIN_eff the EDGE
        pipeline's (z_initial, z_embedding.*, conv*.lin.*: their last contribution is the backward tail of the edge
        stream, ~150 us after the node pipeline's gradients are final).  They are laid out BEHIND all other parameters,
        so that the two groups are two contiguous buckets: `all_reduce_early` can run while the tail is still computing
        (its collective overlaps it), `all_reduce_late` carries the rest plus the node-count slot."""
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("FlatBucket: no parameters")
        late_ids = {id(p) for p in (late or [])}
        if late_ids:
            self.params = [p for p in self.params if id(p) not in late_ids] + [p for p in self.params if id(p) in late_ids]
        dev = self.params[0].device
        self.offsets, n = [], 0
        self.early_numel = None
        for p in self.params:
            if self.early_numel is None and id(p) in late_ids:
                self.early_numel = n
            self.offsets.append(n)
            n += -(-p.numel() // self.ALIGN) * self.ALIGN
        if self.early_numel is None:
            self.early_numel = n
        self.numel = n
        self.flat_param = torch.zeros(n, dtype=torch.float32, device=dev)     # padding stays 0 (zero grad => no update)
        self._grad_store = torch.zeros(n + 1, dtype=torch.float32, device=dev)
        self.flat_grad = self._grad_store[:n]
        for p, off in zip(self.params, self.offsets):
            k = p.numel()
            self.flat_param[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + k].view(p.shape)
            p.grad = self.flat_grad[off:off + k].view(p.shape)
        self._offset_of = {id(p): off for p, off in zip(self.params, self.offsets)}
        self.engine_direct = True          # the whole-step engines may write a backward's gradients straight into the bucket
        self._clean_version = None         # version of the gradient store right after zero_grad(): nobody has written since
        _BUCKETS[self._grad_store.data_ptr()] = self

    def zero_grad(self, set_to_none=False):
        self._grad_store.zero_()
        base = self.flat_grad.data_ptr()
        for p, off in zip(self.params, self.offsets):      # autograd may have replaced .grad: re-bind the views
            g = p.grad
            if g is None or g.data_ptr() != base + 4 * off:
                p.grad = self.flat_grad[off:off + p.numel()].view(p.shape)
        self._clean_version = self._grad_store._version

    def direct_grad_addresses(self, params):
        """Addresses of the gradient slots of `params` inside this bucket IF a backward may write them directly: the bucket
        is still exactly as zero_grad() left it (no accumulation has touched it — torch's version counter of the store
        is unchanged — and no engine has written into it), and every parameter's .grad is this bucket's view.  Writing
        into zeros equals accumulating into them, later AccumulateGrad nodes add on top as usual.  None otherwise."""
        if not self.engine_direct or self._clean_version is None or self._clean_version != self._grad_store._version:
            return None
        base, out = self.flat_grad.data_ptr(), []
        for p in params:
            off = self._offset_of.get(id(p))
            if off is None or p.grad is None or p.grad.data_ptr() != base + 4 * off or not p.requires_grad:
                return None
            out.append(base + 4 * off)
        self._clean_version = None                         # (the raw write does not move torch's version counter)
        return out

    def grad_offsets(self, params):
        """float offsets of the gradient slots of `params` inside flat_grad if every .grad still is this bucket's view
        (the engine node's backward then accumulates a second backward's gradients with ONE add over a scratch copy of the
        same layout); None otherwise"""
        base, out = self.flat_grad.data_ptr(), []
        for p in params:
            off = self._offset_of.get(id(p))
            if off is None or p.grad is None or p.grad.data_ptr() != base + 4 * off:
                return None
            out.append(off)
        return out

    def all_reduce_weighted(self, n_local, group=None):
        """grad <- sum_r n_r * grad_r / sum_r n_r   with one SUM all-reduce."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return float(n_local)
        self.flat_grad.mul_(float(n_local))
        self._grad_store[-1] = float(n_local)
        dist.all_reduce(self._grad_store, op=dist.ReduceOp.SUM, group=group)
        total = self._grad_store[-1:].clone()
        self.flat_grad.div_(total)
        return total

    def all_reduce_sum(self, n_local, group=None):
        """For gradients that are already SUMS over the local targets (StepEngine.train_step(loss_denom=1)): one SUM
        all-reduce of `grad ++ [n_local]`; returns the device scalar holding the global count, to be handed to
        FlatAdam.step(grad_denom=...) — no scaling pass over the bucket before or after the collective."""
        self._grad_store[-1:].fill_(float(n_local))
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self._grad_store, op=dist.ReduceOp.SUM, group=group)
        return self._grad_store[-1:]


    # ---- the same exchange in two buckets (sum-form gradients, StepEngine.begin_step / end_step) ------------------
    def all_reduce_early(self, group=None):
        """SUM all-reduce of the bucket of parameters that are NOT `late`: call between StepEngine.begin_step and
        end_step — on the node stream the collective sits behind the node pipeline's gradient reductions and runs while
        the edge pipeline finishes its backward tail."""
        if self.early_numel and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self._grad_store[:self.early_numel], op=dist.ReduceOp.SUM, group=group)

    def all_reduce_late(self, n_local, group=None):
        """... and of the `late` bucket ++ [n_local] after end_step; returns the device scalar with the global count
        (FlatAdam.step(grad_denom=...)).  early + late == all_reduce_sum, element for element."""
        self._grad_store[-1:].fill_(float(n_local))
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self._grad_store[self.early_numel:], op=dist.ReduceOp.SUM, group=group)
        return self._grad_store[-1:]


def edge_pipeline_parameters(model):
    """The `late` set of a NestedGIN_eff-shaped model: parameters whose gradients come off the edge pipeline
    (run_graphcount.py:54-61 z_embedding, :51 z_initial, GINEConv.lin)."""
    out = []
    for name, p in model.named_parameters():
        head = name.split(".")[0]
        if head in ("z_initial", "z_embedding") or ".lin." in ("." + name) and (head == "conv1" or head == "convs"):
            out.append(p)
    return out


def broadcast_buffers(model, src=0, group=None):
    """BatchNorm running statistics are updated from rank-local shards: before an evaluation or a checkpoint every rank
    takes rank `src`'s (what DistributedDataParallel(broadcast_buffers=True) does), so that the logged validation number
    is reproducible from the saved model."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in model.buffers():
        dist.broadcast(t.data, src, group=group)


def broadcast_parameters(model, src=0, group=None):
    """Identical replicas at start (parameters and BN buffers)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src, group=group)

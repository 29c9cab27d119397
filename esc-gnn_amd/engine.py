"""StepEngine — NestedGIN_eff training/eval step as ONE call into libescgnn_hip.so
(esc_engine_train_step / esc_engine_predict, csrc/engine.hip).

The autograd path (`model(batch)`; `loss.backward()`) stays the drop-in interface; this is the fast
path for the same module: it reads the module's parameters, writes the same `.grad` slots
(views of the optimiser's flat gradient bucket) and updates the same BatchNorm buffers, so
checkpoints, `optimizer.step()` and the data-parallel all-reduce are unchanged.
"""
import ctypes
from ctypes import c_float, c_int32, c_int64, c_uint64, c_void_p

import torch

from . import _native as nv
from .plan import plan_of

MAX_LAYERS = 16


class _Linear(ctypes.Structure):
    _fields_ = [("w", c_void_p), ("b", c_void_p), ("dw", c_void_p), ("db", c_void_p),
                ("in_dim", c_int64), ("out_dim", c_int64)]


class _BN(ctypes.Structure):
    _fields_ = [("gamma", c_void_p), ("beta", c_void_p), ("dgamma", c_void_p), ("dbeta", c_void_p),
                ("running_mean", c_void_p), ("running_var", c_void_p), ("eps", c_float), ("momentum", c_float)]


class _MLP(ctypes.Structure):
    _fields_ = [("lin0", _Linear), ("bn0", _BN), ("lin1", _Linear), ("bn1", _BN)]


class _Conv(ctypes.Structure):
    _fields_ = [("eps", c_void_p), ("deps", c_void_p), ("nn", _MLP), ("lin", _Linear)]


class _Model(ctypes.Structure):
    _fields_ = [("num_layers", c_int64), ("hidden", c_int64), ("in_dim", c_int64), ("z_rows", c_int64),
                ("z_table", c_void_p), ("dz_table", c_void_p),
                ("zbn0", _BN), ("zlin", _Linear), ("zbn1", _BN), ("xemb", _MLP),
                ("conv", _Conv * MAX_LAYERS), ("lin1", _Linear), ("bn_lin1", _BN), ("lin2", _Linear)]


class _Batch(ctypes.Structure):
    _fields_ = ([("N", c_int64), ("E", c_int64), ("Z", c_int64), ("x", c_void_p), ("y", c_void_p)] +
                [(n, c_void_p) for n in ("in_ptr", "in_edge", "in_src", "out_ptr", "out_edge", "out_dst",
                                         "row_ptr", "bag_idx", "bag_val", "col_ptr", "col_row", "col_val", "col_col")])


class _Embed(ctypes.Structure):
    _fields_ = [("w", c_void_p), ("dw", c_void_p), ("rows", c_int64), ("dim", c_int64)]


class _ZincModel(ctypes.Structure):
    _fields_ = [("num_layers", c_int64), ("hidden", c_int64), ("z_rows", c_int64),
                ("z_table", c_void_p), ("dz_table", c_void_p),
                ("zbn0", _BN), ("zlin", _Linear), ("zbn1", _BN), ("node_emb", _Embed), ("edge_emb", _Embed),
                ("conv", _Conv * MAX_LAYERS), ("lin1", _Linear), ("bn_lin1", _BN), ("lin2", _Linear)]


class _MolBatch(ctypes.Structure):
    _fields_ = ([("N", c_int64), ("E", c_int64), ("Z", c_int64), ("G", c_int64), ("node_type", c_void_p),
                 ("edge_type", c_void_p), ("y", c_void_p), ("graph_ptr", c_void_p)] +
                [(n, c_void_p) for n in ("in_ptr", "in_edge", "in_src", "out_ptr", "out_edge", "out_dst",
                                         "row_ptr", "bag_idx", "bag_val", "col_ptr", "col_row", "col_val", "col_col")])


def _grad_ptr(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p.data)
    return p.grad.data_ptr()


def _lin(mod, gp=_grad_ptr):
    s = _Linear()
    s.w, s.dw = mod.weight.data_ptr(), gp(mod.weight)
    s.b, s.db = (mod.bias.data_ptr(), gp(mod.bias)) if mod.bias is not None else (None, None)
    s.in_dim, s.out_dim = mod.in_features, mod.out_features
    return s


def _bn(mod, gp=_grad_ptr):
    s = _BN()
    s.gamma, s.beta = mod.weight.data_ptr(), mod.bias.data_ptr()
    s.dgamma, s.dbeta = gp(mod.weight), gp(mod.bias)
    s.running_mean, s.running_var = mod.running_mean.data_ptr(), mod.running_var.data_ptr()
    s.eps, s.momentum = mod.eps, mod.momentum
    return s


def _mlp(seq, gp=_grad_ptr):            # Sequential(Linear, Dropout, BN, ReLU, Linear, Dropout, BN, ReLU)
    s = _MLP()
    s.lin0, s.bn0, s.lin1, s.bn1 = _lin(seq[0], gp), _bn(seq[2], gp), _lin(seq[4], gp), _bn(seq[6], gp)
    return s


def describe(m, gp=_grad_ptr):
    """esc_nested_gin_t of a NestedGIN_eff module; `gp(parameter)` names the address its gradient is written to."""
    d = _Model()
    convs = [m.conv1] + list(m.convs)
    if len(convs) > MAX_LAYERS:
        raise ValueError("at most %d layers" % MAX_LAYERS)
    d.num_layers, d.hidden = len(convs), m.lin2.in_features
    d.in_dim, d.z_rows = m.x_embedding[0].in_features, m.z_initial.num_embeddings
    d.z_table, d.dz_table = m.z_initial.weight.data_ptr(), gp(m.z_initial.weight)
    d.zbn0, d.zlin, d.zbn1 = _bn(m.z_embedding[1], gp), _lin(m.z_embedding[3], gp), _bn(m.z_embedding[5], gp)
    d.xemb = _mlp(m.x_embedding, gp)
    for i, cv in enumerate(convs):
        c = _Conv()
        c.eps, c.deps = cv.eps.data_ptr(), gp(cv.eps)
        c.nn, c.lin = _mlp(cv.nn, gp), _lin(cv.lin, gp)
        d.conv[i] = c
    d.lin1, d.bn_lin1, d.lin2 = _lin(m.lin1, gp), _bn(m.bn_lin1, gp), _lin(m.lin2, gp)
    return d


def _bns(m, cache=None):
    """the model's esc BatchNorm modules.  `cache`: a node cache the caller has just validated (engine_forward: a replaced
    module brings new buffers, which invalidates it) — the per-step callers then skip the walk over ~200 submodules"""
    if cache is None:
        cache = m.__dict__.get("_esc_node_cache")
        if cache is not None and not cache.valid():
            cache = None
    if cache is not None:
        return cache.bns
    return [mod for mod in m.modules() if hasattr(mod, "sync_group")]


def _sync_groups(m, cache=None):
    return [mod.sync_group for mod in _bns(m, cache) if mod.sync_group is not False]


def engine_supports(m, cache=None):
    """The configuration the whole-step engine covers: the run_graphcount one (reference :465).  BatchNorm statistics over
    several ranks (nn.BatchNorm1d.convert_sync) are served too: the engine exchanges them through a collective provider
    (install_collective below) — all BatchNorms of the model must then use the same process group."""
    if m.graph_pred or m.dropout != 0 or not m.use_cycle or m.lin1.weight.device.type != "cuda":
        return False
    if m.lin2.out_features != 1 or m.lin2.in_features % 4 != 0:
        return False
    return _one_sync_group(m, cache)


def _one_sync_group(m, cache=None):
    """no SyncBN at all, or EVERY BatchNorm of the model on the same process group (the engines exchange all statistics
    through one collective provider)"""
    bns = _bns(m, cache)
    groups = [mod.sync_group for mod in bns if mod.sync_group is not False]
    return not groups or (len(groups) == len(bns) and all(g is groups[0] for g in groups))


def _bn_width(m, cache=None):
    """widest BatchNorm of the model (the exchange buffers hold world * 3 * width floats)"""
    mods = _bns(m, cache)
    return max([mod.num_features for mod in mods if isinstance(mod, torch.nn.BatchNorm1d)] or [1])


def _arm_collective(model, device, cache=None):
    """SyncBN models: (re)install the engines' all-reduce for the model's group; others: nothing to do"""
    groups = _sync_groups(model, cache)
    if not groups:
        return
    if not _one_sync_group(model, cache):
        raise NotImplementedError("step engine: all BatchNorm layers must share one sync group")
    width = _bn_width(model, cache)
    if _collective.get("group", False) is not groups[0] or _collective.get("width", 0) < width:
        install_collective(width, device, groups[0])


_ALLREDUCE_T = ctypes.CFUNCTYPE(ctypes.c_int, c_void_p, c_int64, c_void_p, c_void_p)
_collective = {}            # the installed provider: keeps the callback object, the exchange buffers and the group alive


def install_collective(hidden, device, group=None):
    """Give the step engine its all-reduce (SyncBN under graph-sharded data parallelism): RCCL through
    torch.distributed (backend "nccl" on MI355X; gloo in the two-rank tests).  The engine calls back with one of the two
    exchange buffers allocated here and the HIP stream the exchange must be ordered on."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) <= 1:
        nv.call("esc_engine_set_collective", None, None, 0, 1, None, None, 0)
        _collective.clear()
        return False
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    cap = world * 3 * hidden
    bufs = [torch.zeros(cap, dtype=torch.float32, device=device) for _ in range(2)]
    by_ptr = {b.data_ptr(): b for b in bufs}

    def allreduce(buf, n, stream, user):
        try:
            t = by_ptr[buf][:n]
            cur = torch.cuda.current_stream(device)
            if stream is None or stream == cur.cuda_stream:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            else:                                         # the edge pipeline's own stream
                with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=device)):
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return 0
        except Exception as exc:                          # never let an exception cross the C boundary
            print("esc_gnn_amd: collective provider failed: %r" % (exc,))
            return 1

    cb = _ALLREDUCE_T(allreduce)
    nv.call("esc_engine_set_collective", ctypes.cast(cb, c_void_p), None, rank, world, bufs[0].data_ptr(), bufs[1].data_ptr(), cap)
    _collective.update(cb=cb, bufs=bufs, group=group, world=world, width=hidden)
    return True


class _AddressGuard(object):
    """The step engines hand the device RAW addresses of the parameters, their gradients and the BatchNorm buffers.
    An optimiser that re-homes them (FlatAdam moves every parameter and gradient into one bucket),
    zero_grad(set_to_none=True) or module.to() leaves those addresses dangling — a step would read freed weights and
    write its gradients over whatever the allocator put there.  Every step therefore compares a few sentinel addresses
    (first / last parameter and their gradients, first buffer: the realistic changes move all of them) with the ones
    the descriptor was built from and rebuilds the descriptor when they differ."""

    def _sentinels(self):
        ps = self._guard_params
        first, last = ps[0], ps[-1]
        g0, g1 = first.grad, last.grad
        bufs = self._guard_buffers
        return (first.data_ptr(), last.data_ptr(), g0.data_ptr() if g0 is not None else 0,
                g1.data_ptr() if g1 is not None else 0, bufs[0].data_ptr() if bufs else 0)

    def _guard_arm(self):
        self._guard_params = list(self.model.parameters())
        self._guard_buffers = [b for b in self.model.buffers() if b.is_floating_point()]
        self._guard_sig = self._sentinels()

    def _guard_check(self):
        if self._sentinels() != self._guard_sig:
            self.refresh()

    def _mark_bucket_written(self):
        """train_step OVERWRITES the .grad storage through raw addresses, which torch's version counter does not see: a
        FlatBucket that still believes it is as zero_grad() left it would let a later `model(batch).backward()` write
        straight into it (overwriting this step's gradients instead of accumulating onto them)."""
        from .parallel import _BUCKETS
        g0 = self._guard_params[0].grad
        base = getattr(g0, "_base", None) if g0 is not None else None
        bucket = _BUCKETS.get(base.data_ptr()) if base is not None else None
        if bucket is not None:
            bucket._clean_version = None


class StepEngine(_AddressGuard):
    def __init__(self, model):
        if model.graph_pred or model.dropout != 0 or not model.use_cycle:
            raise NotImplementedError("StepEngine covers the run_graphcount configuration "
                                      "(graph_pred=False, dropout=0, use_cycle=True); use model(batch) otherwise")
        if model.lin1.weight.device.type != "cuda":
            raise RuntimeError("StepEngine runs on the HIP device only; there is no CPU fallback")
        self.model = model
        self._ws = None
        _arm_collective(model, model.lin1.weight.device)     # SyncBN: the engine exchanges the statistics itself
        self._bn_counters = [m.num_batches_tracked for m in model.modules()
                             if isinstance(m, torch.nn.BatchNorm1d) and m.num_batches_tracked is not None]
        self.refresh()

    def refresh(self):
        """(Re)read parameter / gradient / buffer addresses — call after the optimiser re-homed them."""
        m = self.model
        d = describe(m)
        self._desc = d
        self._keep = [p for p in m.parameters()]
        self._open = None
        self._guard_arm()

    def _batch(self, data, need_y):
        dev = (self.model if isinstance(self, StepEngine) else self).lin1.weight.device
        if data.x.device != dev:
            data.to(dev)
        plan = plan_of(data)
        if plan.in_ptr.device != dev:
            raise RuntimeError("StepEngine: batch plan lives on %s, model on %s" % (plan.in_ptr.device, dev))
        b = _Batch()
        b.N, b.E, b.Z = plan.num_nodes, plan.num_edges, plan.nnz
        x = data.x if data.x.is_contiguous() else data.x.contiguous()
        b.x = x.data_ptr()
        y = None
        if need_y:
            y = data.y.reshape(-1)
            y = y if (y.dtype == torch.float32 and y.is_contiguous()) else y.float().contiguous()
            if y.numel() != plan.num_nodes:
                raise ValueError("StepEngine: expected one target per node")
            b.y = y.data_ptr()
        for f in ("in_ptr", "in_edge", "in_src", "out_ptr", "out_edge", "out_dst", "row_ptr", "bag_idx", "bag_val",
                  "col_ptr", "col_row", "col_val", "col_col"):
            setattr(b, f, getattr(plan, f).data_ptr())
        return b, (x, y, plan)

    def _workspace(self, b):
        need = nv.lib().esc_engine_workspace_floats(ctypes.byref(self._desc), b.N, b.E, b.Z)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(int(need * 1.25), dtype=torch.float32, device=self.model.lin1.weight.device)
        return self._ws

    def train_step(self, data, loss_denom=None, return_pred=False, _entry="esc_engine_train_step"):
        """forward + L1 + backward; gradients land in the parameters' .grad (overwritten). Returns loss (0-d)."""
        dev = self.model.lin1.weight.device
        self._guard_check()
        b, keep = self._batch(data, True)
        ws = self._workspace(b)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        pred = torch.empty(b.N, dtype=torch.float32, device=dev) if return_pred else None
        nv.call(_entry, ctypes.byref(self._desc), ctypes.byref(b), ws.data_ptr(),
                int(loss_denom or 0), loss.data_ptr(), nv.ptr(pred), nv.stream())
        self._mark_bucket_written()
        self._open = (keep, ws) if _entry.endswith("_begin") else None     # operands stay alive until end_step
        if self._bn_counters:
            torch._foreach_add_(self._bn_counters, 1)
        return (loss.view(()), pred.view(-1, 1)) if return_pred else loss.view(())

    def begin_step(self, data, loss_denom=None, return_pred=False):
        """train_step up to (not including) the join with the edge stream: what the caller enqueues next on the current
        stream — typically `store.collate(next_ids)` — overlaps the tail of the edge pipeline.  Call end_step() before
        using the gradients / the loss."""
        return self.train_step(data, loss_denom, return_pred, _entry="esc_engine_train_step_begin")

    def end_step(self):
        nv.call("esc_engine_train_step_end")
        self._open = None

    @torch.no_grad()
    def predict(self, data):
        dev = self.model.lin1.weight.device
        self._guard_check()
        b, keep = self._batch(data, False)
        ws = self._workspace(b)
        pred = torch.empty(b.N, dtype=torch.float32, device=dev)
        nv.call("esc_engine_predict", ctypes.byref(self._desc), ctypes.byref(b), ws.data_ptr(), pred.data_ptr(),
                nv.stream())
        return pred.view(-1, 1)


class _NodeCache(object):
    """Per-model host-side cache of the autograd node: the parameter list, the BatchNorm step counters and a template of
    the model descriptor with the positions of its gradient pointers, so that a step costs two small numpy patches
    instead of two descriptor builds (~0.5 ms of Python)."""

    MARK = 0x5E5C00000000

    def __init__(self, model):
        import numpy as np
        self.params = list(model.parameters())
        self.buffers = [b for b in model.buffers() if b.is_floating_point()]   # the template bakes the running-stat addresses in
        self.key = tuple(t.data_ptr() for t in self.params + self.buffers)
        self.counters = [m.num_batches_tracked for m in model.modules()
                         if isinstance(m, torch.nn.BatchNorm1d) and m.num_batches_tracked is not None]
        self.bns = [m for m in model.modules() if hasattr(m, "sync_group")]     # see _bns
        with torch.enable_grad():       # the AccumulateGrad node of the first parameter: _node_backward asks the engine about it
            p0 = self.params[0]
            self.acc0 = p0.view_as(p0).grad_fn.next_functions[0][0] if p0.requires_grad else None
        index = {id(p): i for i, p in enumerate(self.params)}
        describe_fn = getattr(self, "_describe", describe)
        self.template = describe_fn(model, lambda p: self.MARK + index[id(p)])
        words = np.frombuffer(self.template, dtype=np.uint64)
        hits = np.nonzero((words >= self.MARK) & (words < self.MARK + len(self.params)))[0]
        self.slots = hits                                                  # word positions of the gradient pointers
        self.slot_param = (words[hits] - self.MARK).astype(np.int64)      # ... and whose gradient each one is
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += -(-p.numel() // 16) * 16                             # 64-byte aligned slices of one flat buffer
        self.offsets, self.total = offs, total
        self.byte_offsets = np.asarray(offs, dtype=np.uint64)[self.slot_param] * np.uint64(4)

    def valid(self):
        return self.key == tuple(t.data_ptr() for t in self.params + self.buffers)

    def descriptor(self, grad_base):
        import numpy as np
        d = getattr(self, "_struct", _Model).from_buffer_copy(self.template)
        np.frombuffer(d, dtype=np.uint64)[self.slots] = np.uint64(grad_base) + self.byte_offsets
        return d

    def descriptor_at(self, addresses):
        """descriptor whose gradient slot of parameter i is addresses[i] (a FlatAdam bucket's own layout)"""
        import numpy as np
        d = getattr(self, "_struct", _Model).from_buffer_copy(self.template)
        np.frombuffer(d, dtype=np.uint64)[self.slots] = np.asarray(addresses, dtype=np.uint64)[self.slot_param]
        return d

    def owning_bucket(self):
        """the FlatAdam / FlatBucket that owns the parameters' .grad storage and lets the engines write into it, or None.  None
        also when a parameter carries tensor hooks / post-accumulate hooks (DDP-style wrappers: they only fire on gradients
        that come back through autograd) or does not require a gradient."""
        from .parallel import _BUCKETS
        g0 = self.params[0].grad
        base = getattr(g0, "_base", None) if g0 is not None else None
        bucket = _BUCKETS.get(base.data_ptr()) if base is not None else None
        if bucket is None or not bucket.engine_direct:
            return None
        for p in self.params:
            if not p.requires_grad or p._backward_hooks or getattr(p, "_post_accumulate_grad_hooks", None):
                return None
        return bucket

    def direct_bucket(self):
        """owning_bucket() that is moreover still clean (see FlatBucket.direct_grad_addresses), as a list of gradient
        addresses — or None"""
        bucket = self.owning_bucket()
        return bucket.direct_grad_addresses(self.params) if bucket is not None else None

    def node_inputs(self):
        """The differentiable inputs of the engine's autograd node.  Normally every parameter (their gradients come back
        through autograd).  When a FlatAdam / FlatBucket owns every .grad, only the FIRST parameter: the backward writes
        (clean bucket) or adds (otherwise) the gradients into the bucket itself and returns none, so the graph carries one
        edge instead of one AccumulateGrad per parameter — 104 of them cost the reference's loop ~0.25 ms of host time per
        step (tools/measure/dropin_prof.py).  torch.autograd.grad through such a node reaches its one input only."""
        return (self.params[0],) if self.owning_bucket() is not None else tuple(self.params)


def _under_autograd_grad(cache):
    """inside a backward: is this torch.autograd.grad (functional: nothing may be accumulated) rather than .backward()?  The autograd
    engine refuses the question about a leaf's AccumulateGrad node exactly in that case."""
    if cache.acc0 is None:
        return False
    try:
        torch._C._will_engine_execute_node(cache.acc0)
        return False
    except RuntimeError as exc:
        return "autograd.grad" in str(exc)
    except Exception:
        return False


def _node_backward(ctx, dpred, entry):
    """backward of the three engine nodes: gradients straight into a clean FlatAdam bucket, else returned (or, for a
    node built on node_inputs()' short form, added into the bucket / the .grad tensors by hand)"""
    cache = ctx.cache
    if ctx.ws is None:
        raise RuntimeError("esc_gnn_amd: this engine node's workspace was released by its first backward; a second "
                           "backward through the same forward (retain_graph=True) needs the per-op path")
    g = dpred.reshape(-1)
    g = g if (g.dtype == torch.float32 and g.is_contiguous()) else g.float().contiguous()
    slim = ctx.n_in < len(cache.params)
    none = (None, None, None) + (None,) * ctx.n_in
    if _under_autograd_grad(cache):
        # torch.autograd.grad(...): a functional call — no .grad may change.  The gradients of the node's inputs are returned (all
        # parameters, or the one a short-form node was built on; asking for another parameter of such a node is autograd's own
        # "not used in the graph" error)
        flat = torch.empty(cache.total, dtype=torch.float32, device=dpred.device)
        desc = cache.descriptor(flat.data_ptr())
        nv.call(entry, ctypes.byref(desc), ctypes.byref(ctx.b), ctx.ws.data_ptr(), g.data_ptr(), nv.stream())
        ctx.ws = ctx.keep = None
        grads = tuple(flat[o:o + p.numel()].view(p.shape) if p.requires_grad else None
                      for p, o in zip(cache.params[:ctx.n_in], cache.offsets[:ctx.n_in]))
        return (None, None, None) + grads

    def run(desc):
        nv.call(entry, ctypes.byref(desc), ctypes.byref(ctx.b), ctx.ws.data_ptr(), g.data_ptr(), nv.stream())
        ctx.ws = ctx.keep = None

    bucket = cache.owning_bucket()
    direct = bucket.direct_grad_addresses(cache.params) if bucket is not None else None
    if direct is not None:             # every .grad is a clean FlatAdam bucket view: write there, nothing to accumulate
        run(cache.descriptor_at(direct))
        return none
    if slim:
        # the bucket has been written since its zero_grad() (a second backward before the optimiser step): accumulate
        offs = bucket.grad_offsets(cache.params) if bucket is not None else None
        if offs is not None:           # ... in one pass over a scratch copy with the bucket's own layout
            tmp = torch.zeros_like(bucket.flat_grad)
            run(cache.descriptor_at([tmp.data_ptr() + 4 * o for o in offs]))
            bucket.flat_grad.add_(tmp)
            return none
        flat = torch.zeros(cache.total, dtype=torch.float32, device=dpred.device)    # .grad was re-bound by the caller
        run(cache.descriptor(flat.data_ptr()))
        for p, o in zip(cache.params, cache.offsets):
            gp = flat[o:o + p.numel()].view(p.shape)
            if p.grad is None:
                p.grad = gp
            else:
                p.grad.add_(gp)
        return none
    flat = torch.empty(cache.total, dtype=torch.float32, device=dpred.device)   # fresh: the views alias nothing older
    run(cache.descriptor(flat.data_ptr()))
    grads = tuple(flat[o:o + p.numel()].view(p.shape) if p.requires_grad else None
                  for p, o in zip(cache.params, cache.offsets))
    return (None, None, None) + grads


def _node_cache(model):
    c = model.__dict__.get("_esc_node_cache")
    if c is None or not c.valid():
        c = _NodeCache(model)
        model.__dict__["_esc_node_cache"] = c
    return c


class _EngineNode(torch.autograd.Function):
    """`model(batch)` of a training-mode NestedGIN_eff as ONE autograd node on the whole-step engine: forward =
    esc_engine_forward_train, backward = esc_engine_backward with d(loss)/d(pred) of whatever loss the caller built.
    The user's own loop (`loss = L1Loss()(model(data), y); loss.backward(); optimizer.step()`, reference
    run_graphcount.py:494-505) then runs at engine speed instead of one autograd node per op."""

    @staticmethod
    def forward(ctx, model, data, cache, *params):
        dev = model.lin1.weight.device
        b, keep = StepEngine._batch(model, data, False)            # (used unbound: only reads the module)
        desc = cache.descriptor(0)                                 # the forward writes no gradient
        need = nv.lib().esc_engine_workspace_floats(ctypes.byref(desc), b.N, b.E, b.Z)
        ws = torch.empty(int(need), dtype=torch.float32, device=dev)   # private: stays intact until the backward
        pred = torch.empty(b.N, dtype=torch.float32, device=dev)
        nv.call("esc_engine_forward_train", ctypes.byref(desc), ctypes.byref(b), ws.data_ptr(), pred.data_ptr(), nv.stream())
        if cache.counters:
            torch._foreach_add_(cache.counters, 1)
        ctx.cache, ctx.b, ctx.keep, ctx.ws, ctx.n_in = cache, b, keep, ws, len(params)
        return pred.view(-1, 1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dpred):
        return _node_backward(ctx, dpred, "esc_engine_backward")


@torch.no_grad()
def engine_predict(model, data):
    """eval-mode `model(batch)` of the counting model as one call (esc_engine_predict: running statistics, no gradient
    state) — the validation / test passes of a training run are most of its batches"""
    cache = _node_cache(model)
    b, keep = StepEngine._batch(model, data, False)
    desc = cache.descriptor(0)
    need = nv.lib().esc_engine_workspace_floats(ctypes.byref(desc), b.N, b.E, b.Z)
    ws = torch.empty(int(need), dtype=torch.float32, device=model.lin1.weight.device)
    pred = torch.empty(b.N, dtype=torch.float32, device=model.lin1.weight.device)
    nv.call("esc_engine_predict", ctypes.byref(desc), ctypes.byref(b), ws.data_ptr(), pred.data_ptr(), nv.stream())
    return pred.view(-1, 1)


def engine_forward(model, data, cache=None):
    cache = cache if cache is not None else _node_cache(model)
    _arm_collective(model, model.lin1.weight.device, cache)  # SyncBN: the engine needs its all-reduce
    return _EngineNode.apply(model, data, cache, *cache.node_inputs())


# ---- ZINC variant (zinc_models.NestedGIN_eff; csrc/engine.hip esc_zinc_*) ---------------------------------------------
def _embed(mod, gp=_grad_ptr):
    s = _Embed()
    s.w, s.dw = mod.weight.data_ptr(), gp(mod.weight)
    s.rows, s.dim = mod.num_embeddings, mod.embedding_dim
    return s


def describe_zinc(m, gp=_grad_ptr):
    """esc_zinc_gin_t of a zinc_models.NestedGIN_eff module"""
    d = _ZincModel()
    convs = [m.conv1] + list(m.convs)
    if len(convs) > MAX_LAYERS:
        raise ValueError("at most %d layers" % MAX_LAYERS)
    d.num_layers, d.hidden, d.z_rows = len(convs), m.lin2.in_features, m.z_initial.num_embeddings
    d.z_table, d.dz_table = m.z_initial.weight.data_ptr(), gp(m.z_initial.weight)
    d.zbn0, d.zlin, d.zbn1 = _bn(m.z_embedding[1], gp), _lin(m.z_embedding[3], gp), _bn(m.z_embedding[5], gp)
    d.node_emb, d.edge_emb = _embed(m.node_type_embedding, gp), _embed(m.edge_type_embedding, gp)
    for i, cv in enumerate(convs):
        c = _Conv()
        c.eps, c.deps = cv.eps.data_ptr(), gp(cv.eps)
        c.nn, c.lin = _mlp(cv.nn, gp), _lin(cv.lin, gp)
        d.conv[i] = c
    d.lin1, d.bn_lin1, d.lin2 = _lin(m.lin1, gp), _bn(m.bn_lin1, gp), _lin(m.lin2, gp)
    return d


def zinc_engine_supports(m, data=None):
    """what esc_zinc_* covers: the run_zinc configuration (dropout 0; BatchNorm statistics per rank or, after
    nn.BatchNorm1d.convert_sync, over ONE process group through the collective provider), sparse ESC bag, at least two graphs
    in the batch (the reference skips bn_lin1 for one, zinc_models.py:603-604)"""
    if m.dropout != 0 or m.lin1.weight.device.type != "cuda" or m.lin2.out_features != 1 or not _one_sync_group(m):
        return False
    if data is not None:
        if "edge_pos" in data or "pos_batch" not in data or data.edge_index.size(1) < 2 or data["edge_attr"] is None:
            return False
    return True


def zinc_engine_ready(m, data):
    """zinc_engine_supports and at least two graphs in this batch"""
    if not zinc_engine_supports(m, data):
        return False
    from .plan import graph_ptr_of
    from .run_graphcount import Z_TABLE_ROWS
    return graph_ptr_of(data, plan_of(data, Z_TABLE_ROWS))[1] >= 2


def _zinc_batch(model, data, need_y):
    from .plan import graph_ptr_of
    from .run_graphcount import Z_TABLE_ROWS
    dev = model.lin1.weight.device
    if data.edge_index.device != dev:
        data.to(dev)
    plan = plan_of(data, Z_TABLE_ROWS)
    gptr, G = graph_ptr_of(data, plan)
    b = _MolBatch()
    b.N, b.E, b.Z, b.G = plan.num_nodes, plan.num_edges, plan.nnz, G
    nt = data.x.reshape(-1)
    et = data.edge_attr.reshape(-1)
    nt = nt if (nt.dtype == torch.int64 and nt.is_contiguous()) else nt.to(torch.int64).contiguous()
    et = et if (et.dtype == torch.int64 and et.is_contiguous()) else et.to(torch.int64).contiguous()
    if nt.numel() != b.N or et.numel() != b.E:
        raise ValueError("ZINC engine: expected one type id per node and per edge")
    # the lookup kernels turn an out-of-range id into a zero row (in bounds, but silent): ids are trusted only when the
    # device store signed these very tensors with a dataset-wide range inside the tables; anything else is checked once
    # per tensor version (one read-back), like torch.nn.Embedding's IndexError on the per-op path
    rng, sig = data.__dict__.get("_esc_int_ranges") or ({}, {})
    for key, src, ids, rows in (("x", data.x, nt, model.node_type_embedding.num_embeddings),
                                ("edge_attr", data.edge_attr, et, model.edge_type_embedding.num_embeddings)):
        r = rng.get(key) if sig.get(key) == (src.data_ptr(), src._version) else None
        if r is not None and min(r[0]) >= 0 and max(r[1]) < rows:
            continue
        seen = getattr(src, "_esc_zinc_checked", None)
        if seen == (src._version, rows):
            continue
        if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= rows):
            raise IndexError("ZINC engine: %s holds a type id outside the %d-row embedding table" % (key, rows))
        src._esc_zinc_checked = (src._version, rows)
    b.node_type, b.edge_type, b.graph_ptr = nt.data_ptr(), et.data_ptr(), gptr.data_ptr()
    y = None
    if need_y:
        y = data.y.reshape(-1)
        y = y if (y.dtype == torch.float32 and y.is_contiguous()) else y.float().contiguous()
        if y.numel() != G:
            raise ValueError("ZINC engine: expected one target per graph")
        b.y = y.data_ptr()
    for f in ("in_ptr", "in_edge", "in_src", "out_ptr", "out_edge", "out_dst", "row_ptr", "bag_idx", "bag_val",
              "col_ptr", "col_row", "col_val", "col_col"):
        setattr(b, f, getattr(plan, f).data_ptr())
    return b, (nt, et, y, plan, gptr)


class ZincStepEngine(_AddressGuard):
    """Training / eval step of zinc_models.NestedGIN_eff as ONE call (esc_zinc_train_step / esc_zinc_predict): same
    parameters, `.grad` slots and BatchNorm buffers as the module, like StepEngine for the counting model."""

    def __init__(self, model):
        if not zinc_engine_supports(model):
            raise NotImplementedError("ZincStepEngine covers dropout 0, lin2 -> 1 output, BatchNorm on one sync group, on the HIP device")
        self.model = model
        self._ws = None
        _arm_collective(model, model.lin1.weight.device)
        self._bn_counters = [m.num_batches_tracked for m in model.modules()
                             if isinstance(m, torch.nn.BatchNorm1d) and m.num_batches_tracked is not None]
        self.refresh()

    def refresh(self):
        self._desc = describe_zinc(self.model)
        self._guard_arm()

    def _workspace(self, b):
        need = nv.lib().esc_zinc_workspace_floats(ctypes.byref(self._desc), b.N, b.E, b.Z, b.G)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(int(need * 1.25), dtype=torch.float32, device=self.model.lin1.weight.device)
        return self._ws

    def prepare(self, data):
        """per-batch plans a prefetching loader can build ahead (harness.prefetched): the collate's own plan is all this
        model needs"""
        _zinc_batch(self.model, data, False)
        return data

    def train_step(self, data, loss_denom=None, return_pred=False):
        """forward + L1 over the graphs + backward; gradients land in the parameters' .grad (overwritten)"""
        dev = self.model.lin1.weight.device
        self._guard_check()
        b, keep = _zinc_batch(self.model, data, True)
        ws = self._workspace(b)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        pred = torch.empty(b.G, dtype=torch.float32, device=dev) if return_pred else None
        nv.call("esc_zinc_train_step", ctypes.byref(self._desc), ctypes.byref(b), ws.data_ptr(), int(loss_denom or 0),
                loss.data_ptr(), nv.ptr(pred), nv.stream())
        self._mark_bucket_written()
        if self._bn_counters:
            torch._foreach_add_(self._bn_counters, 1)
        return (loss.view(()), pred.view(-1, 1)) if return_pred else loss.view(())

    @torch.no_grad()
    def predict(self, data):
        dev = self.model.lin1.weight.device
        self._guard_check()
        b, keep = _zinc_batch(self.model, data, False)
        ws = self._workspace(b)
        pred = torch.empty(b.G, dtype=torch.float32, device=dev)
        nv.call("esc_zinc_predict", ctypes.byref(self._desc), ctypes.byref(b), ws.data_ptr(), pred.data_ptr(), nv.stream())
        return pred.view(-1, 1)


class _ZincNodeCache(_NodeCache):
    def __init__(self, model):
        self._describe, self._struct = describe_zinc, _ZincModel
        super().__init__(model)


class _ZincEngineNode(torch.autograd.Function):
    """`model(batch)` of a training-mode ZINC NestedGIN_eff as one autograd node (esc_zinc_forward_train / _backward)"""

    @staticmethod
    def forward(ctx, model, data, cache, *params):
        dev = model.lin1.weight.device
        b, keep = _zinc_batch(model, data, False)
        desc = cache.descriptor(0)
        need = nv.lib().esc_zinc_workspace_floats(ctypes.byref(desc), b.N, b.E, b.Z, b.G)
        ws = torch.empty(int(need), dtype=torch.float32, device=dev)
        pred = torch.empty(b.G, dtype=torch.float32, device=dev)
        nv.call("esc_zinc_forward_train", ctypes.byref(desc), ctypes.byref(b), ws.data_ptr(), pred.data_ptr(), nv.stream())
        if cache.counters:
            torch._foreach_add_(cache.counters, 1)
        ctx.cache, ctx.b, ctx.keep, ctx.ws, ctx.n_in = cache, b, keep, ws, len(params)
        return pred.view(-1, 1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dpred):
        return _node_backward(ctx, dpred, "esc_zinc_backward")


def _mol_cache(model, kind):
    c = model.__dict__.get("_esc_node_cache")
    if c is None or not c.valid():
        c = kind(model)
        model.__dict__["_esc_node_cache"] = c
    return c


@torch.no_grad()
def zinc_engine_predict(model, data):
    """eval-mode forward of the ZINC model as one call (esc_zinc_predict)"""
    cache = _mol_cache(model, _ZincNodeCache)
    b, keep = _zinc_batch(model, data, False)
    desc = cache.descriptor(0)
    need = nv.lib().esc_zinc_workspace_floats(ctypes.byref(desc), b.N, b.E, b.Z, b.G)
    dev = model.lin1.weight.device
    ws = torch.empty(int(need), dtype=torch.float32, device=dev)
    pred = torch.empty(b.G, dtype=torch.float32, device=dev)
    nv.call("esc_zinc_predict", ctypes.byref(desc), ctypes.byref(b), ws.data_ptr(), pred.data_ptr(), nv.stream())
    return pred.view(-1, 1)


def zinc_engine_forward(model, data):
    c = model.__dict__.get("_esc_node_cache")
    if c is None or not c.valid():
        c = _ZincNodeCache(model)
        model.__dict__["_esc_node_cache"] = c
    _arm_collective(model, model.lin1.weight.device, c)
    return _ZincEngineNode.apply(model, data, c, *c.node_inputs())


# ---- OGB molecule variant (ogb_mol_gnn.GNN(gnn_type="gin_eff"); csrc/engine.hip esc_ogb_*) -----------------------------
MAX_TABLES = 64


class _TableList(ctypes.Structure):
    _fields_ = [("count", c_int32), ("rows", c_int32 * MAX_TABLES), ("w", c_void_p * MAX_TABLES), ("dw", c_void_p * MAX_TABLES)]


class _OgbLayer(ctypes.Structure):
    _fields_ = [("eps", c_void_p), ("deps", c_void_p), ("pos", _Linear), ("lin0", _Linear), ("bn0", _BN), ("lin1", _Linear),
                ("bn", _BN), ("vlin0", _Linear), ("vbn0", _BN), ("vlin1", _Linear), ("vbn1", _BN), ("bond_row0", c_int64)]


class _OgbModel(ctypes.Structure):
    _fields_ = [("num_layers", c_int64), ("hidden", c_int64), ("z_rows", c_int64), ("num_tasks", c_int64),
                ("residual", c_int32), ("mean_pool", c_int32), ("drop_ratio", c_float), ("pad_", c_int32),
                ("z_table", c_void_p), ("dz_table", c_void_p), ("zbn0", _BN), ("zlin", _Linear), ("zbn1", _BN),
                ("tables", _TableList), ("atom_rows", c_int64), ("bond_rows", c_int64), ("vn_w", c_void_p), ("vn_dw", c_void_p),
                ("layer", _OgbLayer * MAX_LAYERS), ("head", _Linear)]


class _BagPlan(ctypes.Structure):
    _fields_ = [("n_entries", c_int64)] + [(n, c_void_p) for n in ("row_ptr", "idx", "ones", "col_ptr", "c_row", "c_col")]


class _OgbBatch(ctypes.Structure):
    _fields_ = ([("N", c_int64), ("E", c_int64), ("Z", c_int64), ("G", c_int64), ("atoms", _BagPlan), ("bonds", _BagPlan),
                 ("y", c_void_p), ("graph_ptr", c_void_p), ("zero_idx", c_void_p)] +
                [(n, c_void_p) for n in ("in_ptr", "in_edge", "in_src", "out_ptr", "out_edge", "out_dst",
                                         "row_ptr", "bag_idx", "bag_val", "col_ptr", "col_row", "col_val", "col_col")] +
                [("seed", c_uint64)])


def ogb_engine_supports(m, data=None):
    """what esc_ogb_* covers: the run_ogb_mol `--gnn gin_eff` configuration — ogbg-mol* encoders, virtual node, JK last,
    sum / mean pooling, BatchNorm statistics per rank or over ONE process group (nn.BatchNorm1d.convert_sync); sparse ESC bag;
    at least two graphs per batch"""
    from .ogb_mol_gnn import AtomEncoder, BondEncoder
    g = m.gnn_node
    if g.JK != "last" or not g.virtual_node or g.skip_node_encoder or m.graph_pooling not in ("sum", "mean") or not _one_sync_group(m):
        return False
    if m.graph_pred_linear.weight.device.type != "cuda" or m.emb_dim % 4 != 0 or not isinstance(g.node_encoder, AtomEncoder):
        return False
    if not all(isinstance(cv.edge_encoder, BondEncoder) for cv in g.convs):
        return False
    if data is not None:
        if "edge_pos" in data or "pos_batch" not in data or data.edge_index.size(1) < 2 or data["edge_attr"] is None:
            return False
        if data.x.dim() != 2 or data.x.size(1) != 9 or data.edge_attr.dim() != 2 or data.edge_attr.size(1) != 3:
            return False
    return True


def describe_ogb(m, gp=_grad_ptr):
    """esc_ogb_gnn_t of an ogb_mol_gnn.GNN module"""
    g = m.gnn_node
    L = g.num_layer
    if L > MAX_LAYERS:
        raise ValueError("at most %d layers" % MAX_LAYERS)
    d = _OgbModel()
    d.num_layers, d.hidden, d.z_rows, d.num_tasks = L, m.emb_dim, g.z_initial.num_embeddings, m.num_tasks
    d.residual, d.mean_pool, d.drop_ratio = int(bool(g.residual)), int(m.graph_pooling == "mean"), float(g.drop_ratio)
    d.z_table, d.dz_table = g.z_initial.weight.data_ptr(), gp(g.z_initial.weight)
    d.zbn0, d.zlin, d.zbn1 = _bn(g.z_embedding[1], gp), _lin(g.z_embedding[3], gp), _bn(g.z_embedding[5], gp)
    tabs = list(g.node_encoder.atom_embedding_list)
    d.atom_rows = sum(t.num_embeddings for t in tabs)
    bond_rows = sum(t.num_embeddings for t in g.convs[0].edge_encoder.bond_embedding_list)
    d.bond_rows = bond_rows
    for l, cv in enumerate(g.convs):
        tabs += list(cv.edge_encoder.bond_embedding_list)
    if len(tabs) > MAX_TABLES:
        raise ValueError("too many embedding tables for the OGB engine")
    d.tables.count = len(tabs)
    for j, t in enumerate(tabs):
        d.tables.rows[j], d.tables.w[j], d.tables.dw[j] = t.num_embeddings, t.weight.data_ptr(), gp(t.weight)
    d.vn_w, d.vn_dw = g.virtualnode_embedding.weight.data_ptr(), gp(g.virtualnode_embedding.weight)
    for l, cv in enumerate(g.convs):
        q = _OgbLayer()
        q.eps, q.deps = cv.eps.data_ptr(), gp(cv.eps)
        q.pos = _lin(cv.edge_encoder_pos, gp)
        q.lin0, q.bn0, q.lin1 = _lin(cv.mlp[0], gp), _bn(cv.mlp[1], gp), _lin(cv.mlp[3], gp)
        q.bn = _bn(g.batch_norms[l], gp)
        if l < L - 1:
            v = g.mlp_virtualnode_list[l]
            q.vlin0, q.vbn0, q.vlin1, q.vbn1 = _lin(v[0], gp), _bn(v[1], gp), _lin(v[3], gp), _bn(v[4], gp)
        q.bond_row0 = d.atom_rows + l * bond_rows
        d.layer[l] = q
    d.head = _lin(m.graph_pred_linear, gp)
    return d


def _bag_plan(plan):
    b = _BagPlan()
    b.n_entries = plan["entries"]
    b.row_ptr, b.idx, b.ones = plan["row_ptr"].data_ptr(), plan["idx32"].data_ptr(), plan["ones"].data_ptr()
    b.col_ptr, b.c_row, b.c_col = plan["col_ptr"].data_ptr(), plan["c_row"].data_ptr(), plan["c_col"].data_ptr()
    return b


_zero_idx = {}


def _ogb_batch(model, data, need_y, seed):
    from .ogb_mol_gnn import ATOM_FEATURE_DIMS, BOND_FEATURE_DIMS
    from .ops import embed_plan
    from .plan import graph_ptr_of
    from .run_graphcount import Z_TABLE_ROWS
    dev = model.graph_pred_linear.weight.device
    if data.edge_index.device != dev:
        data.to(dev)
    plan = plan_of(data, Z_TABLE_ROWS)
    gptr, G = graph_ptr_of(data, plan)
    rng, sig = data.__dict__.get("_esc_int_ranges") or ({}, {})      # from the device store: no per-batch range read-back

    def known(key):                                          # ... as long as the tensor is the collate's, unedited
        t = data[key]
        return rng.get(key) if sig.get(key) == (t.data_ptr(), t._version) else None
    pa = embed_plan(data.x, ATOM_FEATURE_DIMS, known("x"))
    pb = embed_plan(data.edge_attr, BOND_FEATURE_DIMS, known("edge_attr"))
    b = _OgbBatch()
    b.N, b.E, b.Z, b.G = plan.num_nodes, plan.num_edges, plan.nnz, G
    if data.x.size(0) != b.N or data.edge_attr.size(0) != b.E:
        raise ValueError("OGB engine: expected one feature row per node and per edge")
    b.atoms, b.bonds = _bag_plan(pa), _bag_plan(pb)
    zero = _zero_idx.get(dev)
    if zero is None or zero.numel() < G:
        zero = _zero_idx[dev] = torch.zeros(max(G, 1024), dtype=torch.int64, device=dev)
    b.graph_ptr, b.zero_idx, b.seed = gptr.data_ptr(), zero.data_ptr(), int(seed) & ((1 << 64) - 1)
    y = None
    if need_y:
        y = data.y.reshape(G, -1)
        y = y if (y.dtype == torch.float32 and y.is_contiguous()) else y.float().contiguous()
        if y.size(1) != model.num_tasks:
            raise ValueError("OGB engine: expected [num_graphs, num_tasks] targets")
        b.y = y.data_ptr()
    for f in ("in_ptr", "in_edge", "in_src", "out_ptr", "out_edge", "out_dst", "row_ptr", "bag_idx", "bag_val",
              "col_ptr", "col_row", "col_val", "col_col"):
        setattr(b, f, getattr(plan, f).data_ptr())
    return b, (y, plan, gptr, pa, pb, zero)


def ogb_engine_ready(m, data):
    if not ogb_engine_supports(m, data):
        return False
    from .plan import graph_ptr_of
    from .run_graphcount import Z_TABLE_ROWS
    return graph_ptr_of(data, plan_of(data, Z_TABLE_ROWS))[1] >= 2


def _drop_seed(model):
    """dropout stream of the next step: torch's seed (torch.manual_seed makes runs repeatable) + a per-model step counter"""
    n = model.__dict__.get("_esc_drop_step", 0)
    model.__dict__["_esc_drop_step"] = n + 1
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + n) & ((1 << 64) - 1)


class OgbStepEngine(_AddressGuard):
    """Training / eval step of ogb_mol_gnn.GNN(gnn_type='gin_eff') as ONE call (esc_ogb_train_step / esc_ogb_predict)"""

    def __init__(self, model):
        if not ogb_engine_supports(model):
            raise NotImplementedError("OgbStepEngine covers ogbg-mol* gin_eff with a virtual node, JK=last, sum/mean pooling")
        self.model = model
        self._ws = None
        _arm_collective(model, model.graph_pred_linear.weight.device)
        self._bn_counters = [m.num_batches_tracked for m in model.modules()
                             if isinstance(m, torch.nn.BatchNorm1d) and m.num_batches_tracked is not None]
        self.refresh()

    def refresh(self):
        self._desc = describe_ogb(self.model)
        self._guard_arm()

    def _workspace(self, b):
        need = nv.lib().esc_ogb_workspace_floats(ctypes.byref(self._desc), b.N, b.E, b.Z, b.G, b.atoms.n_entries, b.bonds.n_entries)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(int(need * 1.25), dtype=torch.float32, device=self.model.graph_pred_linear.weight.device)
        return self._ws

    def prepare(self, data):
        """builds (and caches on the batch's tensors) every index plan train_step / predict will ask for — the call a
        prefetching loader makes on its side stream (harness.prefetched)"""
        _ogb_batch(self.model, data, False, 0)
        return data

    def train_step(self, data, loss_denom=None, return_pred=False):
        """forward + masked BCE-with-logits + backward; gradients land in the parameters' .grad (overwritten)"""
        dev = self.model.graph_pred_linear.weight.device
        self._guard_check()
        b, keep = _ogb_batch(self.model, data, True, _drop_seed(self.model))
        ws = self._workspace(b)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        pred = torch.empty((b.G, self.model.num_tasks), dtype=torch.float32, device=dev) if return_pred else None
        nv.call("esc_ogb_train_step", ctypes.byref(self._desc), ctypes.byref(b), ws.data_ptr(), int(loss_denom or 0),
                loss.data_ptr(), nv.ptr(pred), nv.stream())
        self._mark_bucket_written()
        if self._bn_counters:
            torch._foreach_add_(self._bn_counters, 1)
        return (loss.view(()), pred) if return_pred else loss.view(())

    @torch.no_grad()
    def predict(self, data):
        dev = self.model.graph_pred_linear.weight.device
        self._guard_check()
        b, keep = _ogb_batch(self.model, data, False, 0)
        ws = self._workspace(b)
        pred = torch.empty((b.G, self.model.num_tasks), dtype=torch.float32, device=dev)
        nv.call("esc_ogb_predict", ctypes.byref(self._desc), ctypes.byref(b), ws.data_ptr(), pred.data_ptr(), nv.stream())
        return pred


class _OgbNodeCache(_NodeCache):
    def __init__(self, model):
        self._describe, self._struct = describe_ogb, _OgbModel
        super().__init__(model)


class _OgbEngineNode(torch.autograd.Function):
    """`model(batch)` of a training-mode OGB GNN as one autograd node (esc_ogb_forward_train / esc_ogb_backward)"""

    @staticmethod
    def forward(ctx, model, data, cache, *params):
        dev = model.graph_pred_linear.weight.device
        b, keep = _ogb_batch(model, data, False, _drop_seed(model))
        desc = cache.descriptor(0)
        need = nv.lib().esc_ogb_workspace_floats(ctypes.byref(desc), b.N, b.E, b.Z, b.G, b.atoms.n_entries, b.bonds.n_entries)
        ws = torch.empty(int(need), dtype=torch.float32, device=dev)
        pred = torch.empty((b.G, model.num_tasks), dtype=torch.float32, device=dev)
        nv.call("esc_ogb_forward_train", ctypes.byref(desc), ctypes.byref(b), ws.data_ptr(), pred.data_ptr(), nv.stream())
        if cache.counters:
            torch._foreach_add_(cache.counters, 1)
        ctx.cache, ctx.b, ctx.keep, ctx.ws, ctx.n_in = cache, b, keep, ws, len(params)
        return pred

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dpred):
        return _node_backward(ctx, dpred, "esc_ogb_backward")


@torch.no_grad()
def ogb_engine_predict(model, data):
    """eval-mode forward of the OGB model as one call (esc_ogb_predict: no dropout, running statistics)"""
    cache = _mol_cache(model, _OgbNodeCache)
    b, keep = _ogb_batch(model, data, False, 0)
    desc = cache.descriptor(0)
    need = nv.lib().esc_ogb_workspace_floats(ctypes.byref(desc), b.N, b.E, b.Z, b.G, b.atoms.n_entries, b.bonds.n_entries)
    dev = model.graph_pred_linear.weight.device
    ws = torch.empty(int(need), dtype=torch.float32, device=dev)
    pred = torch.empty((b.G, model.num_tasks), dtype=torch.float32, device=dev)
    nv.call("esc_ogb_predict", ctypes.byref(desc), ctypes.byref(b), ws.data_ptr(), pred.data_ptr(), nv.stream())
    return pred


def ogb_engine_forward(model, data):
    c = model.__dict__.get("_esc_node_cache")
    if c is None or not c.valid():
        c = _OgbNodeCache(model)
        model.__dict__["_esc_node_cache"] = c
    _arm_collective(model, model.graph_pred_linear.weight.device, c)
    return _OgbEngineNode.apply(model, data, c, *c.node_inputs())

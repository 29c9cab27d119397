"""Execution plan of a batch: the compact int32 CSR/CSC views the HIP kernels walk.

The reference never builds these — PyG's propagate gathers by `edge_index[0]` and scatter-adds by
`edge_index[1]` in edge order (run_graphcount.py:161,169) and the bag scatter-adds by `pos_batch`
(:155).  Atomic scatters cap at ~16 % of HBM bandwidth on gfx950, so the batch is viewed through
*stable* sorted permutations instead (stable => per-destination sums run in ascending edge order,
i.e. bit-identical to a sequential scatter).  User-visible tensors are never reordered.

  in_*   : edges grouped by destination  (aggregate forward)
  out_*  : edges grouped by source       (aggregate backward: dx and d_e in one pass)
  row_*  : bag entries grouped by edge   (bag forward; pos_batch is already non-decreasing)
  col_*  : bag entries grouped by histogram bin (table gradient)

The fast path gets these from the device collate (DeviceGraphStore.collate); for a foreign batch
they are derived here by esc_plan_csr (csrc/plan.hip), once per batch and cached on the Data object.
"""
import torch


def _csr(key, n_keys, want_perm=True, bad=None):
    """stable grouping of positions 0..len(key)-1 by key -> (ptr int32[n_keys+1], perm int32 or None), on the device:
    esc_plan_csr (csrc/plan.hip: LSD radix passes of the positions + an integer histogram), no library sort.
    Raises IndexError when a key lies outside [0, n_keys) — or, when the caller passes `bad` (a 1-element int32 device
    tensor), only sets it, so that several calls can share ONE host read-back."""
    from . import _native as nv
    if key.device.type != "cuda":
        raise RuntimeError("esc_gnn_amd: execution plans are built on the GPU (got a %s tensor) - move the batch to the "
                           "device first" % key.device.type)
    key = key.contiguous()
    if key.dtype != torch.int64:
        key = key.to(torch.int64)
    n, dev = key.numel(), key.device
    ptr = torch.empty(n_keys + 1, dtype=torch.int32, device=dev)
    perm = torch.empty(n, dtype=torch.int32, device=dev) if want_perm else None
    scratch = torch.empty(nv.lib().esc_plan_csr_scratch(n, n_keys), dtype=torch.int32, device=dev)
    deferred = bad is not None
    if not deferred:
        bad = torch.empty(1, dtype=torch.int32, device=dev)
    nv.call("esc_plan_csr", nv.ptr(key), n, n_keys, nv.ptr(ptr), nv.ptr(perm), nv.ptr(scratch), nv.ptr(bad), nv.stream())
    if not deferred and int(bad.item()):
        raise IndexError("plan: a key lies outside [0, %d)" % n_keys)
    return ptr, perm


class BatchPlan(object):
    FIELDS = ("in_ptr", "in_edge", "in_src", "out_ptr", "out_edge", "out_dst",
              "row_ptr", "bag_idx", "bag_val", "col_ptr", "col_row", "col_val", "col_col")

    def __init__(self, **kw):
        for f in self.FIELDS:
            setattr(self, f, kw.get(f))
        self.num_nodes = kw["num_nodes"]
        self.num_edges = kw["num_edges"]
        self.nnz = kw["nnz"]
        self.n_cols = kw.get("n_cols", 1800)
        self.graph_ptr, self.num_graphs = None, None      # node range of every graph: set by the device collate / graph_ptr_of
        self._batch_sig = None                            # ... valid for exactly this `batch` vector (address, shape, version)

    def to(self, device):
        for f in self.FIELDS:
            v = getattr(self, f)
            if v is not None:
                setattr(self, f, v.to(device))
        if self.graph_ptr is not None:
            self.graph_ptr = self.graph_ptr.to(device)
        return self

    @staticmethod
    def from_tensors(edge_index, num_nodes, pos_enc=None, pos_index=None, pos_batch=None, n_cols=1800):
        src, dst = edge_index[0], edge_index[1]
        E = src.numel()
        flags = torch.zeros(5, dtype=torch.int32, device=src.device)      # one host read-back for all the range checks
        in_ptr, in_edge = _csr(dst, num_nodes, bad=flags[0:1])
        out_ptr, out_edge = _csr(src, num_nodes, bad=flags[1:2])
        kw = dict(in_ptr=in_ptr, in_edge=in_edge, in_src=src[in_edge.long()].to(torch.int32),
                  out_ptr=out_ptr, out_edge=out_edge, out_dst=dst[out_edge.long()].to(torch.int32),
                  num_nodes=int(num_nodes), num_edges=int(E), nnz=0, n_cols=n_cols)
        if pos_batch is not None:
            Z = pos_batch.numel()
            if Z > 1:
                flags[4:5] = (pos_batch[1:] < pos_batch[:-1]).any().to(torch.int32)
            row_ptr, _ = _csr(pos_batch, E, want_perm=False, bad=flags[2:3])        # already grouped by edge: pointers only
            col_ptr, col_perm = _csr(pos_index, n_cols, bad=flags[3:4])
            cp = col_perm.long()
            kw.update(row_ptr=row_ptr, bag_idx=pos_index.to(torch.int32), bag_val=pos_enc.to(torch.int32),
                      col_ptr=col_ptr, col_row=pos_batch[cp].to(torch.int32),
                      col_val=pos_enc[cp].to(torch.int32), col_col=pos_index[cp].to(torch.int32), nnz=int(Z))
        f = flags.tolist()
        if f[0] or f[1]:
            raise IndexError("plan: edge_index refers to a node outside [0, %d)" % num_nodes)
        if f[4]:
            raise ValueError("pos_batch must be non-decreasing (as create_subgraphs emits it)")
        if f[2]:
            raise ValueError("pos_batch refers to an edge beyond the batch's %d edges" % E)
        if f[3]:
            raise IndexError("pos_index outside the %d-row z_initial table" % n_cols)
        return BatchPlan(**kw)


def _sig(t):
    return None if t is None else (t.data_ptr(), tuple(t.shape), t._version, str(t.device))


def plan_key(data, n_cols=1800):
    """identity of everything a plan was derived from: (address, shape, in-place version, device) of edge_index and the
    bag tensors, the table height and the node count"""
    return (_sig(data.edge_index), _sig(data["pos_enc"]), _sig(data["pos_index"]), _sig(data["pos_batch"]), n_cols,
            data.num_nodes if data.x is None else data.x.size(0))


def plan_of(data, n_cols=1800):
    """Cached plan of a Data/Batch (built on first use, on the device the tensors live on).  The cache is keyed on
    plan_key(): assigning or editing edge_index / pos_* — edge dropout, augmentation, a re-used Batch object — rebuilds
    the plan instead of aggregating over a stale one (which would read out-of-range int32 indices on the GPU)."""
    d = object.__getattribute__(data, "__dict__")
    cached = d.get("_esc_plan")
    key = plan_key(data, n_cols)
    if cached is not None and getattr(cached, "_key", None) == key:
        return cached
    plan = BatchPlan.from_tensors(data.edge_index, key[5], data["pos_enc"], data["pos_index"], data["pos_batch"], n_cols)
    plan._key = key
    object.__setattr__(data, "_esc_plan", plan)
    return plan


def graph_ptr_of(data, plan):
    """int32 [G+1] node range of every graph of a batch (its `batch` vector is non-decreasing) and G; from the device
    collate when the batch came from there, else one esc_plan_csr call, cached on the plan."""
    if plan.graph_ptr is None or plan._batch_sig != _sig(data.batch):     # a re-assigned / edited `batch` moves the graph bounds
        batch = data.batch
        G = int(batch[-1].item()) + 1 if batch.numel() else 0
        if batch.numel() > 1 and not bool((batch[1:] >= batch[:-1]).all()):
            raise ValueError("the batch vector must be non-decreasing")
        plan.graph_ptr, plan.num_graphs = _csr(batch, max(G, 1), want_perm=False)[0], G
        plan._batch_sig = _sig(batch)
    return plan.graph_ptr, plan.num_graphs

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
they are derived here with torch sorts (host-side plumbing, once per batch).
"""
import torch


def _csr(key, n_keys):
    """stable grouping of positions 0..len(key)-1 by key -> (ptr int32[n_keys+1], perm int32)."""
    order = torch.sort(key, stable=True)[1]
    counts = torch.bincount(key, minlength=n_keys)
    ptr = torch.zeros(n_keys + 1, dtype=torch.int32, device=key.device)
    ptr[1:] = torch.cumsum(counts, 0)
    return ptr, order.to(torch.int32)


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

    def to(self, device):
        for f in self.FIELDS:
            v = getattr(self, f)
            if v is not None:
                setattr(self, f, v.to(device))
        return self

    @staticmethod
    def from_tensors(edge_index, num_nodes, pos_enc=None, pos_index=None, pos_batch=None, n_cols=1800):
        src, dst = edge_index[0], edge_index[1]
        E = src.numel()
        in_ptr, in_edge = _csr(dst, num_nodes)
        out_ptr, out_edge = _csr(src, num_nodes)
        kw = dict(in_ptr=in_ptr, in_edge=in_edge, in_src=src[in_edge.long()].to(torch.int32),
                  out_ptr=out_ptr, out_edge=out_edge, out_dst=dst[out_edge.long()].to(torch.int32),
                  num_nodes=int(num_nodes), num_edges=int(E), nnz=0, n_cols=n_cols)
        if pos_batch is not None:
            Z = pos_batch.numel()
            counts = torch.bincount(pos_batch, minlength=E)
            if counts.numel() != E:
                raise ValueError("pos_batch refers to edge %d but the batch has %d edges" % (counts.numel() - 1, E))
            row_ptr = torch.zeros(E + 1, dtype=torch.int32, device=src.device)
            row_ptr[1:] = torch.cumsum(counts, 0)
            if Z and not bool((pos_batch[1:] >= pos_batch[:-1]).all()):
                raise ValueError("pos_batch must be non-decreasing (as create_subgraphs emits it)")
            if Z and (int(pos_index.min()) < 0 or int(pos_index.max()) >= n_cols):
                raise IndexError("pos_index outside the %d-row z_initial table" % n_cols)
            col_ptr, col_perm = _csr(pos_index, n_cols)
            cp = col_perm.long()
            kw.update(row_ptr=row_ptr, bag_idx=pos_index.to(torch.int32), bag_val=pos_enc.to(torch.int32),
                      col_ptr=col_ptr, col_row=pos_batch[cp].to(torch.int32),
                      col_val=pos_enc[cp].to(torch.int32), col_col=pos_index[cp].to(torch.int32), nnz=int(Z))
        return BatchPlan(**kw)


def plan_of(data, n_cols=1800):
    """Cached plan of a Data/Batch (built on first use, on the device the tensors live on)."""
    d = object.__getattribute__(data, "__dict__")
    plan = d.get("_esc_plan")
    ei = data.edge_index
    if plan is None or plan.in_ptr.device != ei.device:
        plan = BatchPlan.from_tensors(ei, data.num_nodes if data.x is None else data.x.size(0),
                                      data["pos_enc"], data["pos_index"], data["pos_batch"], n_cols)
        object.__setattr__(data, "_esc_plan", plan)
    return plan

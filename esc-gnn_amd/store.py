"""DeviceGraphStore — the pre-processed dataset resident in HBM + the device collate.

The reference caches a pre-transformed dataset as PyG's `(data, slices)` pair
(GraphCountDataset.py:119-120: per-key concatenation + slice pointers) in host memory and re-collates
every mini-batch in python (batch.py:25-149, dataloader.py:26-29) before one H2D copy per tensor.
With 288 GB of HBM the whole dataset stays on the device in that same concatenated layout, and a
mini-batch is ONE gather kernel (csrc/collate.hip) that produces the reference's batch tensors plus the
compact execution plan.  The sorted per-graph views the plan needs are derived once, here.
"""
import ctypes

import numpy as np
import torch

from . import _native as nv
from .batch import Batch
from .plan import BatchPlan, plan_key, _sig

N_COLS = 1800


def _stable_group(key, n_keys):
    """positions grouped by key (stable) -> (ptr int64[n_keys+1], perm int64)."""
    perm = torch.sort(key, stable=True)[1]
    ptr = torch.zeros(n_keys + 1, dtype=torch.int64, device=key.device)
    ptr[1:] = torch.cumsum(torch.bincount(key, minlength=n_keys), 0)
    return ptr, perm


class DeviceGraphStore(object):
    def __init__(self, data_list, device):
        if len(data_list) == 0:
            raise ValueError("DeviceGraphStore: empty dataset")
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("DeviceGraphStore lives in HBM; there is no CPU fallback")
        self.device = dev
        G = len(data_list)
        n = torch.tensor([d.x.size(0) for d in data_list], dtype=torch.int64)
        e = torch.tensor([d.edge_index.size(1) for d in data_list], dtype=torch.int64)
        z = torch.tensor([d.pos_enc.numel() for d in data_list], dtype=torch.int64)
        ys = [d.y.reshape(d.y.size(0), -1) if d.y.dim() > 0 else d.y.reshape(1, 1) for d in data_list]
        yr = torch.tensor([t.size(0) for t in ys], dtype=torch.int64)

        def ptr(c):
            p = torch.zeros(G + 1, dtype=torch.int64)
            p[1:] = torch.cumsum(c, 0)
            return p
        self.h_node_ptr, self.h_edge_ptr, self.h_nnz_ptr, self.h_y_ptr = ptr(n), ptr(e), ptr(z), ptr(yr)
        self.node_ptr, self.edge_ptr = self.h_node_ptr.to(dev), self.h_edge_ptr.to(dev)
        self.nnz_ptr, self.y_ptr = self.h_nnz_ptr.to(dev), self.h_y_ptr.to(dev)
        self.num_graphs = G
        self.x_is_int = not data_list[0].x.is_floating_point()       # categorical node features (ZINC / OGB)
        self.x_was_1d = data_list[0].x.dim() == 1
        self.x_all = torch.cat([d.x.reshape(d.x.size(0), -1).float() for d in data_list]).contiguous().to(dev)
        ea = [d.edge_attr for d in data_list]
        self.edge_attr_all = None if any(a is None for a in ea) else torch.cat(ea, dim=0).contiguous().to(dev)
        self._compute_int_ranges()
        self.y_all = torch.cat(ys).float().contiguous().to(dev)
        self.x_dim, self.y_dim = self.x_all.size(1), self.y_all.size(1)
        self.y_is_vector = all(d.y.dim() <= 1 for d in data_list)
        ei = torch.cat([d.edge_index for d in data_list], dim=1).to(torch.int64)
        self.esrc_all, self.edst_all = ei[0].contiguous().to(dev), ei[1].contiguous().to(dev)
        self.pos_enc_all = torch.cat([d.pos_enc for d in data_list]).to(torch.int64).to(dev)
        self.pos_index_all = torch.cat([d.pos_index for d in data_list]).to(torch.int64).to(dev)
        self.pos_batch_all = torch.cat([d.pos_batch for d in data_list]).to(torch.int64).to(dev)
        if int(self.pos_index_all.max()) >= N_COLS or int(self.pos_index_all.min()) < 0:
            raise IndexError("pos_index outside the %d-row z_initial table" % N_COLS)

        # ---- sorted views, built once (host-side plumbing on the device; not on the step path) ----
        Nn, Ee, Zz = int(self.h_node_ptr[-1]), int(self.h_edge_ptr[-1]), int(self.h_nnz_ptr[-1])
        gid_e = torch.repeat_interleave(torch.arange(G, device=dev), e.to(dev))
        node_off_e = self.node_ptr[gid_e]
        self.in_ptr_all, self.in_edge_all = _stable_group(self.edst_all + node_off_e, Nn)
        self.out_ptr_all, self.out_edge_all = _stable_group(self.esrc_all + node_off_e, Nn)
        gid_z = torch.repeat_interleave(torch.arange(G, device=dev), z.to(dev))
        edge_glob = self.pos_batch_all + self.edge_ptr[gid_z]
        if Zz and not bool((edge_glob[1:] >= edge_glob[:-1]).all()):
            raise ValueError("pos_batch must be non-decreasing inside every graph")
        self.row_ptr_all = torch.zeros(Ee + 1, dtype=torch.int64, device=dev)
        self.row_ptr_all[1:] = torch.cumsum(torch.bincount(edge_glob, minlength=Ee), 0)
        gc_key = gid_z * N_COLS + self.pos_index_all
        gc_ptr, self.c_perm_all = _stable_group(gc_key, G * N_COLS)
        self.c_rank_all = (torch.arange(Zz, device=dev) - gc_ptr[gc_key[self.c_perm_all]]).to(torch.int32)
        self.col_cnt_all = (gc_ptr[1:] - gc_ptr[:-1]).to(torch.int32).view(G, N_COLS).contiguous()

    def _compute_int_ranges(self):
        """value range of the integer features, per column: batches gathered from this store are known to stay inside it,
        so the embedding encoders need no per-batch range check (a device read-back that drains the stream)"""
        self.int_ranges = {}
        if self.x_is_int and self.x_all.numel():
            xi = self.x_all.long()
            self.int_ranges["x"] = (xi.min(0)[0].tolist(), xi.max(0)[0].tolist())
        ea = self.edge_attr_all
        if ea is not None and not ea.is_floating_point() and ea.numel():
            ea2 = ea.reshape(ea.size(0), -1)
            self.int_ranges["edge_attr"] = (ea2.min(0)[0].tolist(), ea2.max(0)[0].tolist())

    def __len__(self):
        return self.num_graphs

    # ---- on-disk cache: the processed dataset, like the reference's data_*.pt (GraphCountDataset.py:119-120) ----
    _HOST = ("h_node_ptr", "h_edge_ptr", "h_nnz_ptr", "h_y_ptr")
    _META = ("num_graphs", "x_dim", "y_dim", "y_is_vector", "x_is_int", "x_was_1d")

    def save(self, path):
        """torch.save of every tensor of the store (sorted views included) — reload with DeviceGraphStore.load."""
        blob = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in self.__dict__.items()
                if torch.is_tensor(v) or k in self._META}
        torch.save(blob, path)

    # ---- the reference's own processed layout: InMemoryDataset's `(data, slices)` (GraphCountDataset.py:119-120) -------
    def to_data_slices(self):
        """-> (data, slices): per-key concatenation of all graphs (graph-LOCAL node ids in edge_index, graph-local edge ids
        in pos_batch, exactly what InMemoryDataset.collate stores) and the per-key slice offsets, as plain dicts of CPU
        tensors — `torch.save((data, slices), path)` writes a file that loads with weights_only=True and, under PyG, is
        what `Data.from_dict(data)` / the dataset's `slices` expect."""
        x = self.x_all.cpu()
        x = (x.long().view(-1) if self.x_was_1d else x.long()) if self.x_is_int else x
        y = self.y_all.cpu()
        data = dict(x=x, edge_index=torch.stack([self.esrc_all, self.edst_all]).cpu(),
                    y=y.view(-1) if (self.y_is_vector and self.y_dim == 1) else y,
                    pos_enc=self.pos_enc_all.cpu(), pos_index=self.pos_index_all.cpu(), pos_batch=self.pos_batch_all.cpu())
        slices = dict(x=self.h_node_ptr.clone(), edge_index=self.h_edge_ptr.clone(), y=self.h_y_ptr.clone(),
                      pos_enc=self.h_nnz_ptr.clone(), pos_index=self.h_nnz_ptr.clone(), pos_batch=self.h_nnz_ptr.clone())
        if self.edge_attr_all is not None:
            data["edge_attr"], slices["edge_attr"] = self.edge_attr_all.cpu(), self.h_edge_ptr.clone()
        return data, slices

    @classmethod
    def from_data_slices(cls, data, slices, device):
        """Build the store from the reference's `(data, slices)` pair: `data` is a mapping or any object with the keys as
        attributes (a PyG `Data` loaded by the user's own torch_geometric), `slices` maps each key to its G+1 offsets."""
        from .data import Data
        get = (lambda k: data[k] if k in data else None) if isinstance(data, dict) else (lambda k: getattr(data, k, None))
        keys = [k for k in ("x", "edge_index", "edge_attr", "y", "pos_enc", "pos_index", "pos_batch") if get(k) is not None]
        for k in ("x", "edge_index", "y", "pos_enc", "pos_index", "pos_batch"):
            if k not in keys or k not in slices:
                raise KeyError("from_data_slices: key %r missing from data / slices" % k)
        G = int(slices["x"].numel()) - 1
        graphs = []
        for g in range(G):
            kw = {}
            for k in keys:
                t, a, b = get(k), int(slices[k][g]), int(slices[k][g + 1])
                kw[k] = t[:, a:b] if k == "edge_index" else t[a:b]
            graphs.append(Data(**kw))
        return cls(graphs, device)

    @classmethod
    def load(cls, path, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("DeviceGraphStore lives in HBM; there is no CPU fallback")
        blob = torch.load(path, map_location="cpu", weights_only=True)
        self = cls.__new__(cls)
        self.device = dev
        self.edge_attr_all = None                             # optional tensors are absent from the blob when None
        for k, v in blob.items():
            setattr(self, k, v if (k in cls._HOST or k in cls._META) else v.to(dev))
        self._compute_int_ranges()
        return self

    def nbytes(self):
        return sum(v.numel() * v.element_size() for v in self.__dict__.values() if torch.is_tensor(v) and v.is_cuda)

    def collate(self, graph_ids):
        """Batch.from_data_list over the selected graphs, on the device.  graph_ids: 1-D LongTensor/list."""
        # host ids are the fast path: ids and offsets travel through pinned staging buffers with async copies, so the
        # host never waits for the device inside a training loop (device ids cost a D2H synchronisation)
        if torch.is_tensor(graph_ids) and graph_ids.is_cuda:
            ids_d = graph_ids.to(torch.int64).contiguous()
            ids_h = ids_d.cpu()
        else:
            ids_h = torch.as_tensor(graph_ids, dtype=torch.int64).reshape(-1)
            ids_p = torch.empty(ids_h.numel(), dtype=torch.int64, pin_memory=True)
            ids_p.copy_(ids_h)
            ids_d = ids_p.to(self.device, non_blocking=True)
        B = ids_h.numel()
        if B == 0:
            raise ValueError("collate: empty batch")
        if int(ids_h.min()) < 0 or int(ids_h.max()) >= self.num_graphs:
            raise IndexError("collate: graph id out of range")
        # (numpy on the host pointers: the same arithmetic through torch costs ~5x the host time, and the loop is host-bound
        # on some boxes)
        ptrs = self.__dict__.get("_np_ptrs")
        if ptrs is None:
            ptrs = np.stack([t.numpy() for t in (self.h_node_ptr, self.h_edge_ptr, self.h_nnz_ptr, self.h_y_ptr)])
            self.__dict__["_np_ptrs"] = ptrs
        offs = torch.zeros(4, B + 1, dtype=torch.int64, pin_memory=True)
        idn = ids_h.numpy()
        np.cumsum(ptrs[:, idn + 1] - ptrs[:, idn], axis=1, out=offs.numpy()[:, 1:])
        N, E, Z, Y = (int(v) for v in offs.numpy()[:, B])
        offs_d = offs.to(self.device, non_blocking=True)
        dev, i64, i32, f32 = self.device, torch.int64, torch.int32, torch.float32
        # categorical node features (ZINC / OGB) leave the fill kernel as int64; per-edge attribute rows are gathered by it too
        x = torch.empty((N, self.x_dim), dtype=i64 if self.x_is_int else f32, device=dev)
        ea_all = self.edge_attr_all
        ea_row = 1
        for d_ in (ea_all.shape[1:] if ea_all is not None else ()):
            ea_row *= int(d_)
        ea_fused = ea_all is not None and ea_all.is_contiguous() and ea_row > 0 and (ea_all.element_size() * ea_row) % 4 == 0
        edge_attr = torch.empty((E,) + tuple(ea_all.shape[1:]), dtype=ea_all.dtype, device=dev) if ea_fused else None
        graph_ptr = torch.empty(B + 1, dtype=i32, device=dev)
        y = torch.empty((Y, self.y_dim), dtype=f32, device=dev)
        edge_index = torch.empty((2, E), dtype=i64, device=dev)
        batch = torch.empty(N, dtype=i64, device=dev)
        pos_enc = torch.empty(Z, dtype=i64, device=dev)
        pos_index = torch.empty(Z, dtype=i64, device=dev)
        pos_batch = torch.empty(Z, dtype=i64, device=dev)
        # plan buffers: one int32 slab
        sizes = [N + 1, E, E, N + 1, E, E, E + 1, Z, Z, Z, Z, Z, N_COLS + 1, N_COLS, B * N_COLS]
        slab = torch.empty(sum(sizes), dtype=i32, device=dev)
        parts, o = [], 0
        for s_ in sizes:
            parts.append(slab[o:o + s_])
            o += s_
        (in_ptr, in_edge, in_src, out_ptr, out_edge, out_dst, row_ptr, bag_idx, bag_val, col_row, col_val, col_col,
         col_ptr, col_total, col_prefix) = parts
        s = nv.stream()
        nv.call("esc_collate_cols", nv.ptr(self.col_cnt_all), N_COLS, nv.ptr(ids_d), B, nv.ptr(col_prefix),
                nv.ptr(col_total), nv.ptr(col_ptr), s)
        tpl = self.__dict__.get("_args_tpl")
        if tpl is None:                 # the store's own arrays never move: their addresses are filled in once
            t0 = nv.CollateArgs()
            t0.x_dim, t0.y_dim, t0.n_cols = self.x_dim, self.y_dim, N_COLS
            for name in ("node_ptr", "edge_ptr", "nnz_ptr", "y_ptr", "x_all", "y_all", "esrc_all", "edst_all", "pos_enc_all",
                         "pos_index_all", "pos_batch_all", "in_ptr_all", "in_edge_all", "out_ptr_all", "out_edge_all",
                         "row_ptr_all", "c_perm_all", "c_rank_all"):
                setattr(t0, name, getattr(self, name).data_ptr())
            tpl = bytes(t0)
            self.__dict__["_args_tpl"] = tpl
        a = nv.CollateArgs.from_buffer_copy(tpl)
        a.B = B
        a.graph_ids, a.offsets, a.col_ptr, a.col_prefix = ids_d.data_ptr(), offs_d.data_ptr(), col_ptr.data_ptr(), col_prefix.data_ptr()
        a.x, a.y, a.edge_index, a.batch = x.data_ptr(), y.data_ptr(), edge_index.data_ptr(), batch.data_ptr()
        a.pos_enc, a.pos_index, a.pos_batch = pos_enc.data_ptr(), pos_index.data_ptr(), pos_batch.data_ptr()
        base32 = slab.data_ptr()        # the plan arrays are consecutive int32 ranges of one slab
        o = 0
        for name, s_ in zip(("in_ptr", "in_edge", "in_src", "out_ptr", "out_edge", "out_dst", "row_ptr", "bag_idx", "bag_val",
                             "col_row", "col_val", "col_col"), sizes):
            setattr(a, name, base32 + 4 * o)
            o += s_
        if self.x_is_int:
            a.x, a.x_long = None, x.data_ptr()
        if ea_fused:
            a.edge_attr_all, a.edge_attr = ea_all.data_ptr(), edge_attr.data_ptr()
            a.ea_words = ea_all.element_size() * ea_row // 4
        a.graph_ptr = graph_ptr.data_ptr()
        nv.call("esc_collate_fill", ctypes.byref(a), s)
        out = Batch()
        if self.x_is_int:                                    # small categorical ids survive the fp32 round trip exactly
            x = x.view(-1) if self.x_was_1d else x
        out.x, out.edge_index = x, edge_index
        if ea_fused:
            out.edge_attr = edge_attr
        elif self.edge_attr_all is not None:                 # (row size not a multiple of 4 bytes) gather by a device-built index
            cnt = (offs_d[1, 1:] - offs_d[1, :-1])
            src0 = self.edge_ptr[ids_d]
            gather = torch.arange(E, device=dev) + torch.repeat_interleave(src0 - offs_d[1, :-1], cnt, output_size=E)   # (output_size: no read-back)
            out.edge_attr = self.edge_attr_all[gather]
        out.y = y.view(-1) if (self.y_is_vector and self.y_dim == 1) else y
        out.pos_enc, out.pos_index, out.pos_batch, out.batch = pos_enc, pos_index, pos_batch, batch
        plan = BatchPlan(in_ptr=in_ptr, in_edge=in_edge, in_src=in_src, out_ptr=out_ptr, out_edge=out_edge,
                         out_dst=out_dst, row_ptr=row_ptr, bag_idx=bag_idx, bag_val=bag_val, col_ptr=col_ptr,
                         col_row=col_row, col_val=col_val, col_col=col_col, num_nodes=N, num_edges=E, nnz=Z,
                         n_cols=N_COLS)
        plan._keepalive = (slab, offs_d, ids_d)
        plan.graph_ptr, plan.num_graphs = graph_ptr, B       # node range of every graph (readout pooling)
        plan._batch_sig = _sig(batch)
        object.__setattr__(out, "_num_graphs", B)
        batch._esc_seg = {(batch._version, None): plan.graph_ptr, (batch._version, B): plan.graph_ptr}   # pooling ops: no rebuild, no read-back
        # the dataset-wide value range of the integer features, valid for exactly these tensors in their current version
        sig = {k: (out[k].data_ptr(), out[k]._version) for k in ("x", "edge_attr") if out[k] is not None}
        object.__setattr__(out, "_esc_int_ranges", (getattr(self, "int_ranges", None) or {}, sig))
        for k, rng in (getattr(self, "int_ranges", None) or {}).items():     # ... also on the tensors themselves (per-op encoders)
            if out[k] is not None:
                out[k]._esc_known_range = (rng, out[k]._version)
        plan._key = plan_key(out, N_COLS)                    # valid as long as nobody swaps or edits the index tensors
        object.__setattr__(out, "_esc_plan", plan)
        has_attr = self.edge_attr_all is not None

        def lazy_slices():            # Batch.to_data_list bookkeeping (reference batch.py:151-211), only when asked for
            node, edge, nnz, yy = (offs[r].tolist() for r in range(4))
            zero = [0] * B
            slices = dict(x=node, edge_index=edge, y=yy, pos_enc=nnz, pos_index=nnz, pos_batch=nnz)
            shifts = dict(x=zero, edge_index=node[:-1], y=zero, pos_enc=zero, pos_index=zero, pos_batch=edge[:-1])
            if has_attr:
                slices["edge_attr"], shifts["edge_attr"] = edge, zero
            return slices, shifts
        object.__setattr__(out, "_lazy_slices", lazy_slices)
        return out


class DeviceLoader(object):
    """Iterates mini-batches of a DeviceGraphStore like the reference's DataLoader(dataset, batch_size, shuffle)."""

    def __init__(self, store, batch_size=1, shuffle=False, generator=None):
        self.store, self.batch_size, self.shuffle, self.generator = store, batch_size, shuffle, generator

    def __len__(self):
        return (len(self.store) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        G = len(self.store)
        order = torch.randperm(G, generator=self.generator) if self.shuffle else torch.arange(G)
        for i in range(0, G, self.batch_size):
            yield self.store.collate(order[i:i + self.batch_size])

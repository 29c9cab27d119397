"""`Batch.from_data_list`: merge graphs into one disconnected graph with the increment rules of
/root/reference/batch.py:25-149 (a-5 in SURVEY.md §8):

  * `edge_index` (and any `*index*` / `face` key)  -> shifted by the running node count, cat dim -1
  * `pos_batch`                                     -> shifted by `pos_batch.max()+1` (= #edges) (:70-71)
  * `pos_enc`, `pos_index`, `edge_pos`              -> never shifted (:72-73)
  * any other `*batch*` key                         -> PyG default (`max+1` of the shifted item)
  * everything else                                 -> concatenated along dim 0, unshifted
  * `batch[i]` = graph id of node i (:120-123); bool tensors are never shifted (:55)

Unlike the reference (python loop over graphs x keys with one tensor add per item) this builds
every key with ONE concatenation plus ONE vectorised offset add.  The baseline keys of the other
(non-efficient) subgraph pipelines (node_to_subgraph, assignment_index_2, ...) are out of scope.
"""
import torch

from .data import Data

_NEVER_SHIFTED = ("pos_enc", "pos_index", "edge_pos")


def _is_number(v):
    return isinstance(v, (int, float)) and not isinstance(v, bool)


def _repeat(values, counts, device):
    """values[i] repeated counts[i] times.  On the host through numpy: torch.repeat_interleave on CPU tensors takes tens of
    milliseconds per call under a multi-threaded OpenMP runtime (60 ms for 5*10^5 outputs on 8 threads, 0.7 ms in numpy),
    three calls per batch."""
    import numpy as np
    out = torch.from_numpy(np.repeat(values.numpy(), np.asarray(counts, dtype=np.int64)))
    return out if device is None or torch.device(device).type == "cpu" else out.to(device)


class Batch(Data):
    def __init__(self, batch=None, **fields):
        super().__init__(**fields)
        self.batch = batch
        object.__setattr__(self, "_slices", None)
        object.__setattr__(self, "_shifts", None)
        object.__setattr__(self, "_data_class", Data)

    @staticmethod
    def from_data_list(data_list, follow_batch=()):
        if len(data_list) == 0:
            raise ValueError("Batch.from_data_list: empty data_list")
        keys = []
        for d in data_list:
            for k in d.keys:
                if k not in keys:
                    keys.append(k)
        assert "batch" not in keys
        first = data_list[0]
        out = Batch()
        object.__setattr__(out, "_data_class", first.__class__)
        slices, shifts = {}, {}
        node_counts = []
        for d in data_list:
            n = d.num_nodes
            node_counts.append(None if n is None else int(n))

        for key in keys:
            items = [d[key] for d in data_list if key in d]
            owners = [d for d in data_list if key in d]
            probe = items[0]
            if torch.is_tensor(probe):
                dim = first.__cat_dim__(key, probe)
                sizes = [int(t.size(dim)) if t.dim() > 0 else 1 for t in items]
                # running offset BEFORE each graph
                running, offs = 0, []
                for d, t in zip(owners, items):
                    offs.append(running)
                    if t.dtype == torch.bool or key in _NEVER_SHIFTED:
                        inc = 0
                    elif key == "pos_batch":
                        inc = int(t.max()) + 1
                    elif "batch" in key:
                        inc = int(t.max()) + 1 + running          # PyG: __inc__ sees the shifted item
                    else:
                        inc = d.__inc__(key, t)
                        inc = int(inc) if not torch.is_tensor(inc) else int(inc.item())
                    running += inc
                merged = torch.cat([t if t.dim() > 0 else t.view(1) for t in items], dim=dim)
                if any(offs) and merged.dtype != torch.bool:
                    shape = [1] * merged.dim()
                    shape[dim] = -1
                    if merged.device.type == "cpu":      # in place through numpy (torch.cat made a fresh tensor): see _repeat
                        import numpy as np
                        m = merged.numpy()
                        m += np.repeat(np.asarray(offs, dtype=m.dtype), np.asarray(sizes, dtype=np.int64)).reshape(shape)
                    else:
                        off_t = _repeat(torch.tensor(offs, dtype=merged.dtype), sizes, merged.device)
                        merged = merged + off_t.view(shape)
                out[key] = merged
                bounds = [0]
                for s in sizes:
                    bounds.append(bounds[-1] + s)
                slices[key], shifts[key] = bounds, offs
                if key in follow_batch:
                    out["%s_batch" % key] = _repeat(torch.arange(len(items)), sizes, merged.device)
            else:
                out[key] = torch.tensor(items) if _is_number(probe) else list(items)
                slices[key] = list(range(len(items) + 1))
                shifts[key] = [0] * len(items)

        if all(n is not None for n in node_counts):
            dev = None
            for d in data_list:
                for k in d.keys:
                    if torch.is_tensor(d[k]):
                        dev = d[k].device
                        break
                if dev is not None:
                    break
            out.batch = _repeat(torch.arange(len(data_list)), node_counts, dev)
        else:
            out.batch = None
        object.__setattr__(out, "_slices", slices)
        object.__setattr__(out, "_shifts", shifts)
        return out.contiguous()

    def to_data_list(self):
        """Inverse of from_data_list (reference batch.py:151-211)."""
        lazy = getattr(self, "_lazy_slices", None)
        if self._slices is None and lazy is not None:      # device-collated batches derive them on demand
            slices, shifts = lazy()
            object.__setattr__(self, "_slices", slices)
            object.__setattr__(self, "_shifts", shifts)
        if self._slices is None:
            raise RuntimeError("Cannot reconstruct data list from batch because the batch object was "
                               "not created using Batch.from_data_list()")
        n_graphs = max(len(b) for b in self._slices.values()) - 1
        out = []
        for i in range(n_graphs):
            d = self._data_class()
            for key, bounds in self._slices.items():
                if i + 1 >= len(bounds):
                    continue
                v = self[key]
                if torch.is_tensor(v) and v.dim() > 0:
                    dim = d.__cat_dim__(key, v)
                    piece = v.narrow(dim, bounds[i], bounds[i + 1] - bounds[i])
                    if piece.dtype != torch.bool and self._shifts[key][i]:
                        piece = piece - self._shifts[key][i]
                    d[key] = piece
                else:
                    d[key] = v[bounds[i]:bounds[i + 1]] if bounds[i + 1] - bounds[i] != 1 else v[bounds[i]]
            out.append(d)
        return out

    @property
    def num_graphs(self):
        """Number of graphs in the batch (reference batch.py:214-217).  A batch that came from the device collate knows the
        count on the host (no device read-back, which would stall the caller until the stream has drained)."""
        n = self.__dict__.get("_num_graphs")
        return n if n is not None else int(self.batch[-1].item()) + 1

"""`Data`: the attribute container the reference gets from torch_geometric.data.Data, restated
without any PyG dependency (attribute + `in` + `[]` access, `.keys`,
`.num_nodes`, `.to(device)`, collate hooks `__cat_dim__` / `__inc__`).

Used by create_subgraphs (utils_edge_efficient.py:29-32,146-151 in the reference), by
Batch.from_data_list (batch.py:25-149) and by NestedGIN_eff.forward (run_graphcount.py:134-135).
"""
import torch

_POSITIONAL = ("x", "edge_index", "edge_attr", "y", "pos")


class Data(object):
    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, pos=None, **fields):
        object.__setattr__(self, "_store", {})
        object.__setattr__(self, "_num_nodes", None)
        for name, value in zip(_POSITIONAL, (x, edge_index, edge_attr, y, pos)):
            self._store[name] = value
        for name, value in fields.items():
            setattr(self, name, value)

    # ---- field access --------------------------------------------------------------------
    def __getattr__(self, name):
        store = object.__getattribute__(self, "_store")
        if name in store:
            return store[name]
        if name in _POSITIONAL:
            return None
        raise AttributeError("%s has no field %r" % (type(self).__name__, name))

    def __setattr__(self, name, value):
        if name == "num_nodes":
            object.__setattr__(self, "_num_nodes", value)
        elif name.startswith("_") or isinstance(getattr(type(self), name, None), property):
            object.__setattr__(self, name, value)
        else:
            self._store[name] = value
            if name == "batch":                              # a host-cached graph count (device collate) belonged to the old vector
                object.__getattribute__(self, "__dict__").pop("_num_graphs", None)

    def __delattr__(self, name):
        if name in self._store:
            del self._store[name]
        else:
            object.__delattr__(self, name)

    def __getitem__(self, key):
        return self._store.get(key)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def __contains__(self, key):
        return self._store.get(key) is not None

    @property
    def keys(self):
        """Names of the fields that hold a value (None-valued fields are hidden, like PyG)."""
        return [k for k, v in self._store.items() if v is not None]

    def __iter__(self):
        for k in sorted(self.keys):
            yield k, self._store[k]

    def __len__(self):
        return len(self.keys)

    # ---- sizes ---------------------------------------------------------------------------
    @property
    def num_nodes(self):
        if self._num_nodes is not None:
            return self._num_nodes
        for k in ("x", "pos", "batch"):
            v = self._store.get(k)
            if torch.is_tensor(v):
                return v.size(0)
        ei = self._store.get("edge_index")
        if torch.is_tensor(ei) and ei.numel():
            return int(ei.max()) + 1
        return None

    @property
    def num_edges(self):
        ei = self._store.get("edge_index")
        return ei.size(1) if torch.is_tensor(ei) else None

    @property
    def num_features(self):
        x = self._store.get("x")
        return 0 if x is None else (1 if x.dim() == 1 else x.size(1))

    # ---- collate hooks (PyG semantics: *index*/face keys concatenate along the last dim and are
    # shifted by the node count, *batch* keys by max+1, everything else along dim 0 unshifted)
    def __cat_dim__(self, key, value):
        return -1 if ("index" in key or "face" in key) else 0

    def __inc__(self, key, value):
        if "batch" in key:
            return int(value.max()) + 1
        if "index" in key or "face" in key:
            return self.num_nodes
        return 0

    # ---- movement ------------------------------------------------------------------------
    def apply(self, fn):
        for k in self.keys:
            v = self._store[k]
            if torch.is_tensor(v):
                self._store[k] = fn(v)
        return self

    def to(self, device, non_blocking=False):
        d = object.__getattribute__(self, "__dict__")
        plan = d.get("_esc_plan")
        still_valid = False
        if plan is not None:                                 # judged BEFORE the move: a plan that is already stale (edge dropout,
            from .plan import plan_key                       # an edited pos_* tensor) must not be stamped valid by the re-key below
            from .plan import _sig
            still_valid = getattr(plan, "_key", None) == plan_key(self, plan.n_cols)
            batch_valid = plan.graph_ptr is not None and plan._batch_sig == _sig(self._store.get("batch"))
        ranges = d.get("_esc_int_ranges")
        trusted = ()
        if ranges is not None:                               # which integer features are still the tensors the store signed
            trusted = [k for k, (addr, ver) in ranges[1].items() if torch.is_tensor(self._store.get(k)) and
                       (self._store[k].data_ptr(), self._store[k]._version) == (addr, ver)]
        out = self.apply(lambda t: t.to(device, non_blocking=non_blocking))
        if plan is not None:
            if still_valid:                                  # the plan follows its tensors; re-key it on their new identity
                plan.to(device)
                plan._key = plan_key(self, plan.n_cols)
                if batch_valid:
                    plan._batch_sig = _sig(self._store.get("batch"))
                else:
                    plan.graph_ptr, plan.num_graphs, plan._batch_sig = None, None, None
            else:
                d.pop("_esc_plan", None)                     # plan_of() rebuilds it from the tensors as they are now
        if ranges is not None:                               # (address, version) signatures of the integer features: follow the move
            d["_esc_int_ranges"] = (ranges[0], {k: (self._store[k].data_ptr(), self._store[k]._version) for k in trusted})
        return out

    def contiguous(self):
        return self.apply(lambda t: t.contiguous())

    def clone(self):
        out = type(self)()
        for k in self.keys:
            v = self._store[k]
            out._store[k] = v.clone() if torch.is_tensor(v) else v
        object.__setattr__(out, "_num_nodes", self._num_nodes)
        return out

    def __repr__(self):
        parts = []
        for k in self.keys:
            v = self._store[k]
            parts.append("%s=%s" % (k, list(v.shape) if torch.is_tensor(v) else v))
        return "%s(%s)" % (type(self).__name__, ", ".join(parts))

"""Shim `torch_geometric.data.Data` with the PyG-2.0.x behaviours the reference
relies on: positional (x, edge_index, edge_attr, y, pos, **kw) constructor,
attribute / item / `in` access, `keys` hiding None and dunder entries,
`num_nodes` inference, `__cat_dim__` / `__inc__` collate rules, contiguous()."""
import re
import torch

_CORE = ("x", "edge_index", "edge_attr", "y", "pos")


class Data(object):
    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, pos=None, **kwargs):
        for name, val in zip(_CORE, (x, edge_index, edge_attr, y, pos)):
            object.__setattr__(self, name, val)
        for name, val in kwargs.items():
            if name == "num_nodes":
                object.__setattr__(self, "__num_nodes__", val)
            else:
                object.__setattr__(self, name, val)

    # -- access ---------------------------------------------------------
    def __getattr__(self, name):
        # only reached when normal lookup fails
        if name in _CORE:
            return None
        raise AttributeError(name)

    def __getitem__(self, key):
        return getattr(self, key, None)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def __contains__(self, key):
        return key in self.keys

    @property
    def keys(self):
        out = []
        for k, v in self.__dict__.items():
            if v is None or (k.startswith("__") and k.endswith("__")):
                continue
            out.append(k)
        return out

    def __iter__(self):
        for k in sorted(self.keys):
            yield k, self[k]

    # -- sizes ----------------------------------------------------------
    @property
    def num_nodes(self):
        explicit = self.__dict__.get("__num_nodes__")
        if explicit is not None:
            return explicit
        for k in ("x", "pos", "batch"):
            v = self.__dict__.get(k)
            if torch.is_tensor(v):
                return v.size(0)
        ei = self.__dict__.get("edge_index")
        if torch.is_tensor(ei) and ei.numel() > 0:
            return int(ei.max()) + 1
        return None

    @num_nodes.setter
    def num_nodes(self, value):
        object.__setattr__(self, "__num_nodes__", value)

    @property
    def num_edges(self):
        ei = self.__dict__.get("edge_index")
        return None if ei is None else ei.size(1)

    # -- collate rules --------------------------------------------------
    def __cat_dim__(self, key, value):
        return -1 if re.search("(index|face)", key) else 0

    def __inc__(self, key, value):
        if "batch" in key:
            return int(value.max()) + 1
        if re.search("(index|face)", key):
            return self.num_nodes
        return 0

    def contiguous(self):
        for k in self.keys:
            v = self[k]
            if torch.is_tensor(v):
                self[k] = v.contiguous()
        return self

    def to(self, device):
        for k in self.keys:
            v = self[k]
            if torch.is_tensor(v):
                self[k] = v.to(device)
        return self

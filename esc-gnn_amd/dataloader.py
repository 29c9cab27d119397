"""`DataLoader`: torch DataLoader whose collate merges `Data` objects with Batch.from_data_list
(/root/reference/dataloader.py:11-48).

The reference's training loop is `for data in train_loader: data = data.to(device); ...`
(run_graphcount.py:453-455,487-505): every mini-batch is collated by a python loop over graphs x keys on the
host (~12 ms per 128-graph batch of the counting data) and copied to the device tensor by tensor.  With a HIP
device present this loader keeps that loop UNCHANGED and fast: on the first `iter()` it pins the dataset into a
`DeviceGraphStore` (the reference's `(data, slices)` layout, resident in HBM) and yields batches collated on the
device by ONE gather kernel (csrc/collate.hip) — the same tensors, bit for bit, as `Batch.from_data_list` of
the same graphs moved with `.to(device)`, in the same order as torch's own sampler draws them (`data.to(device)`
in the caller's loop is then a no-op).

The device path is taken only for what it reproduces exactly: `Data` items that carry the ESC keys (x, edge_index,
y, pos_enc, pos_index, pos_batch, optionally edge_attr) and nothing else, no `follow_batch`, no worker processes, the
default sampler / collate.  Anything else — and `device=None` / a CPU-only box — is the host collate of the
reference.  (The host path is the reference's loader, not a fallback of the HIP kernels: the model itself refuses CPU
tensors.)  A dataset that is edited after the first iteration must be re-pinned with `loader.repin()`.
"""
import collections.abc as abc

import torch.utils.data

from .batch import Batch
from .data import Data

_STORE_KEYS = ("x", "edge_index", "y", "pos_enc", "pos_index", "pos_batch")


def _merge(samples, follow_batch):
    head = samples[0]
    if isinstance(head, Data):
        return Batch.from_data_list(samples, follow_batch)
    if isinstance(head, float):
        return torch.tensor(samples, dtype=torch.float)
    if isinstance(head, int):
        return torch.tensor(samples)
    if isinstance(head, (str, bytes)):
        return samples
    if isinstance(head, abc.Mapping):
        return {k: _merge([s[k] for s in samples], follow_batch) for k in head}
    if isinstance(head, tuple) and hasattr(head, "_fields"):
        return type(head)(*(_merge(list(col), follow_batch) for col in zip(*samples)))
    if isinstance(head, abc.Sequence):
        return [_merge(list(col), follow_batch) for col in zip(*samples)]
    raise TypeError("DataLoader found invalid type: {}".format(type(head)))


def _storable(item):
    """a graph the device store reproduces exactly: the ESC keys (+ edge_attr), all tensors, nothing else"""
    if not isinstance(item, Data):
        return False
    keys = set(item.keys)
    if not set(_STORE_KEYS) <= keys or not keys <= set(_STORE_KEYS + ("edge_attr",)):
        return False
    return all(torch.is_tensor(item[k]) for k in keys) and item.x.dim() >= 1 and item.x.size(0) > 0


class DataLoader(torch.utils.data.DataLoader):
    def __init__(self, dataset, batch_size=1, shuffle=False, follow_batch=(), device="auto", **kwargs):
        """device: "auto" (default) pins the dataset on the current HIP device when there is one, a torch device / string
        pins it there, None keeps the reference's host collate."""
        follow = tuple(follow_batch)
        self._esc_plain = not follow and not kwargs.get("num_workers") and \
            all(kwargs.get(k) is None for k in ("collate_fn", "batch_sampler", "sampler"))
        super().__init__(dataset, batch_size, shuffle,
                         collate_fn=lambda samples: _merge(samples, follow), **kwargs)
        self.__dict__["_esc_device"] = device
        self.__dict__["_esc_store"] = None          # None: not decided yet; False: host path; else the DeviceGraphStore

    def repin(self):
        """forget the pinned copy (the dataset was edited): the next iter() pins it again"""
        self.__dict__["_esc_store"] = None

    def _pinned(self):
        store = self.__dict__.get("_esc_store")
        if store is not None:
            return store
        store = False
        dev = self.__dict__.get("_esc_device")
        if self._esc_plain and dev is not None and torch.cuda.is_available() and self.batch_sampler is not None:
            dev = torch.device("cuda", torch.cuda.current_device()) if dev == "auto" else torch.device(dev)
            if dev.type == "cuda":
                try:
                    n = len(self.dataset)
                    items = [self.dataset[i] for i in range(n)]
                except TypeError:
                    items = None                   # iterable-style dataset: host path
                if items and all(_storable(g) for g in items):
                    from .store import DeviceGraphStore
                    store = DeviceGraphStore(items, dev)
        self.__dict__["_esc_store"] = store
        return store

    def __iter__(self):
        store = self._pinned()
        if store is False:
            return super().__iter__()
        return self._device_batches(store)

    def _device_batches(self, store):
        # the random stream is consumed exactly as torch's own iterator consumes it (one base-seed draw when the iterator
        # is made, then the sampler's), so a seeded run visits the same graphs in the same order on either path
        torch.empty((), dtype=torch.int64).random_(generator=self.generator)
        for ids in self.batch_sampler:
            yield store.collate(ids)

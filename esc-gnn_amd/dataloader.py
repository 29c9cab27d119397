"""`DataLoader`: torch DataLoader whose collate merges `Data` objects with Batch.from_data_list
(/root/reference/dataloader.py:11-48)."""
import collections.abc as abc

import torch.utils.data

from .batch import Batch
from .data import Data


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


class DataLoader(torch.utils.data.DataLoader):
    def __init__(self, dataset, batch_size=1, shuffle=False, follow_batch=(), **kwargs):
        follow = tuple(follow_batch)
        super().__init__(dataset, batch_size, shuffle,
                         collate_fn=lambda samples: _merge(samples, follow), **kwargs)

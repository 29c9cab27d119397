"""Pieces shared by the training drivers (run_zinc / run_ogb_mol): device + process-group set-up, seeding,
result directory bookkeeping and graph-sharded batch iteration (rank r takes a contiguous slice of
every global batch; the only collective of a step is the weighted all-reduce of the flat gradient bucket)."""
import os
import random
import shutil
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

from .parallel import shard_slice


class Context(object):
    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if not torch.cuda.is_available():
            raise RuntimeError("needs a HIP device (the hot path has no CPU fallback)")
        local %= torch.cuda.device_count()
        torch.cuda.set_device(local)
        self.device = torch.device("cuda", local)
        if self.world > 1 and not dist.is_initialized():
            backend = os.environ.get("ESC_DIST_BACKEND", "nccl")      # gloo only to rehearse N>1 on one GPU
            dist.init_process_group(backend, **({"device_id": self.device} if backend == "nccl" else {}))

    def say(self, *a, **kw):
        if self.rank == 0:
            print(*a, **kw)

    def all_reduce(self, t):
        if self.world > 1:
            dist.all_reduce(t)
        return t

    def close(self):
        if self.world > 1 and dist.is_initialized():
            dist.destroy_process_group()


def seed_everything(seed):
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    random.seed(seed)
    np.random.seed(seed)


def open_result_dir(ctx, res_dir, sources):
    """results/<...> with the driver sources and the command line saved next to the logs (reference drivers do the
    same, e.g. run_zinc.py:100-113)."""
    cmd_input = "python " + " ".join(sys.argv) + "\n"
    if ctx.rank == 0:
        print("Results will be saved in " + res_dir)
        os.makedirs(res_dir, exist_ok=True)
        here = os.path.dirname(os.path.abspath(__file__))
        for f in sources:
            shutil.copy(os.path.join(here, f), res_dir)
        with open(os.path.join(res_dir, "cmd_input.txt"), "a") as fh:
            fh.write(cmd_input)
        print("Command line input: " + cmd_input + " is saved.")
    return cmd_input


def default_appendix(appendix):
    return appendix if appendix != "" else "_" + time.strftime("%Y%m%d%H%M%S")


def sharded_batches(store, batch_size, ctx, shuffle, generator=None):
    """Global batches of `batch_size` graphs in loader order; this rank collates its contiguous share."""
    G = len(store)
    if shuffle and ctx.world > 1 and generator is None:
        # the global CPU RNG would give every rank the same permutation only while all ranks consume it identically;
        # any rank-dependent draw would silently shard DIFFERENT orders (graphs duplicated / dropped, no error anywhere)
        raise ValueError("sharded_batches(shuffle=True) on %d ranks needs a dedicated, identically seeded torch.Generator" % ctx.world)
    order = torch.randperm(G, generator=generator) if shuffle else torch.arange(G)
    for i in range(0, G, batch_size):
        ids = order[i:i + batch_size]
        if ids.numel() < ctx.world:
            # a remainder smaller than the rank count would leave some ranks without data while the others enter the
            # step's collectives: training drops it (every rank alike), evaluation hands it to rank 0
            if shuffle or ctx.rank != 0:
                continue
            yield store.collate(ids), ids.numel()
            continue
        lo, hi = shard_slice(ids.numel(), ctx.rank, ctx.world)
        yield store.collate(ids[lo:hi]), ids.numel()


_prefetch_streams = {}


def prefetched(batches, device, warm=None):
    """Iterates `batches` (a generator that collates device batches: sharded_batches, DeviceLoader) ONE ITEM AHEAD on a
    side HIP stream: batch i+1 is collated — and `warm(item)` builds whatever per-batch index plans the step engine will
    ask for — while the caller trains on batch i.  This is the role of the reference's DataLoader worker processes
    (run_ogb_mol.py:229-234, num_workers) with the dataset resident in HBM: the ~60 small gather / counting-sort launches
    of a molecule batch leave the step's critical path.
    Ordering: the side stream first waits for everything the caller's stream has been given so far (all of step i-1), so
    the caching allocator may hand it the blocks of batches that are already dropped; the caller's stream waits for the
    side stream's event before it touches the batch.  Tensors live in the side stream's pool but are only ever reused
    behind such a wait."""
    device = torch.device(device)
    if device.type != "cuda":
        for item in batches:
            if warm is not None:
                warm(item)
            yield item
        return
    side = _prefetch_streams.get(device)
    if side is None:
        side = _prefetch_streams[device] = torch.cuda.Stream(device)
    it = iter(batches)

    def issue():
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            try:
                item = next(it)
            except StopIteration:
                return None
            if warm is not None:
                warm(item)
            ready = torch.cuda.Event()
            ready.record(side)
        return item, ready

    ahead = issue()
    while ahead is not None:
        item, ready = ahead
        torch.cuda.current_stream(device).wait_event(ready)
        ahead = issue()                                    # queued BEFORE the caller enqueues step i: overlaps it
        yield item

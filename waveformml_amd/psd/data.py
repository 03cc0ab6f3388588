"""Batch assembly for the PSD path: mirror of the reference's ``collate_fn``
(src/engineering/PSDDataModule.py:10-20) plus a synthetic in-memory dataset with the item structure
of ``HDF5Dataset.__getitem__`` (src/datasets/HDF5Dataset.py:186-217: ONE item = one file's event
range, ``[[coords, feats], labels]`` with event ids starting at 0).
"""
import numpy as np
import queue as _queue
import time as _time

import torch
from torch.utils.data import DataLoader, Dataset

from . import synthetic


def collate_fn(batch):
    """Concatenate per-item COO chunks; event ids of item i>0 are shifted by the number of events
    before it -- IN PLACE on column 2, like the reference (which hard-codes column 2 even for the
    4-column 3-D layout, SURVEY.md 7 "reference quirks"; ``event_column`` below lifts that for callers
    that need the 3-D batch column, see collate_fn_3d)."""
    return _collate(batch, 2)


def collate_fn_3d(batch):
    """The same collate with the event id in column 3 (x, y, t, evt), which is what a 3-D
    ``SPConvNet`` reads its batch size from (reference src/models/SPConvNet.py:63)."""
    return _collate(batch, 3)


def _collate(batch, event_column):
    offset = 0
    for i, b in enumerate(batch):
        if i > 0:
            b[0][0][:, event_column] += offset
        offset += b[1].size()[0]
    coords = torch.cat([b[0][0] for b in batch])
    if isinstance(batch[0][0][1], list):       # additional per-row fields
        feats = [torch.cat([b[0][1][i] for b in batch]) for i in range(len(batch[0][0][1]))]
    else:
        feats = torch.cat([b[0][1] for b in batch])
    return [coords, feats], torch.cat([b[1] for b in batch])


class SyntheticPulseDataset(Dataset):
    """``n_items`` chunks of ``events_per_item`` synthetic events each (a chunk stands for one HDF5 file's
    event range); deterministic in (seed, rank, item)."""

    def __init__(self, n_items, events_per_item, n_samples, n_type=3, layout="3d", seed=1234, rank=0,
                 use_half=False):
        self.n_items, self.events_per_item = n_items, events_per_item
        self.n_samples, self.n_type, self.layout = n_samples, n_type, layout
        self.seed, self.rank = seed, rank
        self.valtype = torch.float16 if use_half else torch.float32

    def __len__(self):
        return self.n_items

    def __getitem__(self, index):
        c, f, y = synthetic.generate(self.events_per_item, self.n_samples, self.n_type,
                                     seed=self.seed + 7919 * index, rank=self.rank, layout=self.layout)
        return [torch.from_numpy(c), torch.from_numpy(f).type(self.valtype)], torch.from_numpy(y)


def rank_sampler(dataset, shuffle, seed=0):
    """Under torch.distributed with more than one rank: a DistributedSampler that gives this rank its 1/N share of the
    ITEMS (an item = one file's event range, so ranks read disjoint file ranges) -- what Lightning's DDP plugin swaps
    into the reference's loaders (src/utils/util.py:228-239).  None in a single process.  The Trainer calls
    ``sampler.set_epoch(epoch)`` so that every epoch is shuffled differently but identically on all ranks."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() <= 1:
        return None
    from torch.utils.data.distributed import DistributedSampler
    return DistributedSampler(dataset, num_replicas=dist.get_world_size(), rank=dist.get_rank(), shuffle=bool(shuffle),
                              seed=seed)


class _SharedRing(object):
    """A ring of message slots in ONE shared-memory block created before the DataLoader forks its workers: workers write
    their batches straight into a free slot and send only (slot, layout) through the queue; the trainer process page-locks
    the block once (hipHostRegister) and copies to the GPU from the slot itself.  No shared-memory segment per message,
    no pinning thread, no copy in the trainer process: the stock route (a fresh segment per message + DataLoader's
    ``pin_memory`` thread) tops out at ~385 k events/s whatever the number of workers, the ring at ~670 k with the same
    16 CPUs (tools/exp/loader_scaling.py; profiles/r03_soak_from_files.json)."""

    def __init__(self, slots, slot_bytes):
        import multiprocessing
        self.slots, self.slot_bytes = int(slots), int(slot_bytes)
        self.buf = torch.empty((self.slots, self.slot_bytes), dtype=torch.uint8).share_memory_()
        self.free = multiprocessing.get_context("fork").Queue()
        for i in range(self.slots):
            self.free.put(i)
        self.registered = False

    def register(self):
        """Page-lock the block in THIS process (the trainer): ``.to(device, non_blocking=True)`` from a slot is then a true
        asynchronous copy.  Needs an initialised GPU; a no-op without one."""
        if self.registered or not torch.cuda.is_available():
            return self.registered
        try:
            rc = torch.cuda.cudart().cudaHostRegister(self.buf.data_ptr(), self.buf.numel(), 0)
            self.registered = int(rc) == 0 if not isinstance(rc, tuple) else int(rc[0]) == 0
        except Exception:          # noqa: BLE001  -- unregistered memory still works (synchronous staging copy)
            self.registered = False
        return self.registered

    def close(self):
        if self.registered:
            try:
                torch.cuda.cudart().cudaHostUnregister(self.buf.data_ptr())
            except Exception:      # noqa: BLE001
                pass
            self.registered = False


class RingBatch(list):
    """``[[coords, feats], labels]`` whose tensors are views of a ring slot; ``release()`` hands the slot back (called by
    DevicePrefetcher once the host->device copy has completed; otherwise the loader reclaims it a few messages later)."""
    token = None

    def release(self):
        if self.token is not None:
            self.token.done()


class _SlotToken(object):
    def __init__(self, ring, slot, batches):
        self.ring, self.slot, self.left = ring, slot, batches
        # events of copies that read FROM the slot (DevicePrefetcher records one per staged batch): the slot must not
        # go back to the workers before they have completed
        self.guards = []

    def done(self):
        self.left -= 1
        if self.left == 0 and self.slot is not None:
            self.ring.free.put(self.slot)
            self.slot = None

    def force(self):
        """Reclaim by age (PackedLoader.HOLD).  A consumer that copies asynchronously out of the slot registered its
        copy events in ``guards``: they are WAITED for first -- an issued copy whose completion nobody has waited on
        may still be reading the slot a worker is about to overwrite."""
        if self.slot is not None:
            for ev in self.guards:
                ev.synchronize()
            self.guards = []
            self.ring.free.put(self.slot)
            self.slot = None


class _PackedCollate(object):
    """Collate in the worker, then pack the batch's tensors into ONE byte buffer (each at a 16-byte boundary): a batch
    then crosses the worker -> trainer process boundary as one shared-memory segment instead of three (or more), and is
    pinned with one copy.  ``unpack`` rebuilds the tensors as views of the buffer.

    ``group`` > 1: one message carries ``group`` consecutive batches (the worker is handed group x items and collates
    them ``items_per_batch`` at a time): the per-message cost of the DataLoader machinery (queue hand-over, pinning
    thread; ~1 ms, i.e. a ceiling of ~1000 batches/s per trainer process whatever the number of workers) is paid once
    per group."""

    def __init__(self, collate, items_per_batch=None, group=1, ring=None):
        self.collate, self.items_per_batch, self.group, self.ring = collate, items_per_batch, int(group), ring

    def __call__(self, items):
        if self.group <= 1 or not self.items_per_batch:
            chunks = [items]
        else:
            chunks = [items[i:i + self.items_per_batch] for i in range(0, len(items), self.items_per_batch)]
        batches, off = [], 0
        for chunk in chunks:
            (coords, feats), labels = self.collate(chunk)
            feats_list = feats if isinstance(feats, list) else [feats]
            tensors = [coords] + feats_list + [labels]
            meta = []
            for t in tensors:
                nbytes = t.numel() * t.element_size()
                meta.append((off, nbytes, t.dtype, tuple(t.shape)))
                off += (nbytes + 15) // 16 * 16
            batches.append((tensors, meta, isinstance(feats, list)))
        slot = None
        if self.ring is not None and off <= self.ring.slot_bytes:
            try:
                slot = self.ring.free.get(timeout=20.0)  # blocks while every slot is in flight: back-pressure
            except Exception:      # noqa: BLE001  -- queue.Empty: slots lost to an abandoned epoch; the ordinary route
                slot = None
        buf = self.ring.buf[slot] if slot is not None else torch.empty(max(off, 16), dtype=torch.uint8)
        for tensors, meta, _ in batches:
            for t, (o, nbytes, _, _) in zip(tensors, meta):
                if nbytes:
                    buf[o:o + nbytes].view(t.dtype).view(t.shape).copy_(t)
        return (("ring", slot) if slot is not None else buf), [(meta, is_list) for _t, meta, is_list in batches]

    @staticmethod
    def unpack(packed, ring=None):
        """The batches of one message, in order (RingBatch objects when the message lives in a ring slot)."""
        buf, metas = packed
        token = None
        if isinstance(buf, tuple):
            token = _SlotToken(ring, buf[1], len(metas))
            buf = ring.buf[buf[1]]
        out = []
        for meta, feats_is_list in metas:
            ts = [buf[o:o + nbytes].view(dtype).view(shape) if nbytes else torch.empty(shape, dtype=dtype)
                  for (o, nbytes, dtype, shape) in meta]
            feats = ts[1:-1] if feats_is_list else ts[1]
            if token is None:
                out.append(([ts[0], feats], ts[-1]))
            else:
                b = RingBatch(([ts[0], feats], ts[-1]))
                b.token = token
                out.append(b)
        return out


class PackedLoader(object):
    """A ``DataLoader`` whose worker processes hand over each batch as one buffer (see _PackedCollate); iterating yields
    the same ``[[coords, feats], labels]`` batches as the plain loader.  ``group``: batches per message (default
    $WFS_LOADER_GROUP or 1).  Measured on the MI355X box's host, 255-event batches of the bench's events out of 3 files
    each (tools/soak_from_files.py -> profiles/r03_soak_from_files.json)."""

    HOLD = 3            # messages a consumer that never calls RingBatch.release() may keep alive
    HOLD_MAX = 8        # slots are provisioned for a HOLD up to this (a deeper consumer falls back on the copy guards)

    def __init__(self, dataset, collate_fn, group=None, ring_slot_mb=None, **loader_kwargs):
        import os
        self.group = int(group if group is not None else os.environ.get("WFS_LOADER_GROUP", "1"))
        self.items_per_batch = int(loader_kwargs.get("batch_size", 1) or 1)
        if self.group > 1:
            loader_kwargs["batch_size"] = self.items_per_batch * self.group
        # ring of message slots shared with the workers (needs worker PROCESSES forked from this one): the slot size is
        # $WFS_LOADER_SLOT_MB per batch of the message (default 4: twice a 255-event batch of the bench's events); a
        # message that does not fit takes the ordinary route.  WFS_LOADER_RING=0 turns it off.
        nw = int(loader_kwargs.get("num_workers", 0) or 0)
        self.ring = None
        if nw > 0 and os.environ.get("WFS_LOADER_RING", "1") != "0" and loader_kwargs.get("multiprocessing_context") is None:
            mb = float(ring_slot_mb if ring_slot_mb is not None else os.environ.get("WFS_LOADER_SLOT_MB", "4"))
            # + HOLD_MAX: a consumer may raise HOLD up to that (DevicePrefetcher: its staging depth + 1) without the
            # workers running out of slots
            slots = nw * int(loader_kwargs.get("prefetch_factor", 2) or 2) + self.HOLD_MAX + 3
            try:
                self.ring = _SharedRing(slots, int(mb * (1 << 20)) * max(self.group, 1))
                loader_kwargs["pin_memory"] = False       # the ring is page-locked once instead
            except Exception:      # noqa: BLE001  -- e.g. /dev/shm too small: the ordinary route
                self.ring = None
        self.loader = DataLoader(dataset, collate_fn=_PackedCollate(collate_fn, self.items_per_batch, self.group, self.ring),
                                 **loader_kwargs)
        self.dataset = dataset
        self.drop_last = bool(loader_kwargs.get("drop_last", False))

    @property
    def sampler(self):
        return self.loader.sampler

    def __len__(self):
        n = len(self.loader.sampler) if self.loader.sampler is not None else len(self.dataset)
        return n // self.items_per_batch if self.drop_last else (n + self.items_per_batch - 1) // self.items_per_batch

    def __iter__(self):
        if self.ring is not None:
            self.ring.register()
        held = []                           # slot tokens of the last messages, oldest first
        try:
            for packed in self.loader:
                batches = _PackedCollate.unpack(packed, self.ring)
                if batches and isinstance(batches[0], RingBatch):
                    held.append(batches[0].token)
                    while len(held) > self.HOLD:
                        held.pop(0).force()         # a consumer that kept it this long has copied what it needs
                for batch in batches:
                    yield batch
        finally:
            for t in held:
                t.force()


def make_loader(dataset, items_per_batch, num_workers=0, shuffle=False, pin_memory=True):
    fn = collate_fn_3d if getattr(dataset, "layout", "2d") == "3d" else collate_fn
    sampler = rank_sampler(dataset, shuffle)
    return DataLoader(dataset, batch_size=items_per_batch, shuffle=shuffle and sampler is None, sampler=sampler,
                      num_workers=num_workers, collate_fn=fn, pin_memory=pin_memory)


def to_device(batch, device, feature_dtype=None, non_blocking=True):
    (c, f), y = batch
    c = c.to(device, non_blocking=non_blocking)
    f = f.to(device, non_blocking=non_blocking)
    if feature_dtype is not None:
        f = f.to(feature_dtype)
    return [c, f], y.to(device, non_blocking=non_blocking)


class DevicePrefetcher(object):
    """Pinned double-buffering of the loader -> HBM hand-over: while step i computes, batch i+1's
    host->device copies run on a dedicated copy stream; the consumer's stream waits on the copy's event, never on the
    host.  Replaces the reference's synchronous ``.to(self.device)`` inside ``_concat_range``
    (src/datasets/HDF5Dataset.py:250-300), which a forked DataLoader worker cannot do on a HIP device anyway."""

    def __init__(self, loader, device, feature_dtype=None, depth=2, on_stage=None, on_exhausted=None, threaded=None):
        self.loader, self.device, self.feature_dtype, self.depth = loader, torch.device(device), feature_dtype, depth
        import os
        self.threaded = (os.environ.get("WFS_PREFETCH_THREAD", "0") != "0") if threaded is None else bool(threaded)
        self.wait_seconds = 0.0
        self.copy_stream = torch.cuda.Stream(device=self.device)
        # this consumer releases ring slots itself (after each copy): the loader's reclaim-by-age must sit behind the
        # staging depth, so that it only ever fires for batches this queue has already let go
        if hasattr(loader, "HOLD"):
            loader.HOLD = max(int(loader.HOLD), int(depth) + 1)
        # on_stage(rows, labels) for every batch as it is staged (``depth`` batches before it is yielded), on_exhausted()
        # when the loader has no more: the multi-rank Trainer agrees on batch shapes ahead of time through these
        self.on_stage, self.on_exhausted = on_stage, on_exhausted

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch, notify=True):
        (c, f), y = batch
        if notify and self.on_stage is not None:
            self.on_stage(int(c.shape[0]), int(y.shape[0]))
        ring = isinstance(batch, RingBatch)            # views of a page-locked ring slot: copy straight from it
        host = [t if (ring or t.is_pinned()) else t.pin_memory() for t in (c, f, y)]
        with torch.cuda.stream(self.copy_stream):
            dev = [t.to(self.device, non_blocking=True) for t in host]
            if self.feature_dtype is not None:
                dev[1] = dev[1].to(self.feature_dtype)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        if ring and batch.token is not None:
            batch.token.guards.append(done)            # reclaim-by-age waits for this copy (see _SlotToken.force)
        # (host buffers kept alive until the copy has been waited on, device tensors, copy event, rows, labels)
        return (batch if ring else host), dev, done, int(c.shape[0]), int(y.shape[0])

    def _producer(self, out, stop):
        """Background thread: pulls batches off the loader and stages them (worker-message unpacking, the host -> device
        copies and the dtype cast all happen here); hands (staged | exception | None at the end) to the consumer."""
        try:
            if self.device.type == "cuda":
                torch.cuda.set_device(self.device)
            for batch in self.loader:
                if stop.is_set():
                    break
                item = self._stage(batch, notify=False)
                while not stop.is_set():
                    try:
                        out.put(item, timeout=0.05)
                        break
                    except _queue.Full:
                        continue
            out.put(None)
        except BaseException as e:          # noqa: BLE001  -- delivered to the consumer, which re-raises it
            out.put(e)

    def _staged_batches(self):
        """Staged batches in loader order; with ``threaded`` (WFS_PREFETCH_THREAD=1; off by default) they are produced by a
        background thread, so that the training thread's step is the hand-over launch + the graph replay and nothing
        else.  Measured from files at the bench's density (tools/soak_from_files.py 600 85 8 16, round 4): 466-469 k
        events/s per epoch with the thread, 462-468 k without -- the training thread then WAITS 45 us per step for the
        next staged batch instead of staging it itself: neutral, so the simpler single-threaded form stays the default."""
        if not self.threaded:
            for batch in self.loader:
                yield self._stage(batch, notify=False)
            return
        import threading
        out, stop = _queue.Queue(maxsize=max(2, self.depth)), threading.Event()
        th = threading.Thread(target=self._producer, args=(out, stop), name="wfs-prefetch", daemon=True)
        th.start()
        try:
            while True:
                item = out.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            stop.set()
            while th.is_alive():             # unblock a producer waiting on a full queue, then let it finish
                try:
                    out.get_nowait()
                except _queue.Empty:
                    pass
                th.join(timeout=0.05)

    def __iter__(self):
        queue = []
        it = self._staged_batches()
        more = True
        while True:
            while more and len(queue) < self.depth:
                try:
                    t_wait = _time.perf_counter()
                    item = next(it)
                    self.wait_seconds += _time.perf_counter() - t_wait      # time the consumer spent waiting for a batch
                    if self.on_stage is not None:       # in consumption order, `depth` batches ahead, on THIS thread
                        self.on_stage(item[3], item[4])
                    queue.append(item[:3])
                except StopIteration:
                    more = False
                    if self.on_exhausted is not None:
                        self.on_exhausted()
            if not queue:
                return
            # ring slots go back as soon as their copy has completed (a query, no wait): a deep queue (the multi-rank
            # Trainer stages several batches ahead) must not sit on slots the loader would otherwise reclaim by age
            for j in range(1, len(queue)):
                h, d, e = queue[j]
                if isinstance(h, RingBatch) and e.query():
                    h.release()
                    queue[j] = (None, d, e)
            host, dev, done = queue.pop(0)
            cur = torch.cuda.current_stream(self.device)
            if not done.query():
                # the copy was issued `depth` batches ago and has normally completed: only then does the consumer's stream
                # need to wait for it at all (a cross-stream wait is a barrier packet in front of the step: ~10 us of
                # launch latency per step in the from-files soak)
                cur.wait_event(done)
            if isinstance(host, RingBatch):
                done.synchronize()        # issued a batch ago: long finished; the slot may now be overwritten
                host.release()
            for t in dev:
                t.record_stream(cur)      # allocated on the copy stream, consumed on the compute stream
            yield [dev[0], dev[1]], dev[2]

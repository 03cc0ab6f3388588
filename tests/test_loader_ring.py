"""CPU checks of the worker -> trainer hand-over through the shared ring of message slots (psd/data._SharedRing,
PackedLoader): the batches must be exactly the plain DataLoader's, over several epochs of persistent workers, with and
without explicit release of the slots, and a message that does not fit a slot must take the ordinary route."""
import os

import torch
from torch.utils.data import DataLoader

from waveformml_amd.psd.data import PackedLoader, RingBatch, SyntheticPulseDataset, collate_fn_3d


def _plain(ds, items):
    return list(DataLoader(ds, batch_size=items, collate_fn=collate_fn_3d, num_workers=0))


def _same(a, b):
    (ca, fa), ya = a
    (cb, fb), yb = b
    return torch.equal(ca, cb) and torch.equal(fa, fb) and torch.equal(ya, yb)


def test_ring_batches_equal_the_plain_loader_over_epochs():
    ds = SyntheticPulseDataset(n_items=14, events_per_item=5, n_samples=32, seed=3)
    want = _plain(ds, 2)
    for group in (1, 3):
        loader = PackedLoader(ds, collate_fn_3d, group=group, batch_size=2, num_workers=2, persistent_workers=True,
                              prefetch_factor=2)
        assert loader.ring is not None and len(loader) == len(want)
        for _epoch in range(3):
            got = []
            for b in loader:
                assert isinstance(b, RingBatch)
                got.append(([b[0][0].clone(), b[0][1].clone()], b[1].clone()))     # never released: reclaimed by HOLD
            assert len(got) == len(want) and all(_same(g, w) for g, w in zip(got, want))
        # every slot is back once the epoch's messages have been reclaimed
        import queue
        free = 0
        try:
            while True:
                loader.ring.free.get(timeout=1.0)           # puts travel through a feeder thread: wait for them
                free += 1
        except queue.Empty:
            pass
        assert free == loader.ring.slots
        del loader


def test_explicit_release_and_oversized_messages():
    ds = SyntheticPulseDataset(n_items=8, events_per_item=4, n_samples=32, seed=5)
    want = _plain(ds, 2)
    loader = PackedLoader(ds, collate_fn_3d, group=1, batch_size=2, num_workers=2, prefetch_factor=2)
    got = []
    for b in loader:
        got.append(([b[0][0].clone(), b[0][1].clone()], b[1].clone()))
        b.release()                                   # what DevicePrefetcher does after the host -> device copy
    assert all(_same(g, w) for g, w in zip(got, want))
    # slots of 1 KiB: nothing fits, every message takes the ordinary shared-memory route, same batches
    small = PackedLoader(ds, collate_fn_3d, group=1, ring_slot_mb=1.0 / 1024, batch_size=2, num_workers=2, prefetch_factor=2)
    got = list(small)
    assert not any(isinstance(b, RingBatch) for b in got)
    assert all(_same(g, w) for g, w in zip(got, want))


def test_ring_can_be_switched_off(monkeypatch):
    monkeypatch.setenv("WFS_LOADER_RING", "0")
    ds = SyntheticPulseDataset(n_items=4, events_per_item=3, n_samples=32, seed=1)
    loader = PackedLoader(ds, collate_fn_3d, batch_size=2, num_workers=1)
    assert loader.ring is None and len(list(loader)) == 2


def test_reclaim_by_age_waits_for_the_consumers_copies():
    """a consumer deeper than HOLD whose copies out of a slot have been ISSUED but not waited for: the
    loader's reclaim-by-age must wait for those copies (the guards registered on the slot's token) before the slot goes
    back to the workers -- and a DevicePrefetcher-style consumer raises HOLD behind its own depth in the first place."""
    ds = SyntheticPulseDataset(n_items=16, events_per_item=4, n_samples=32, seed=7)
    want = _plain(ds, 2)
    loader = PackedLoader(ds, collate_fn_3d, group=1, batch_size=2, num_workers=2, prefetch_factor=2)
    assert loader.ring is not None

    class FakeCopy(object):                 # stands for the event of a non-blocking host -> device copy
        log = []

        def __init__(self, batch):
            self.batch, self.snapshot, self.waited = batch, None, False

        def synchronize(self):              # the "copy" completes only when somebody waits for it: it reads the slot NOW
            if not self.waited:
                self.snapshot = ([self.batch[0][0].clone(), self.batch[0][1].clone()], self.batch[1].clone())
                self.waited = True
                FakeCopy.log.append(self)

    copies = []
    loader.HOLD = 1                          # far smaller than the consumer's depth: every reclaim is "too early"
    for b in loader:
        ev = FakeCopy(b)
        b.token.guards.append(ev)            # what DevicePrefetcher._stage does
        copies.append(ev)                    # ... and it never releases, never waits: the delayed consumer
    for ev in copies:
        ev.synchronize()
    assert len(copies) == len(want)
    assert all(_same(c.snapshot, w) for c, w in zip(copies, want))      # no copy ever read an overwritten slot
    assert sum(1 for c in FakeCopy.log) == len(want)
    # the prefetcher's side of the bargain: HOLD follows the staging depth
    from waveformml_amd.psd import data as _data

    class _NoStream(object):
        def __init__(self, *a, **k):
            pass

    real = torch.cuda.Stream
    torch.cuda.Stream = _NoStream
    try:
        loader2 = PackedLoader(ds, collate_fn_3d, group=1, batch_size=2, num_workers=2, prefetch_factor=2)
        _data.DevicePrefetcher(loader2, "cpu", depth=6)
        assert loader2.HOLD >= 7 and loader2.ring.slots >= 2 * 2 + loader2.HOLD + 3
    finally:
        torch.cuda.Stream = real

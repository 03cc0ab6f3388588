"""HIP-graph capture of the whole PSD training step.

At the reference's batch sizes a step is ~100 small kernels (a 256-event batch is ~10^5 voxels, a layer
moves ~10 MB), so eager execution is bound by Python/launch latency and by the two host read-backs a
data-dependent output size normally needs.  libwfsparse's device-side row counts (include/wfsparse.h)
remove the read-backs: every tensor has a fixed CAPACITY of rows, the valid count lives in device memory.
With that the step -- rulebook builds, forward, backward, gradient packing, optimizer -- has static
shapes and is captured once into a HIP graph (torch.cuda.CUDAGraph) and replayed per batch.

    step = GraphedTrainStep(module, optimizer, reducer, example_batch)
    loss = step(batch)            # copies the batch into the static buffers, replays the graph
    step.check()                  # occasionally: raises if a capacity was exceeded (results invalid)
"""
import contextlib
import gc
import os
import sys

import torch
import torch.distributed as dist


@contextlib.contextmanager
def _no_gc():
    """The cyclic garbage collector off for the length of a capture.  Objects of an EARLIER captured step (its CUDAGraph
    with RCCL nodes, collective work objects, tensors of its private pool) sit in reference cycles -- hook closures <->
    reducer <-> parameters -- so only the cyclic collector frees them, at whatever allocation it happens to run on; run
    inside a later capture, their destructors free device memory and graph resources on a capturing thread (the abort
    seen once in the one-rank RCCL test).  torch.cuda.graph() collects once BEFORE the capture starts; this
    keeps the collector from running again until the capture has ended."""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def _round_up(v, m):
    return int((int(v) + m - 1) // m * m)


class GraphedTrainStep(object):
    def __init__(self, module, optimizer, reducer, example_batch, headroom=None, granule=None, warmup=2, min_rows=0):
        # headroom: row capacities = headroom x the example batch's row counts.  Capacity is not free (the
        # register-resident BatchNorm kernels and the rulebook grids are sized by it: 1.25 -> 1.12 measured -4 % per
        # step).  The voxel count of an E-event PSD batch varies by sigma ~ 0.5 / sqrt(E) of its mean (3.1 % at 256
        # events: psd/synthetic, 40 batches), so the default is 6 sigma above the example batch: 1 + 3 / sqrt(E), i.e.
        # 1.19 at 256 events.  A batch beyond a capacity is detected (check()), never silently cut.
        (coords, feats), labels = example_batch
        headroom, granule = self._headroom_granule(int(coords.shape[0]), int(labels.shape[0]), headroom, granule)
        assert coords.is_cuda and feats.is_cuda and labels.is_cuda
        self.module, self.optimizer, self.reducer = module, optimizer, reducer
        dev = coords.device
        self.n_cap = max(_round_up(headroom * coords.shape[0], granule), _round_up(min_rows, granule))
        self.coords = torch.zeros((self.n_cap, coords.shape[1]), dtype=coords.dtype, device=dev)
        self.feats = torch.zeros((self.n_cap, feats.shape[1]), dtype=feats.dtype, device=dev)
        # targets: one per EVENT (LitPSD: a fixed number per batch), or one per ROW (per-segment modules with
        # ``per_row_targets``: LitSegClassifier) -- then the buffer has the rows' capacity and everything beyond the valid
        # rows holds the criterion's ignore_index, so that the padding rows neither count nor receive a gradient
        self.per_row = bool(getattr(module, "per_row_targets", False)) and labels.shape[0] == coords.shape[0]
        if self.per_row:
            self.ignore_index = int(getattr(getattr(module, "criterion", None), "ignore_index", -100))
            self.labels = torch.full((self.n_cap,) + tuple(labels.shape[1:]), self.ignore_index, dtype=labels.dtype, device=dev)
        else:
            self.labels = torch.zeros_like(labels)
        self.n_valid = torch.zeros((1,), dtype=torch.int64, device=dev)
        # the coordinates in the reference's batch-first column order, written by the same hand-over launch; the net
        # takes them instead of permuting when it is given exactly self.coords (psd/net.py)
        net = getattr(module, "model", None)
        if self.per_row and net is not None and hasattr(net, "batch_size_hint"):
            # the nets read the number of events off the last coordinate row (a host read-back of a padding row here).
            # All the site tables need is an upper bound, and every event has at least one row: the row capacity is one
            net.batch_size_hint = self.n_cap
        perm = getattr(net, "permute_tensor", None)
        self._perm = [int(v) for v in perm.tolist()] if perm is not None else None
        self.indices = torch.zeros_like(self.coords) if self._perm is not None else None
        # the event offsets the event-local rulebook build starts from (spconv.ops.EVENT_LOCAL) are written by the
        # hand-over launch as well: one graph node fewer at the head of the step
        from ..spconv import ops as _ops
        self.n_events = int(self.n_cap if self.per_row else labels.shape[0])
        self.events = None
        if (_ops.EVENT_LOCAL and self.indices is not None and 1 <= self.n_events <= _ops.EVENT_LOCAL_MAX_BATCH
                and hasattr(net, "batch_events")):
            from .. import _lib as _l
            self.events = torch.zeros((int(_l.load().wfs_event_offsets_ints(self.n_events)),), dtype=torch.int32,
                                      device=dev)
            net.batch_events = (self.coords, self.events)
        self.world = reducer.world
        # Gradient exchange of a captured step (world > 1).  With RCCL the collectives are captured INSIDE the graph:
        # the reducer's hooks stay armed, so during the captured backward each bucket (reverse layer order) is packed
        # and its all-reduce issued the moment its last gradient exists -- torch's process group launches it on its own
        # stream, i.e. a fork off the captured stream right after that bucket's last dW, running beside the rest of
        # the backward -- and reducer.finish() joins them in front of the optimizer launch, which is then back inside
        # the graph too.  Other backends (gloo: the one-GPU rehearsal; host copies cannot be captured) exchange the
        # packed buffer after the replay, followed by an eager optimizer launch.
        exchange = bool(getattr(reducer, "exchange", self.world > 1))
        self.in_graph_exchange = (exchange and dist.is_available() and dist.is_initialized()
                                  and dist.get_backend(reducer.group) == "nccl"
                                  and os.environ.get("WFS_GRAPH_EXCHANGE", "1") != "0")
        if self.in_graph_exchange:
            # decided ONCE, on a scratch graph holding a single small all-reduce, before anything expensive is captured
            # on this communicator -- and agreed between the ranks (no in-process retry of a failed capture
            # that contained collectives)
            self.in_graph_exchange = _collective_capture_works(reducer.group, dev)
        self.exchange_after = exchange and not self.in_graph_exchange
        self.in_graph_optimizer = not self.exchange_after
        self._convs = [m for m in module.modules()
                       if hasattr(m, "subm") and hasattr(m, "conv1x1") and not m.subm and not m.conv1x1 and not m.inverse]
        # Stream discipline.  (1) Autograd's AccumulateGrad nodes remember the stream they were created on, and a
        # node born on the legacy default stream cannot take part in a capture.  (2) On ROCm 7.2 ANY eager work on
        # the legacy default (null) stream between two replays makes the next replay hang.  So the runner moves the
        # calling thread onto an ordinary stream for good -- calibration, warm-up, capture, replays and whatever
        # eager work the caller does afterwards (loss.item(), logging, the next batch's copies) all run there.
        if torch.cuda.current_stream(dev) == torch.cuda.default_stream(dev):
            st = torch.cuda.Stream(dev)
            st.wait_stream(torch.cuda.default_stream(dev))
            torch.cuda.set_stream(st)
        self.stream = torch.cuda.current_stream(dev)
        # ---- calibration: one ordinary (exact-size) step tells how many rows each strided layer produces
        reducer.reset()
        loss = module.training_step(([coords, feats], labels), 0)
        loss.backward()
        reducer.finish()
        optimizer.step()
        del loss
        for m in self._convs:
            # in proportion to the ROW capacity, which min_rows (a re-capture's floor, the largest batch among the ranks)
            # may have raised above headroom x this batch: a batch that fits the rows must fit the strided layers too
            own = _round_up(headroom * coords.shape[0], granule)
            factor = headroom if self.n_cap <= own else self.n_cap / max(1.0, float(coords.shape[0]))
            # a tight row capacity (a re-capture sized from the row counts seen, Trainer._recapture_rows) says nothing
            # about the outputs-per-input ratio of the strided layers, which varies by a few per cent from batch to
            # batch: they keep 6 % on top (the conv kernels share out the VALID tiles, so this room is nearly free)
            slack = 1.06 if headroom < 1.05 else 1.0
            m.out_capacity = _round_up(slack * factor * m.last_rulebook.M, granule)
        if self.exchange_after:
            reducer.remove()              # no collectives inside the graph: gradients are exchanged after the replay
        # ---- warm-up in device-count mode, then capture
        if self.indices is not None:
            net.batch_first_indices = (self.coords, self.indices)
        self._load(example_batch)
        failed, err = 0, None
        try:
            self._warm_and_capture(warmup)
        except Exception as e:            # noqa: BLE001
            if not self.in_graph_exchange:
                raise
            failed, err = 1, e
        if self.in_graph_exchange and self.world > 1:
            # a collective library that cannot be captured: ranks must not assume they all failed alike -- they AGREE
            # (MAX over a CPU-side flag on a gloo side group, which needs nothing from the communicator in doubt)
            failed = _agree_max(failed, reducer.group)
        if failed:
            print("[waveformml_amd] in-graph gradient exchange could not be captured on some rank (%s); exchanging "
                  "after the replay" % (("%s: %s" % (type(err).__name__, err)) if err is not None else "another rank"),
                  file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            self.in_graph_exchange, self.exchange_after, self.in_graph_optimizer = False, True, False
            reducer.remove()
            self._warm_and_capture(warmup)
        self._overflow = [m.last_rulebook.overflow for m in self._convs if m.last_rulebook.overflow is not None]
        self._event_flags = _event_flags(module)
        # the builds only ever SET these (sticky); allocated inside the capture they start undefined: cleared here and
        # after every read, so that check() sees a failure of ANY replay since the last check()
        _clear_flags(self._overflow, self._event_flags)

    @staticmethod
    def _headroom_granule(rows, labels, headroom=None, granule=None):
        if headroom is None and os.environ.get("WFS_CAPTURE_HEADROOM"):
            headroom = float(os.environ["WFS_CAPTURE_HEADROOM"])          # experiments: tools/exp (capacity is not free)
        if headroom is None:
            headroom = max(1.1, 1.0 + 3.0 / max(1.0, float(labels)) ** 0.5)
        if granule is None:
            # capacities are rounded up to a granule; padded rows cost real work in the wide-channel (GEMM route) layers
            # of the 2-D nets, whose batches are a few hundred rows, so the granule follows the batch
            granule = 512 if rows >= 32768 else (256 if rows >= 2048 else 64)
        return headroom, granule

    @classmethod
    def capacity_for(cls, rows, labels):
        """The row capacity a step captured on a batch of ``rows`` rows and ``labels`` labels gets: ranks that capture
        together pass the LARGEST row count among them as ``min_rows`` so that they all hold the same capacity (their
        replay-or-ordinary-step decisions are derived from agreed counts and must come out alike)."""
        headroom, granule = cls._headroom_granule(int(rows), int(labels))
        return _round_up(headroom * int(rows), granule)

    def _warm_and_capture(self, warmup):
        for _ in range(warmup):
            self._body()
            if self.exchange_after:
                self._after()
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: the process group's watchdog thread queries events while this thread captures
        mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
        with _no_gc(), torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode=mode):
            self.loss = self._body()

    def close(self, remove_hooks=True):
        """Orderly end of a captured step, BEFORE its process group is destroyed or another step is captured on the same
        reducer: the reducer's hooks come off (``remove_hooks=False``: the reducer goes on to serve the next capture), the
        device drains, the graph (whose nodes include the RCCL kernels and the communicator's stream when the exchange
        was captured) is destroyed and its memory pool released, the references the net holds to the static buffers are
        dropped.  Idempotent; the step cannot be called afterwards."""
        if getattr(self, "graph", None) is None:
            return
        if remove_hooks:
            self.reducer.remove()
        torch.cuda.synchronize(self.coords.device)
        self.graph.reset()
        self.graph = None
        self.loss = None
        net = getattr(self.module, "model", None)
        if net is not None:
            if getattr(net, "batch_events", None) is not None and net.batch_events[0] is self.coords:
                net.batch_events = None
            if getattr(net, "batch_first_indices", None) is not None and net.batch_first_indices[0] is self.coords:
                net.batch_first_indices = None
        torch.cuda.synchronize(self.coords.device)

    def _static_batch(self):
        return ([self.coords, self.feats, self.n_valid], self.labels)

    def _body(self):
        self.reducer.reset()
        from ..spconv import functional as Fsp
        loss = self.module.training_step(self._static_batch(), 0)
        # weight-gradient slab reductions of the whole backward pass run as ONE launch at pack time (pack_all flushes)
        Fsp.defer_dw(self.reducer.flat_param is not None)
        try:
            if loss.dtype == torch.float32 and loss.dim() == 0:
                torch.autograd.backward(loss, grad_tensors=Fsp.unit_loss_grad(loss.device))   # no ones-fill, no multiply
            else:
                loss.backward()
            if self.in_graph_exchange:
                self.reducer.finish()      # packs what no hook packed, joins the bucket all-reduces, averages
            else:
                self.reducer.pack_all()
        finally:
            Fsp.defer_dw(False)
        if self.in_graph_optimizer:
            self.optimizer.step()
        return loss.detach()

    def _after(self):
        self.reducer.exchange_packed()
        self.optimizer.step()

    def eager_step(self, batch):
        """The same step on an exact-size batch without the graph (a batch that does not fit the capture), with the
        SAME sequence of collectives as a replay, so that ranks replaying and ranks stepping eagerly could even mix."""
        self.reducer.reset()
        net = getattr(self.module, "model", None)
        hint = getattr(net, "batch_size_hint", None)
        if self.per_row and hint is not None:
            # the captured step's hint is its ROW capacity as a bound on the events; this batch is exactly the one that
            # exceeds it: let the net read the event count off the batch, as it does outside captured steps
            net.batch_size_hint = None
        try:
            loss = self.module.training_step(batch, 0)
        finally:
            if self.per_row and hint is not None:
                net.batch_size_hint = hint
        loss.backward()
        if self.exchange_after:
            self.reducer.pack_all()
            self.reducer.exchange_packed()
        else:
            self.reducer.finish()
        self.optimizer.step()
        return loss.detach()

    def fits(self, batch):
        (coords, _f), labels = batch
        if self.per_row:
            return coords.shape[0] <= self.n_cap and labels.shape[0] == coords.shape[0]
        return coords.shape[0] <= self.n_cap and tuple(labels.shape) == tuple(self.labels.shape)

    def fits_counts(self, rows_max, labels_min, labels_max, rows_ne_labels):
        """``fits`` from the counts the ranks agreed on (ShapeAgreement): the largest row count, the smallest / largest
        label count and whether any rank's batch has a label count other than its row count."""
        if self.per_row:
            return rows_max <= self.n_cap and not rows_ne_labels
        return rows_max <= self.n_cap and labels_min == labels_max == int(self.labels.shape[0])

    def _load(self, batch):
        (coords, feats), labels = batch
        n = coords.shape[0]
        if n > self.n_cap:
            raise RuntimeError("batch has %d voxels, the captured step holds %d" % (n, self.n_cap))
        if getattr(self, "per_row", False):
            self.coords[:n].copy_(coords, non_blocking=True)
            self.feats[:n].copy_(feats, non_blocking=True)
            self.labels.fill_(self.ignore_index)
            self.labels[:n].copy_(labels, non_blocking=True)
            self.n_valid.fill_(n)
            if self.indices is not None:
                self.indices[:n].copy_(coords[:, self._perm], non_blocking=True)
            self._event_offsets()
            return
        if (coords.is_cuda and coords.dtype == torch.int32 and coords.is_contiguous() and feats.is_cuda
                and feats.is_contiguous() and feats.dtype == self.feats.dtype and labels.is_cuda
                and labels.dtype == torch.int64 and labels.is_contiguous() and labels.shape == self.labels.shape):
            # one launch: coordinates (as they are and batch-first), features, labels, row count
            from .. import _lib
            lib = _lib.load()
            perm = _lib.i32_array(self._perm) if self._perm is not None else None
            ev = self.events if (self.events is not None and perm is not None) else None
            _lib.check(lib.wfs_load_batch(_lib.ptr(coords), n, coords.shape[1], perm, _lib.ptr(self.coords),
                                          _lib.ptr(self.indices), _lib.ptr(feats), _lib.ptr(self.feats),
                                          feats.numel() * feats.element_size(), _lib.ptr(labels), _lib.ptr(self.labels),
                                          labels.numel(), _lib.ptr(self.n_valid), _lib.ptr(ev), self.n_events,
                                          _lib.stream_ptr()))
            if ev is None:
                self._event_offsets()
            return
        self.coords[:n].copy_(coords, non_blocking=True)
        self.feats[:n].copy_(feats, non_blocking=True)
        self.labels.copy_(labels, non_blocking=True)
        self.n_valid.fill_(n)
        if self.indices is not None:
            self.indices[:n].copy_(coords[:, self._perm], non_blocking=True)
        self._event_offsets()

    def _event_offsets(self):
        """The event offsets of the loaded batch when the hand-over launch did not write them (its own launch)."""
        if self.events is None:
            return
        from .. import _lib
        _lib.check(_lib.load().wfs_event_offsets(_lib.ptr(self.indices), self.indices.shape[0], self.indices.shape[1] - 1,
                                                 self.n_events, _lib.ptr(self.n_valid), _lib.ptr(self.events),
                                                 _lib.stream_ptr()))

    def __call__(self, batch):
        if torch.cuda.current_stream(self.coords.device) == torch.cuda.default_stream(self.coords.device):
            torch.cuda.set_stream(self.stream)         # see "Stream discipline" in __init__
        self._load(batch)
        if hasattr(self.optimizer, "sync_hyperparameters"):
            self.optimizer.sync_hyperparameters()      # a scheduler's new lr reaches the captured update
        self.graph.replay()
        if self.exchange_after:
            self._after()
        return self.loss

    def check(self):
        """Synchronises; raises if any strided layer produced more rows than its capacity in the last step -- on ANY
        rank: the flag is all-reduced first, so every rank raises together instead of one rank leaving its peers
        blocked in the next collective."""
        flag = torch.zeros((), dtype=torch.int32, device=self.coords.device)
        if self._overflow:
            flag = torch.stack([o.reshape(()) for o in self._overflow]).any().to(torch.int32)
        evf = torch.zeros((), dtype=torch.int32, device=self.coords.device)
        for f in getattr(self, "_event_flags", ()):
            evf = evf | f[: 2 * (f.numel() // 3)].any().to(torch.int32)
        if self.world > 1 and dist.is_available() and dist.is_initialized():
            dist.all_reduce(evf, op=dist.ReduceOp.MAX, group=self.reducer.group)
        if self.world > 1 and dist.is_available() and dist.is_initialized():
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.reducer.group)
        bad_events, overflow = bool(evf.item()), bool(flag.item())
        _clear_flags(self._overflow, getattr(self, "_event_flags", ()))
        if bad_events:
            raise RuntimeError("a batch was not grouped by event (or an event exceeded the LDS tables of the event-local "
                               "rulebook build, or held duplicate coordinates); set WFS_EVENT_LOCAL=0 and re-capture")
        if overflow:
            raise RuntimeError("a sparse conv output exceeded its captured capacity; re-capture with more headroom")


def _collective_capture_works(group, dev):
    """Can this process group's all-reduce be captured into a HIP graph and replayed?  Probed with one 8-float
    all-reduce on a scratch graph (after an eager one: a communicator must not be created inside a capture); every rank
    gets the same answer (MAX of the failure flags over the gloo side group)."""
    ok = True
    try:
        st = torch.cuda.current_stream(dev)
        if st == torch.cuda.default_stream(dev):
            st = torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            t = torch.ones((8,), dtype=torch.float32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            torch.cuda.synchronize(dev)
            t.fill_(1.0)
            g = torch.cuda.CUDAGraph()
            with _no_gc(), torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            g.replay()
            torch.cuda.synchronize(dev)
            ok = abs(float(t[0].item()) - float(dist.get_world_size(group))) < 1e-3
            g.reset()                       # destroyed here, with the device idle -- not whenever the collector gets to it
            torch.cuda.synchronize(dev)
            del g
    except Exception as e:         # noqa: BLE001  -- whatever the library raises: the answer is "no"
        print("[waveformml_amd] a captured all-reduce is not available here (%s: %s): gradients are exchanged after the "
              "replay" % (type(e).__name__, e), file=sys.stderr, flush=True)
        ok = False
        try:
            torch.cuda.synchronize(dev)
        except Exception:          # noqa: BLE001
            pass
    return not _agree_max(0 if ok else 1, group)


_GLOO_SIDE = {}


def _gloo_side(group):
    """The gloo group over the ranks of ``group`` (created once, collectively, at first use)."""
    key = id(group)
    if key not in _GLOO_SIDE:
        ranks = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
        _GLOO_SIDE[key] = dist.new_group(ranks=ranks, backend="gloo")
    return _GLOO_SIDE[key]


def _agree_max(flag, group):
    """MAX of a host-side integer over the ranks of ``group``, over a gloo group of the same ranks (created once): used
    to agree on things that must not depend on the device communicator being healthy."""
    if not (dist.is_available() and dist.is_initialized()):
        return flag
    t = torch.tensor([int(flag)], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_gloo_side(group))
    return int(t.item())


class ShapeAgreement(object):
    """What every rank must know about a batch before it chooses between a replay and an ordinary step (ranks must
    choose alike: the two issue their collectives differently): the largest row count over the ranks, the smallest and
    largest label count, whether any rank's labels are not one per row.  Host integers, MAX-reduced over the gloo side
    group, ``block`` batches per all-reduce, issued ASYNCHRONOUSLY when the prefetcher stages the batches -- several steps
    before they are consumed -- so the host never waits for it.  It replaces one BLOCKING gloo all-reduce per step,
    which costs 0.26 ms of host time with 2 ranks, 0.49 with 4, 0.85 with 8 (8 CPU cores) against a 0.6-ms step.
    Every rank must stage the same number of batches per epoch (a DistributedSampler guarantees it)."""
    FIELDS = 4

    def __init__(self, group, block=4):
        self.group, self.block = _gloo_side(group), max(1, int(block))
        self.local, self.ready = [], []

    def stage(self, rows, labels):
        self.local.append((int(rows), int(labels), -int(labels), 1 if int(rows) != int(labels) else 0))
        if len(self.local) >= self.block:
            self.flush()

    def flush(self):
        if not self.local:
            return
        t = torch.tensor(self.local, dtype=torch.int64).reshape(-1)
        n, self.local = len(self.local), []
        work = dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group, async_op=True)
        self.ready.extend((work, t, i) for i in range(n))

    def next(self):
        """(rows_max, labels_min, labels_max, rows_ne_labels) of the next staged batch, in staging order."""
        if not self.ready:
            self.flush()                    # fewer batches staged ahead than a block holds (the same on every rank)
        work, t, i = self.ready.pop(0)
        work.wait()
        r = t[self.FIELDS * i: self.FIELDS * (i + 1)].tolist()
        return int(r[0]), -int(r[2]), int(r[1]), bool(r[3])


def _clear_flags(*groups):
    for tensors in groups:
        for t in tensors:
            t.zero_()


def _event_flags(module):
    """Failure flags of the event-local rulebook builds of the module's conv layers (spconv.ops.EVENT_LOCAL)."""
    out, seen = [], set()
    for m in module.modules():
        rb = getattr(m, "last_rulebook", None)
        f = getattr(rb, "event_flags", None) if rb is not None else None
        if f is not None and f.data_ptr() not in seen:
            seen.add(f.data_ptr())
            out.append(f)
    return out


class GraphedEvalStep(object):
    """Forward-only counterpart for the inference loops (validation, ``test_step``, the occlusion study): the eval-mode
    forward of ``module.model`` -- rulebook builds included -- captured once over capacity-padded buffers and replayed per
    batch.  ``logits = step(batch)`` returns the STATIC logits tensor (consume it before the next call).

    ``sweep=True`` additionally captures a forward that REUSES the loaded batch's rulebooks (spconv.ops.reuse_rulebooks:
    geometry is a function of the coordinates only), for evaluating one batch many times with different features --
    the reference's occlusion study zeroes one feature column per pass (scripts/RunOcclusionStudy.py ->
    Evaluate.py --occlude -> LitPSD.test_step, src/engineering/LitPSD.py:133-135):

        step = GraphedEvalStep(module, batch, sweep=True)
        base = step(batch).clone()                     # rulebooks + forward
        for idx in columns:
            logits = step.rerun(occlude_index=idx)     # forward only, column idx of the batch's features zeroed
    """

    def __init__(self, module, example_batch, headroom=None, granule=None, sweep=False):
        from ..spconv import ops
        (coords, feats), labels = example_batch
        assert coords.is_cuda and feats.is_cuda
        self.module = module
        dev = coords.device
        if headroom is None:
            headroom = max(1.1, 1.0 + 3.0 / max(1.0, float(labels.shape[0])) ** 0.5)
        if granule is None:
            n0 = int(coords.shape[0])
            granule = 512 if n0 >= 32768 else (256 if n0 >= 2048 else 64)
        self.n_cap = _round_up(headroom * coords.shape[0], granule)
        self.coords = torch.zeros((self.n_cap, coords.shape[1]), dtype=coords.dtype, device=dev)
        self.feats = torch.zeros((self.n_cap, feats.shape[1]), dtype=feats.dtype, device=dev)
        self.feats_loaded = torch.zeros_like(self.feats) if sweep else None     # the batch's own features (sweeps)
        self.labels = torch.zeros_like(labels)
        self.n_valid = torch.zeros((1,), dtype=torch.int64, device=dev)
        net = getattr(module, "model", None)
        perm = getattr(net, "permute_tensor", None)
        self._perm = [int(v) for v in perm.tolist()] if perm is not None else None
        self.indices = torch.zeros_like(self.coords) if self._perm is not None else None
        self._convs = [m for m in module.modules()
                       if hasattr(m, "subm") and hasattr(m, "conv1x1") and not m.subm and not m.conv1x1 and not m.inverse]
        if torch.cuda.current_stream(dev) == torch.cuda.default_stream(dev):     # see GraphedTrainStep "Stream discipline"
            st = torch.cuda.Stream(dev)
            st.wait_stream(torch.cuda.default_stream(dev))
            torch.cuda.set_stream(st)
        self.stream = torch.cuda.current_stream(dev)
        self.n_events = int(labels.shape[0])
        was_training = module.training
        module.eval()
        self._reuse = None
        try:
            with torch.no_grad():
                # calibration: an ordinary exact-size forward tells how many rows each strided layer produces
                if hasattr(net, "batch_size_hint"):
                    net.batch_size_hint = self.n_events
                net([coords, feats])
                for m in self._convs:
                    m.out_capacity = _round_up(headroom * m.last_rulebook.M, granule)
                if self.indices is not None:
                    net.batch_first_indices = (self.coords, self.indices)
                self._load(example_batch)
                if sweep:
                    self._reuse = ops.reuse_rulebooks()          # stays open for the life of this runner
                    self._reuse.__enter__()
                for _ in range(2):
                    self._forward()
                torch.cuda.synchronize()
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, stream=self.stream):
                    self.logits = self._forward()
                self._overflow = [m.last_rulebook.overflow for m in self._convs if m.last_rulebook.overflow is not None]
                _clear_flags(self._overflow)            # sticky flags allocated inside the capture (see GraphedTrainStep)
                self.graph_fwd = None
                if sweep:
                    # second capture inside the same reuse context: every rulebook build is a cache hit (same static
                    # index buffers, same geometry) -> a graph of the forward kernels alone
                    before = ops.BUILD_COUNT
                    self.graph_fwd = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.graph_fwd, stream=self.stream):
                        self.logits_fwd = self._forward()
                    assert ops.BUILD_COUNT == before, "the forward-only capture rebuilt a rulebook"
        finally:
            module.train(was_training)

    def _forward(self):
        net = self.module.model
        if hasattr(net, "batch_size_hint"):
            net.batch_size_hint = self.n_events
        return net([self.coords, self.feats, self.n_valid]).float()

    _load = GraphedTrainStep._load
    _event_offsets = GraphedTrainStep._event_offsets
    events = None                 # the forward-only graphs compute their event offsets inside the graph

    def fits(self, batch):
        (coords, _f), labels = batch
        return coords.shape[0] <= self.n_cap and tuple(labels.shape) == tuple(self.labels.shape)

    def __call__(self, batch, occlude_index=None):
        if torch.cuda.current_stream(self.coords.device) == torch.cuda.default_stream(self.coords.device):
            torch.cuda.set_stream(self.stream)
        self._load(batch)
        if self.feats_loaded is not None:
            self.feats_loaded.copy_(self.feats)
        if occlude_index:                          # falsy for index 0, exactly as the reference (LitPSD.py:134)
            self.feats[:, occlude_index] = 0
        self.graph.replay()
        return self.logits

    def rerun(self, occlude_index=None):
        """The loaded batch again through the forward-only graph, with one feature column zeroed (None / 0: none)."""
        if self.graph_fwd is None:
            raise RuntimeError("GraphedEvalStep(..., sweep=True) captures the forward-only graph that rerun() replays")
        self.feats.copy_(self.feats_loaded)
        if occlude_index:
            self.feats[:, occlude_index] = 0
        self.graph_fwd.replay()
        return self.logits_fwd

    def check(self):
        if self._overflow and bool(torch.stack([o.reshape(()) for o in self._overflow]).any().item()):
            _clear_flags(self._overflow)
            raise RuntimeError("a sparse conv output exceeded its captured capacity; re-capture with more headroom")

    def close(self):
        if self._reuse is not None:
            self._reuse.__exit__(None, None, None)
            self._reuse = None

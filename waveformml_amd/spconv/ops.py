"""Rulebook front door: counterpart of ``spconv.ops`` (spconv 1.2.1).

``get_indice_pairs`` keeps spconv's signature and return value (bit-identical to its CPU algorithm,
A.3).  The modules in this package use ``build_rulebook`` instead, which returns a :class:`Rulebook`
holding the gather tables libwfsparse's compute kernels consume (include/wfsparse.h) and
materialises spconv's ``indice_pairs`` encoding only when somebody asks for it.
"""
import ctypes

import os

import numpy as np
import torch

from .. import _lib


# When True, index rows are taken to be in range and distinct (what the reference's datasets deliver and
# what spconv silently assumes): SubM rulebook builds then never synchronise with the host and regular
# convs skip their duplicate scan.  Default False = every build validates the indices it is given.
ASSUME_VALID_UNIQUE_INDICES = False


# Concurrency switches (both need a caller that follows the stated contract, so both default to off):
#   PREFETCH_RULEBOOKS  in device-count mode SparseSequential builds the rulebooks of ALL its layers on a side
#                       stream at the start of forward, so the strided layers' builds overlap the first layers'
#                       compute (the builds depend on indices only).
#   OVERLAP_DW          conv backward launches dW on a side stream while dX and the previous layers' backward
#                       continue on the main stream.  Contract: parameter .grad is None when backward starts
#                       (autograd then just stores the tensor) and whoever reads the gradients first calls
#                       functional.join_side_streams() (psd/ddp.FlatGradAllReducer does; a fallback join also runs
#                       when the backward pass ends).
PREFETCH_RULEBOOKS = False
OVERLAP_DW = False
# WFS_PREFETCH_BEFORE_FIRST=1: the branch of the prefetched builds forks BEFORE the first layer when the batch came with its
# event offsets (a captured step's hand-over launch writes them) instead of behind the first layer's conv.  Off: same-box
# A/B 0.4813 / 0.4806 ms against 0.4716 / 0.4704 -- the builds then end 23 us earlier, but the main chain's first kernel
# (the SubM build) starts 18 us after the hand-over launch on the replay's second queue and shares the chip with the
# first strided build (27 us instead of 18)
PREFETCH_BEFORE_FIRST = os.environ.get("WFS_PREFETCH_BEFORE_FIRST", "0") == "1"
# every prefetched rulebook has its own edge back to the main chain: the first strided layer does not wait for the second
# one's build.  WFS_JOIN_PER_BUILD=0: one join for the whole branch (each cross-stream edge of a replayed graph costs
# 6 - 10 us; with the 512-thread builds the per-build edges win: 0.4642 / 0.4659 vs 0.4695 / 0.4699 ms per step)
JOIN_PER_BUILD = os.environ.get("WFS_JOIN_PER_BUILD", "1") != "0"

# Event-local SubM rulebook build (round 3; csrc/evrulebook.hip): in device-count mode -- captured steps, where the index
# rows come from the reference's collate_fn, i.e. grouped by event -- a SubM rulebook is built by a pair of workgroups
# per event with the event's site table in LDS: no site grid over the batch in HBM (18.5 MB cleared per build at the PSD
# batch), no global atomics, traffic = coordinates in + table out; 12.7 vs 22 us alone, the same inside the step;
# bit-identical tables.  The grouping is verified on the device by every build; a batch that violates it (or an event
# beyond the LDS tables, or duplicate coordinates) raises a flag that GraphedTrainStep.check() reports like a capacity
# overflow.  WFS_EVENT_LOCAL=0: the chip-wide build.
EVENT_LOCAL = os.environ.get("WFS_EVENT_LOCAL", "1") != "0"
EVENT_LOCAL_MAX_BATCH = 16384

# Event-local build of REGULAR (strided) convolutions (round 4; csrc/evconv.hip): in device-count mode one launch, one
# workgroup per event with the event's output grid in LDS, instead of the chip-wide build's six launches over dense
# -1-padded tables.  WFS_EVENT_LOCAL_CONV=0: the chip-wide build.  PACKED_TABLES: where the kernel is no longer than the
# stride along the last dimension the by-input table is [K / kl, N] (include/wfsparse.h "packed tables"), consumed as it
# is by the 32 -> 32 dX / dW kernels and expanded on demand for everybody else (Rulebook.nbr_out).
EVENT_LOCAL_CONV = os.environ.get("WFS_EVENT_LOCAL_CONV", "1") != "0"
PACKED_TABLES = os.environ.get("WFS_PACKED_TABLES", "1") != "0"

_SIDE_STREAMS = {}


def side_stream(device, which=0):
    key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), which)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device)
    return _SIDE_STREAMS[key]


def _listify(v, ndim):
    if isinstance(v, (list, tuple, np.ndarray)):
        v = [int(x) for x in v]
        assert len(v) == ndim, "expected %d values, got %s" % (ndim, v)
        return v
    if torch.is_tensor(v):
        return _listify(v.tolist(), ndim)
    return [int(v)] * ndim


def get_conv_output_size(input_size, kernel_size, stride, padding, dilation):
    ndim = len(input_size)
    out = []
    for i in range(ndim):
        size = (int(input_size[i]) + 2 * padding[i] - dilation[i] * (kernel_size[i] - 1) - 1) // stride[i] + 1
        out.append(int(input_size[i]) if kernel_size[i] == -1 else size)
    return out


def get_deconv_output_size(input_size, kernel_size, stride, padding, dilation, output_padding):
    ndim = len(input_size)
    out = []
    for i in range(ndim):
        if kernel_size[i] == -1:
            raise ValueError("deconv don't support kernel_size < 0")
        out.append((int(input_size[i]) - 1) * stride[i] - 2 * padding[i] + kernel_size[i] + output_padding[i])
    return out


class Rulebook(object):
    """Device-resident rulebook of one sparse convolution geometry.

    nbr_out int32 [K, N]: output row of (input row j, offset k) or -1
    nbr_in  int32 [K, M]: input row of (output row i, offset k) or -1; for SubM with odd kernels and
                          dilation 1 it is nbr_out with the k axis mirrored (``kmap``), not stored.
    """

    def __init__(self):
        self.geometry = None
        self.N = self.M = self.K = 0
        self.subm = False
        self.has_dup = False
        self.indices = None          # input indices [N, D+1]
        self.out_indices = None      # [M, D+1] (SubM: the input indices)
        self._nbr_out = None
        # packed by-input table [K / packed_kl, N] of the event-local conv build (include/wfsparse.h "packed tables");
        # .nbr_out then expands it on first use
        self.nbr_out_packed = None
        self.packed_kl = 0
        self.nbr_in = None
        self.kmap_in = None          # ctypes int32[K] or None
        self.centre_k = -1           # SubM: offset computed as a plain X.W[k] (spconv's k*), else -1
        self.out_spatial_shape = None
        # device-count mode (HIP-graph capturable): N / M above are CAPACITIES, the valid row counts live here
        self.n_dev = None            # int64 [1] on the GPU: valid input rows
        self.m_dev = None            # int64 [1]: valid output rows (SubM: n_dev itself)
        self.overflow = None         # int32 [1]: set by the build if M exceeded the output capacity
        self.ready = None            # torch.cuda.Event when the build ran on a side stream (prefetch)
        # regular conv built on a direct grid: (ticket ptr, slot_id ptr, workspace kept alive, out volume) -- the
        # cell -> output row map of the build, which dense() of the conv's output can use (wfs_to_dense_mapped)
        self.cell_map = None
        # event-local build (device-count mode, see EVENT_LOCAL): offsets of the events in the row set
        # (wfs_event_offsets) and the build's failure flags (any word != 0 in the first two thirds: the table is
        # incomplete -- reported by the captured step's check())
        self.events_in = None
        self.events_out = None       # event-local conv build: the event offsets of the OUTPUT rows
        self.event_flags = None
        self._pairs = None
        self._pair_num = None

    @property
    def nbr_out(self):
        if self._nbr_out is None and self.nbr_out_packed is not None:
            dense = torch.empty((self.K, self.N), dtype=torch.int32, device=self.nbr_out_packed.device)
            _lib.check(_lib.load().wfs_unpack_table(_lib.ptr(self.nbr_out_packed), self.K, self.packed_kl, self.N,
                                                    _lib.ptr(self.n_dev), _lib.ptr(dense), _lib.stream_ptr()))
            self._nbr_out = dense
        return self._nbr_out

    @nbr_out.setter
    def nbr_out(self, t):
        self._nbr_out = t

    def table_by_in(self, Ca, Cb, like, which):
        """(table, packed_kl) for a product gathering through the by-input table: the packed form when the kernels of
        that product take it (which = 1: dX, 3: dW), else the dense one."""
        if (self.nbr_out_packed is not None and like.is_cuda
                and like.dtype in (torch.float32, torch.bfloat16, torch.float16)
                and _lib.load().wfs_gather_packed_ok(self.packed_kl, self.K, int(Ca), int(Cb), _lib.dtype_code(like), which)):
            return self.nbr_out_packed, self.packed_kl
        return self.nbr_out, 0

    # gather table addressed by OUTPUT rows, for the forward of conv / SubM
    def table_by_out(self):
        if self.nbr_in is not None:
            return self.nbr_in, None
        return self.nbr_out, self.kmap_in

    def _emit_pairs(self, want_pairs):
        lib = _lib.load()
        dev = self.indices.device
        if self._pair_num is None or (want_pairs and self._pairs is None):
            num = torch.empty((self.K,), dtype=torch.int32, device=dev)
            pairs = torch.empty((2, self.K, self.N), dtype=torch.int32, device=dev) if want_pairs else None
            if self.N == 0:
                num.zero_()
            else:
                # a finished rulebook is compacted with NULL nbr_in / out_indices under SubM rules, so
                # that emit runs only the compaction kernels (they read nbr_out and the tile counters)
                cg = self._compact_geometry()
                nbytes = lib.wfs_rulebook_workspace_bytes(ctypes.byref(cg), self.N)
                ws = torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=dev)
                _lib.check(lib.wfs_rulebook_emit(ctypes.byref(cg), _lib.ptr(self.indices),
                                                 self.N, self.M, _lib.ptr(self.nbr_out), None, None,
                                                 _lib.ptr(pairs), _lib.ptr(num), _lib.ptr(ws), ws.numel(),
                                                 _lib.ptr(self.n_dev), None, _lib.stream_ptr()))
            self._pair_num = num
            if want_pairs:
                self._pairs = pairs

    def _compact_geometry(self):
        # compaction only reads nbr_out; a SubM-flagged copy of the geometry makes emit skip the
        # regular-conv numbering kernels (they already ran)
        g = _lib.Geometry()
        ctypes.memmove(ctypes.byref(g), ctypes.byref(self.geometry), ctypes.sizeof(g))
        g.subm = 1
        g.transposed = 0           # ... and the tables are already in spconv's offset order (emit flipped them once)
        return g

    @property
    def indice_pair_num(self):
        self._emit_pairs(False)
        return self._pair_num

    @property
    def indice_pairs(self):
        self._emit_pairs(True)
        return self._pairs


def default_out_capacity(n_cap, K, cells):
    """Rows to reserve for a regular conv's outputs when their number is not read back: every input can
    open at most K sites and there are at most `cells` sites."""
    return int(min(n_cap * K, cells))


_REUSE = None          # {key: Rulebook} while a reuse_rulebooks() context is open
BUILD_COUNT = 0        # rulebooks actually built (tests / diagnostics)


class reuse_rulebooks(object):
    """Context manager: rulebooks depend on the index rows and the layer geometry only, so while it is open a build for
    the same index tensor (same storage, same shape) and the same geometry returns the rulebook built the first time --
    across forward calls and across layers that have no ``indice_key``.  Made for evaluation sweeps that run ONE batch
    through the net many times with different features (the reference's occlusion study zeroes one feature column per
    pass, scripts/RunOcclusionStudy.py -> Evaluate.py --occlude, src/engineering/LitPSD.py:133-135).
    Contract: index tensors are not modified in place while the context is open (the cache keeps them alive)."""

    def __enter__(self):
        global _REUSE
        self._outer = _REUSE
        if _REUSE is None:
            _REUSE = {}
        return self

    def __exit__(self, *exc):
        global _REUSE
        _REUSE = self._outer
        return False


def reused(tag, source, make):
    """``make()`` -- or, inside a reuse_rulebooks() context, the tensor it returned the first time for the same
    ``source`` storage.  For index tensors DERIVED from the batch (the reference permutes the coordinate columns on
    every forward, src/models/SPConvNet.py:64), so that the derived tensor -- the rulebook cache's key -- is stable."""
    if _REUSE is None:
        return make()
    key = (tag, source.data_ptr(), tuple(source.shape), tuple(source.stride()))
    hit = _REUSE.get(key)
    if hit is None:
        hit = _REUSE[key] = (make(), source)
    return hit[0]


def event_offsets(indices, batch_size, n_dev=None):
    """int32 [batch + 1 + flag words]: first row of every event of an index set grouped by event (wfs_event_offsets)."""
    lib = _lib.load()
    out = torch.empty((int(lib.wfs_event_offsets_ints(int(batch_size))),), dtype=torch.int32, device=indices.device)
    _lib.check(lib.wfs_event_offsets(_lib.ptr(indices), indices.shape[0], indices.shape[1] - 1, int(batch_size),
                                     _lib.ptr(n_dev), _lib.ptr(out), _lib.stream_ptr()))
    return out


def _sticky_flags(n, dev, store=None, name=None):
    """Failure flags a build only ever SETS (overflow of an output capacity, event-local build failures); whoever reads
    them clears them.  ``store`` (a dict owned by the calling layer): the tensor is created once, OUTSIDE any graph
    capture, and handed to every later build of that layer -- memory allocated inside a capture can be the recycled block
    of an earlier temporary of the same graph, which every replay then writes before the build runs: fine for a flag the
    build rewrites, fatal for one it only sets.  Without a store: zeros (empty inside a capture: such a flag is only good
    until the next replay)."""
    capturing = dev.type == "cuda" and torch.cuda.is_current_stream_capturing()
    if store is not None:
        t = store.get(name)
        if t is None or t.numel() != n or t.device != dev:
            if capturing:
                raise RuntimeError("waveformml_amd.spconv: the layer's failure flags must exist before a graph capture "
                                   "(run the step once in device-count mode first, as psd/graph.py does)")
            t = torch.zeros((n,), dtype=torch.int32, device=dev)
            store[name] = t
        return t
    if capturing:
        return torch.empty((n,), dtype=torch.int32, device=dev)
    return torch.zeros((n,), dtype=torch.int32, device=dev)


def build_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, subm,
                   known_unique=None, n_dev=None, out_capacity=None, transposed=False, output_padding=None,
                   events=None, flags=None, want_cell_map=True):
    """Cached front of :func:`_build_rulebook` (see :class:`reuse_rulebooks`).  ``flags``: the calling layer's store of
    sticky failure flags (:func:`_sticky_flags`)."""
    global BUILD_COUNT
    if transposed:
        BUILD_COUNT += 1
        return _build_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, False,
                               known_unique, n_dev, out_capacity, True, output_padding, flags=flags)
    if _REUSE is None:
        BUILD_COUNT += 1
        return _build_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, subm,
                               known_unique, n_dev, out_capacity, events=events, flags=flags,
                               want_cell_map=want_cell_map)
    ndim = indices.shape[1] - 1
    key = (indices.data_ptr(), tuple(indices.shape), tuple(indices.stride()), int(batch_size),
           tuple(int(s) for s in spatial_shape), tuple(_listify(ksize, ndim)), tuple(_listify(stride, ndim)),
           tuple(_listify(padding, ndim)), tuple(_listify(dilation, ndim)), bool(subm),
           None if n_dev is None else n_dev.data_ptr(), out_capacity)
    rb = _REUSE.get(key)
    if rb is None:
        BUILD_COUNT += 1
        rb = _build_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, subm,
                             known_unique, n_dev, out_capacity, events=events, flags=flags)
        rb._keepalive = indices          # the key holds a data_ptr: keep the storage from being recycled
        _REUSE[key] = rb
    return rb


def _event_local_conv(rb, g, lib, indices, batch_size, n_dev, m_cap, events, flags, want_cell_map, stream):
    """Device-count build of a regular conv by wfs_event_rulebook_conv (csrc/evconv.hip): one launch."""
    dev = indices.device
    N, ndim, K = rb.N, indices.shape[1] - 1, rb.K
    B = int(batch_size)
    kl = int(lib.wfs_event_rulebook_conv_packed_kl(ctypes.byref(g))) if PACKED_TABLES else 0
    rb.M = m_cap
    rb.m_dev = torch.empty((1,), dtype=torch.int64, device=dev)
    rb.overflow = _sticky_flags(1, dev, flags, "overflow")
    rb.event_flags = _sticky_flags(3, dev, flags, "conv_events")
    # the launch epoch and the per-event counts the workgroups exchange: zeroed ONCE, then owned by this layer's builds
    n_state = int(lib.wfs_event_rulebook_conv_state_bytes(B)) // 4
    if flags is not None:
        state = _sticky_flags(n_state, dev, flags, "conv_state")
    else:
        state = torch.zeros((n_state,), dtype=torch.int32, device=dev)      # inside a capture: re-zeroed by every replay
    rb.events_in = events if events is not None else event_offsets(indices, B, n_dev)
    rb.events_out = torch.empty((int(lib.wfs_event_offsets_ints(B)),), dtype=torch.int32, device=dev)
    rb.out_indices = torch.empty((m_cap, ndim + 1), dtype=torch.int32, device=dev)
    rb.nbr_in = torch.empty((K, m_cap), dtype=torch.int32, device=dev)
    if kl:
        rb.nbr_out_packed, rb.packed_kl = torch.empty((K // kl, N), dtype=torch.int32, device=dev), kl
        table = rb.nbr_out_packed
    else:
        rb._nbr_out = torch.empty((K, N), dtype=torch.int32, device=dev)
        table = rb._nbr_out
    V = int(np.prod(rb.out_spatial_shape))
    cell_row = torch.empty((B * V,), dtype=torch.int32, device=dev) if want_cell_map else None
    _lib.check(lib.wfs_event_rulebook_conv(ctypes.byref(g), _lib.ptr(indices), N, _lib.ptr(n_dev), _lib.ptr(rb.events_in),
                                           m_cap, _lib.ptr(rb.out_indices), _lib.ptr(rb.m_dev), _lib.ptr(rb.events_out),
                                           _lib.ptr(table), kl, _lib.ptr(rb.nbr_in), _lib.ptr(cell_row),
                                           _lib.ptr(rb.overflow), _lib.ptr(rb.event_flags), _lib.ptr(state), stream))
    if cell_row is not None:
        # wfs_to_dense_mapped's (ticket, slot_id) pair: "-1 = no row" reads as an unset ticket
        rb.cell_map = (cell_row.data_ptr(), cell_row.data_ptr(), cell_row, V)
    rb.has_dup = False
    return rb


def _build_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, subm,
                    known_unique=None, n_dev=None, out_capacity=None, transposed=False, output_padding=None,
                    events=None, flags=None, want_cell_map=True):
    """Builds the device rulebook.  ``known_unique``: True if the caller knows the index rows are
    distinct sites (skips the duplicate check a regular conv would otherwise run once).

    ``n_dev`` (int64 [1] device tensor) switches to device-count mode: ``indices`` holds a CAPACITY of
    rows of which the first ``n_dev[0]`` are valid, nothing is read back to the host (the build is
    asynchronous and HIP-graph capturable), indices are trusted, and a regular conv's outputs get
    ``out_capacity`` rows (default: min(N*K, batch*volume)) with the true count in ``rulebook.m_dev``."""
    if not indices.is_cuda:
        raise RuntimeError("waveformml_amd.spconv: indices must be on the GPU (no CPU path)")
    if indices.dtype != torch.int32:
        raise RuntimeError("waveformml_amd.spconv: indices must be int32")
    indices = indices.contiguous()
    lib = _lib.load()
    N, ndim = indices.shape[0], indices.shape[1] - 1
    spatial_shape = [int(s) for s in spatial_shape]
    g = _lib.make_geometry(ndim, batch_size, spatial_shape, ksize, stride, padding, dilation, subm, transposed,
                           output_padding)
    rb = Rulebook()
    rb.geometry, rb.N, rb.K, rb.subm, rb.indices = g, N, int(g.K), bool(subm), indices
    rb.out_spatial_shape = [int(g.out_shape[i]) for i in range(ndim)]
    dev = indices.device
    stream = _lib.stream_ptr()
    if known_unique is None and (ASSUME_VALID_UNIQUE_INDICES or n_dev is not None):
        known_unique = True
    if (n_dev is not None and not subm and not transposed and EVENT_LOCAL_CONV and known_unique and N > 0
            and 1 <= int(batch_size) <= EVENT_LOCAL_MAX_BATCH and lib.wfs_event_rulebook_conv_ok(ctypes.byref(g))):
        assert n_dev.dtype == torch.int64 and n_dev.is_cuda and n_dev.numel() == 1
        rb.n_dev = n_dev
        cells = int(batch_size) * int(np.prod(rb.out_spatial_shape))
        m_cap = int(out_capacity) if out_capacity else default_out_capacity(N, rb.K, cells)
        if rb.K * m_cap < 2 ** 31 and m_cap < 2 ** 28:
            return _event_local_conv(rb, g, lib, indices, batch_size, n_dev, m_cap, events, flags, want_cell_map, stream)
    nbytes = lib.wfs_rulebook_workspace_bytes(ctypes.byref(g), N)
    ws = torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=dev)
    rb.nbr_out = torch.empty((rb.K, N), dtype=torch.int32, device=dev)
    if n_dev is not None:
        assert n_dev.dtype == torch.int64 and n_dev.is_cuda and n_dev.numel() == 1
        rb.n_dev = n_dev
        if subm:
            rb.M, rb.m_dev = N, n_dev
            m_cap = 0
        else:
            cells = int(batch_size) * int(np.prod(rb.out_spatial_shape))
            m_cap = int(out_capacity) if out_capacity else default_out_capacity(N, rb.K, cells)
            rb.M = m_cap
            rb.m_dev = torch.empty((1,), dtype=torch.int64, device=dev)        # written by the plan
            rb.overflow = _sticky_flags(1, dev, flags, "overflow")           # set (never cleared) by the emit
        if (subm and EVENT_LOCAL and N > 0 and 1 <= int(batch_size) <= EVENT_LOCAL_MAX_BATCH
                and lib.wfs_event_rulebook_ok(ctypes.byref(g))):
            # a pair of workgroups per event, site table in LDS
            rb.events_in = events if events is not None else event_offsets(indices, batch_size, n_dev)
            rb.event_flags = _sticky_flags(int(lib.wfs_event_rulebook_flag_ints(int(batch_size))), dev, flags, "events")
            _lib.check(lib.wfs_event_rulebook_subm(ctypes.byref(g), _lib.ptr(indices), N, _lib.ptr(n_dev),
                                                   _lib.ptr(rb.events_in), _lib.ptr(rb.nbr_out), None,
                                                   _lib.ptr(rb.event_flags), stream))
        else:
            _lib.check(lib.wfs_rulebook_plan(ctypes.byref(g), _lib.ptr(indices), N, _lib.ptr(rb.nbr_out), _lib.ptr(ws),
                                             ws.numel(), None, _lib.ptr(n_dev), None if subm else _lib.ptr(rb.m_dev),
                                             m_cap, stream))
        rb.has_dup = False
    elif subm and known_unique:
        _lib.check(lib.wfs_rulebook_plan(ctypes.byref(g), _lib.ptr(indices), N, _lib.ptr(rb.nbr_out), _lib.ptr(ws),
                                         ws.numel(), None, None, None, 0, stream))
        rb.M, rb.has_dup = N, False
    else:
        info = (ctypes.c_int64 * 2)(0, 0)
        _lib.check(lib.wfs_rulebook_plan(ctypes.byref(g), _lib.ptr(indices), N, _lib.ptr(rb.nbr_out), _lib.ptr(ws),
                                         ws.numel(), info, None, None, 0, stream))
        rb.M = int(info[0])
        rb.has_dup = bool(info[1])
    symmetric = all(int(g.ksize[i]) % 2 == 1 and int(g.dilation[i]) == 1 for i in range(ndim))
    if subm:
        rb.out_indices = indices
        if symmetric:
            rb.kmap_in = _lib.i32_array([rb.K - 1 - k for k in range(rb.K)])
            rb.centre_k = rb.K // 2
        else:
            rb.nbr_in = torch.empty((rb.K, rb.M), dtype=torch.int32, device=dev)
            _lib.check(lib.wfs_rulebook_emit(ctypes.byref(g), _lib.ptr(indices), N, rb.M, _lib.ptr(rb.nbr_out), None,
                                             _lib.ptr(rb.nbr_in), None, None, _lib.ptr(ws), ws.numel(),
                                             _lib.ptr(rb.n_dev), None, stream))
        if (rb.has_dup or not symmetric) and n_dev is None:
            # spconv computes offset k* = argmax(indice_pair_num) as a plain X.W[k*] (A.4); with
            # distinct sites and an odd kernel that is the centre, otherwise it has to be looked up
            rb.centre_k = int(np.argmax(rb.indice_pair_num.cpu().numpy())) if N > 0 else 0   # first max, as A.4
    else:
        if known_unique is None and N > 0 and n_dev is None:
            gs = _lib.make_geometry(ndim, batch_size, spatial_shape, [1] * ndim, [1] * ndim, [0] * ndim,
                                    [1] * ndim, True)
            nb2 = lib.wfs_rulebook_workspace_bytes(ctypes.byref(gs), N)
            ws2 = torch.empty((max(int(nb2), 1),), dtype=torch.uint8, device=dev)
            info2 = (ctypes.c_int64 * 2)(0, 0)
            _lib.check(lib.wfs_indices_check(ctypes.byref(gs), _lib.ptr(indices), N, _lib.ptr(ws2), ws2.numel(),
                                             info2, stream))
            rb.has_dup = bool(info2[1])
        elif known_unique is not None:
            rb.has_dup = not known_unique
        rb.out_indices = torch.empty((rb.M, ndim + 1), dtype=torch.int32, device=dev)
        rb.nbr_in = torch.empty((rb.K, rb.M), dtype=torch.int32, device=dev)
        _lib.check(lib.wfs_rulebook_emit(ctypes.byref(g), _lib.ptr(indices), N, rb.M, _lib.ptr(rb.nbr_out),
                                         _lib.ptr(rb.out_indices), _lib.ptr(rb.nbr_in), None, None, _lib.ptr(ws),
                                         ws.numel(), _lib.ptr(rb.n_dev), _lib.ptr(rb.overflow), stream))
        if not rb.has_dup and N > 0:
            ticket, slot, cells = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64(0)
            if lib.wfs_rulebook_cell_map(ctypes.byref(g), N, _lib.ptr(ws), ctypes.byref(ticket), ctypes.byref(slot),
                                         ctypes.byref(cells)):
                rb.cell_map = (ticket.value, slot.value, ws, int(np.prod(rb.out_spatial_shape)))
    return rb


def get_indice_pairs(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1,
                     out_padding=0, subm=False, transpose=False, grid=None, use_hash=False):
    """spconv.ops.get_indice_pairs: returns (out_indices, indice_pairs [2,K,N], indice_pair_num [K])."""
    ndim = indices.shape[1] - 1
    ksize, stride = _listify(ksize, ndim), _listify(stride, ndim)
    padding, dilation = _listify(padding, ndim), _listify(dilation, ndim)
    for d, s in zip(dilation, stride):
        assert any([s == 1, d == 1]), "don't support this."
    if transpose:
        assert not subm
        rb = build_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, False,
                            transposed=True, output_padding=_listify(out_padding, ndim))
    else:
        rb = build_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, subm)
    return rb.out_indices, rb.indice_pairs, rb.indice_pair_num

"""Autograd functions over libwfsparse: counterpart of ``spconv.functional`` (spconv 1.2.1,
SURVEY.md A.4: ``indice_conv`` / ``indice_subm_conv`` / ``indice_inverse_conv`` and their backward).

All three are the same output-stationary gather kernel with a different table (include/wfsparse.h):

    mode      forward table (rows R)        dX table (rows)          dW stationary / gathered
    conv      nbr_in  [K,M]  gathers X[N]   nbr_out [K,N] dY[M]      S = X [N],  G = dY, nbr_out
    subm      nbr_out mirrored (or nbr_in)  nbr_out [K,N] dY[N]      S = X [N],  G = dY, nbr_out
    inverse   nbr_out [K,N]  gathers X[M]   nbr_in  [K,M] dY[N]      S = dY [N], G = X,  nbr_out (swap)

With duplicate input coordinates the inverse of nbr_out is not a function; the two products that
need it (conv/subm forward, inverse dX) then use the atomic scatter form.
"""
import ctypes

import torch
from torch.autograd import Function

from .. import _lib

CONV, SUBM, INVERSE = 0, 1, 2

# bench.py's accounting pass (never active inside a timed region): when this is a list, every launch
# appends its ALGORITHMIC work as SURVEY.md 8d defines it -- each tensor counted once, the rulebook
# at 8 bytes per pair (spconv's encoding), filters fp32.
ACCOUNT = None


def _account(kind, table, R, rows_in, c_in, rows_out, c_out, K, cw_in, cw_out, esize):
    if ACCOUNT is None:
        return
    pairs = int((table >= 0).sum().item()) if table is not None else int(R)
    ACCOUNT.append(dict(kind=kind, pairs=pairs,
                        bytes=rows_in * c_in * esize + rows_out * c_out * esize + pairs * 8 + K * cw_in * cw_out * 4,
                        flops=2 * pairs * cw_in * cw_out))


def _features_ok(t):
    if not t.is_cuda:
        raise RuntimeError("waveformml_amd.spconv: features must be on the GPU (there is no CPU path)")
    return t.contiguous()


def _rows(shape, like, r_dev):
    """Row-dimensioned output.  In device-count mode rows beyond the valid count are never written NOR read
    by the kernels, so they can stay uninitialised (the conv bias gradient's column sum honours the count too)."""
    return torch.empty(shape, dtype=like.dtype, device=like.device)


def _masked_column_sum(t, r_dev):
    """sum over the valid rows of t [R, C] (fp32): the conv bias gradient (include/wfsparse.h wfs_column_sum)."""
    lib = _lib.load()
    t = t.contiguous()
    R, C = int(t.shape[0]), int(t.shape[1])
    out = torch.empty((C,), dtype=torch.float32, device=t.device)
    ws = torch.empty((int(lib.wfs_column_sum_workspace_bytes(C)),), dtype=torch.uint8, device=t.device)
    _lib.check(lib.wfs_column_sum(_lib.ptr(t), R, C, _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.dtype_code(t),
                                  _lib.ptr(r_dev), _lib.stream_ptr()))
    return out


# Every layer runs in libwfsparse: the 32- and 2-channel MFMA kernels of conv_mfma.hip, the shape-generic 32 x 32-tile MFMA
# kernels of gather_conv.hip (k_gconv_mfma / k_gdw_mfma: 16 / 24 / 64-channel layers) and, from 128 channels on a side,
# the dense 128 x 128-tile matrix-core products of wide.hip (the reference's GEP.json 300 -> 252 -> 158 -> 64 stack,
# SparseConv2DPreserve's 130 ... 154 channels, the hybrid net's 2048 -> 1697 -> 1021 -> 345 stack of BASELINE
# configs[4]) -- 16-bit MFMA for 16-bit rows, exact-fp32 MFMA for fp32 rows.  No torch.mm / library GEMM on this path.


def _fast_shape(Cx, Cy):
    """Channel pairs with shape-specialised kernels (conv_mfma.hip)."""
    return (Cx == 32 and Cy == 32) or (Cx == 2 and Cy == 32) or (Cx == 32 and Cy == 2)


def filters16(W, like):
    """[K, Cin, Cout] fp32 filters in the row type of ``like``, padded as the wide products read them (one conversion
    per layer and step: the forward pass keeps it for dX).  None for fp32 rows whose filters are read in place."""
    lib = _lib.load()
    K, Cw_in, Cw_out = int(W.shape[0]), int(W.shape[1]), int(W.shape[2])
    if like.dtype == torch.float32 and Cw_out % 4 == 0:
        return None
    code = _lib.dtype_code(like)
    out = torch.empty((int(lib.wfs_wide_filters_bytes(K, Cw_in, Cw_out, code)),), dtype=torch.uint8, device=W.device)
    _lib.check(lib.wfs_wide_filters(_lib.ptr(W), K, Cw_in, Cw_out, code, _lib.ptr(out), _lib.stream_ptr()))
    return out


def takes_wide_path(K, R, X, Cy):
    return bool(X.is_cuda and X.dtype in (torch.float32, torch.bfloat16, torch.float16)
                and _lib.load().wfs_wide_conv_ok(K, R, X.shape[0], int(X.shape[1]), int(Cy), _lib.dtype_code(X)))


def _wide_gather_conv(lib, table, kmap, K, identity_k, R, X, W, transpose_w, bias, r_dev, w16=None):
    """>= 128 channels on a side: one dense matrix-core product over the shorter side of the layer
    (csrc/wide.hip; include/wfsparse.h wfs_wide_gather_conv)."""
    Cw_in, Cw_out = int(W.shape[-2]), int(W.shape[-1])
    Cx, Cy = int(X.shape[1]), (Cw_in if transpose_w else Cw_out)
    assert W.dtype == torch.float32 and W.is_contiguous() and X.is_contiguous()
    assert Cx == (Cw_out if transpose_w else Cw_in), (X.shape, W.shape, transpose_w)
    assert table is None or (table.dtype == torch.int32 and table.shape == (K, R)), (None if table is None else table.shape, K, R)
    assert bias is None or (bias.dtype == torch.float32 and bias.numel() == Cy)
    Y = _rows((R, Cy), X, r_dev)
    nbytes = lib.wfs_wide_conv_workspace_bytes(K, R, X.shape[0], Cx, Cy, 0 if table is None else 1, _lib.dtype_code(X))
    ws = torch.empty((int(nbytes),), dtype=torch.uint8, device=X.device)
    _lib.check(lib.wfs_wide_gather_conv(_lib.ptr(table), kmap, K, identity_k, R, _lib.ptr(X), X.shape[0], Cx, _lib.ptr(W),
                                        _lib.ptr(w16), Cw_in, Cw_out, 1 if transpose_w else 0, _lib.ptr(bias), _lib.ptr(Y),
                                        _lib.dtype_code(X), _lib.ptr(r_dev), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
    _account("gather_conv", table, R, X.shape[0], Cx, R, Cy, K, Cw_in, Cw_out, X.element_size())
    return Y


def gather_conv(table, kmap, K, identity_k, R, X, W, transpose_w, bias, r_dev=None, w16=None, packed_kl=0):
    """Y[r] = bias + sum_k X[table[kmap[k], r]] . W[k]   (W fp32 [K, Cin, Cout]; ^T if transpose_w).
    ``packed_kl`` > 0: ``table`` is the packed by-input table [K / packed_kl, R] (Rulebook.table_by_in)."""
    lib = _lib.load()
    Cw_in, Cw_out = int(W.shape[-2]), int(W.shape[-1])
    if packed_kl:
        assert transpose_w and kmap is None and table.shape == (K // packed_kl, R), (table.shape, K, packed_kl, R)
        Y = _rows((R, Cw_in), X, r_dev)
        _lib.check(lib.wfs_gather_conv(_lib.ptr(table), None, K, identity_k, R, _lib.ptr(X), X.shape[0], X.shape[1],
                                       _lib.ptr(W), Cw_in, Cw_out, 1, None, _lib.ptr(Y), _lib.dtype_code(X),
                                       _lib.ptr(r_dev), packed_kl, _lib.stream_ptr()))
        if ACCOUNT is not None:
            _account("gather_conv", (table >> 3).clamp_(min=-1), R, X.shape[0], X.shape[1], R, Cw_in, K, Cw_in, Cw_out,
                     X.element_size())
        return Y
    if X.is_cuda and lib.wfs_wide_conv_ok(K, R, X.shape[0], int(X.shape[1]), Cw_in if transpose_w else Cw_out, _lib.dtype_code(X)):
        return _wide_gather_conv(lib, table, kmap, K, identity_k, R, X, W, transpose_w, bias, r_dev, w16)
    if transpose_w and not _fast_shape(Cw_out, Cw_in):
        # the shape-generic MFMA kernel reads the filter with the OUTPUT channel on the lanes: for dX that is a strided
        # walk over W[k] (a new 128-B line per lane and step; 47 vs 26 us measured at 64 channels) -- hand it W[k]^T
        # instead (one small transpose launch: the filters are at most a few MB)
        W, transpose_w = W.transpose(1, 2).contiguous(), False
        Cw_in, Cw_out = Cw_out, Cw_in
    Cy = Cw_in if transpose_w else Cw_out
    Y = _rows((R, Cy), X, r_dev)
    assert W.dtype == torch.float32 and W.is_contiguous()
    assert X.dim() == 2 and X.shape[1] == (Cw_out if transpose_w else Cw_in), (X.shape, W.shape, transpose_w)
    assert table is None or (table.dtype == torch.int32 and table.shape == (K, R)), (None if table is None else table.shape, K, R)
    assert bias is None or (bias.dtype == torch.float32 and bias.numel() == Cy)
    _lib.check(lib.wfs_gather_conv(_lib.ptr(table), kmap, K, identity_k, R, _lib.ptr(X), X.shape[0], X.shape[1],
                                   _lib.ptr(W), Cw_in, Cw_out, 1 if transpose_w else 0, _lib.ptr(bias),
                                   _lib.ptr(Y), _lib.dtype_code(X), _lib.ptr(r_dev), 0, _lib.stream_ptr()))
    _account("gather_conv", table, R, X.shape[0], X.shape[1], R, Cy, K, Cw_in, Cw_out, X.element_size())
    return Y


def scatter_conv(table, K, identity_k, R, X, W, transpose_w, n_out, bias):
    """fp32-atomic form for duplicate coordinates: Y[table[k, r]] += X[r] . W[k]."""
    lib = _lib.load()
    Cw_in, Cw_out = int(W.shape[-2]), int(W.shape[-1])
    Cy = Cw_in if transpose_w else Cw_out
    Y = torch.zeros((n_out, Cy), dtype=torch.float32, device=X.device)
    if bias is not None:
        Y += bias
    assert table.dtype == torch.int32 and table.shape == (K, R)
    _lib.check(lib.wfs_scatter_conv(_lib.ptr(table), K, identity_k, R, _lib.ptr(X), X.shape[1], _lib.ptr(W), Cw_in,
                                    Cw_out, 1 if transpose_w else 0, _lib.ptr(Y), _lib.dtype_code(X),
                                    _lib.stream_ptr()))
    return Y.to(X.dtype)


_PENDING_SIDE = []          # (event, tensors kept alive until the join) of launches made on a side stream


# ---- parameter gradients written where the optimizer reads them ---------------------------------------------------
# A runner that keeps all parameters / gradients in ONE flat buffer each (psd/ddp.FlatGradAllReducer) registers the
# pair here.  Backward passes then allocate a parameter's gradient as a VIEW of its slot in the flat gradient buffer
# (grad_like), so the kernels write it in place and the runner's pack step has nothing left to copy.
_GRAD_SLOTS = []          # [(weakref(flat_param), weakref(flat_grad))]


def register_grad_slots(flat_param, flat_grad):
    import weakref
    _GRAD_SLOTS[:] = [(p, g) for (p, g) in _GRAD_SLOTS if p() is not None and g() is not None]
    _GRAD_SLOTS.append((weakref.ref(flat_param), weakref.ref(flat_grad)))


_SLOTS_WRITTEN = set()    # slot addresses handed out during the current backward pass


def reset_grad_slots():
    """Forget which slots the current / last backward pass wrote (runs at the end of every backward pass that handed
    one out, and from FlatGradAllReducer.reset())."""
    _SLOTS_WRITTEN.clear()


def grad_like(param, shape=None):
    """An fp32 tensor of ``shape`` (default: param's) for d loss / d param: the parameter's slot in a registered flat
    gradient buffer when ``param`` is a contiguous fp32 view into the matching flat parameter buffer, else fresh memory.

    A slot is handed out ONCE per backward pass and only while the parameter has no gradient yet: a module called twice
    in one graph (two gradients for one parameter, summed by autograd) or a second backward() without reset (gradient
    accumulation, ``p.grad += new``) gets fresh memory for the further gradients -- two operands of autograd's add
    must never alias the same slot."""
    shape = tuple(param.shape) if shape is None else tuple(shape)
    if (param is not None and param.is_cuda and param.dtype == torch.float32 and param.is_contiguous()
            and param.grad is None):
        for pref, gref in _GRAD_SLOTS:
            fp, fg = pref(), gref()
            if fp is None or fg is None or fp.device != param.device or fg.dtype != torch.float32:
                continue
            off = param.data_ptr() - fp.data_ptr()
            if 0 <= off and off + 4 * param.numel() <= 4 * fp.numel() and off % 4 == 0:
                slot = fg.data_ptr() + off
                if slot in _SLOTS_WRITTEN:
                    # a second gradient for this parameter in one pass: autograd is about to ADD it to the first one,
                    # so a pending (deferred) second stage of the first must land in the slot now, and the slot no
                    # longer counts as "filled by a deferred reduction" (the sum is wherever autograd puts it)
                    flush_deferred_dw()
                    _SHARED_SLOTS.add(slot)
                    break
                if not _SLOTS_WRITTEN:
                    try:
                        torch.autograd.Variable._execution_engine.queue_callback(reset_grad_slots)
                    except RuntimeError:          # not inside a backward pass (direct call): the caller resets
                        pass
                _SLOTS_WRITTEN.add(slot)
                return fg[off // 4: off // 4 + param.numel()].view(shape)
    return torch.empty(shape, dtype=torch.float32, device=param.device)


# Deferred second stages of gather_dw (include/wfsparse.h, wfs_dw_job): while a runner has deferral switched on, the
# slab reductions of a whole backward pass are collected and run as ONE launch by flush_deferred_dw() -- only for
# gradients that live in a registered slot (the runner reads the slot, never the tensor autograd holds).
_DEFERRED_DW = None       # None: off; else a list of (DwJob, workspace tensor)
_DEFERRED_SLOTS = set()   # slot addresses written by the last flush
_SHARED_SLOTS = set()     # slots that received more than one gradient in the pass (grad_like)


def defer_dw(on):
    global _DEFERRED_DW
    _DEFERRED_DW = [] if on else None
    _DEFERRED_SLOTS.clear()
    _SHARED_SLOTS.clear()


def was_deferred(slot_ptr):
    return slot_ptr in _DEFERRED_SLOTS and slot_ptr not in _SHARED_SLOTS


def flush_deferred_dw():
    """One launch for every pending slab reduction (no-op when nothing is pending)."""
    if not _DEFERRED_DW:
        return
    lib = _lib.load()
    jobs = (_lib.DwJob * len(_DEFERRED_DW))(*[j for (j, _ws) in _DEFERRED_DW])
    for at in range(0, len(_DEFERRED_DW), 16):
        n = min(16, len(_DEFERRED_DW) - at)
        _lib.check(lib.wfs_dw_reduce_jobs(ctypes.cast(ctypes.byref(jobs, at * ctypes.sizeof(_lib.DwJob)),
                                                      ctypes.POINTER(_lib.DwJob)), n, _lib.stream_ptr()))
    for j, _ws in _DEFERRED_DW:
        _DEFERRED_SLOTS.add(int(j.dW))
    del _DEFERRED_DW[:]


def join_side_streams():
    """Make the current stream wait for everything launched on side streams (OVERLAP_DW) and release the
    tensors that were kept alive for them."""
    if _PENDING_SIDE:
        cur = torch.cuda.current_stream()
        for ev, _keep in _PENDING_SIDE:
            cur.wait_event(ev)
        del _PENDING_SIDE[:]


def gather_dw(table, K, identity_k, R, S, G, swap, kmap=None, r_dev=None, overlap=False, like=None, packed_kl=0):
    """dW[k,a,b] = sum_r S[r,a] G[table[kmap[k],r], b]  (swap: dW[k,b,a]).
    overlap=True launches on the side stream (see ops.OVERLAP_DW): memory is allocated on the calling stream
    and every operand is kept alive until join_side_streams().  ``like``: the parameter this is the gradient of
    (grad_like: written straight into its slot of a registered flat gradient buffer)."""
    lib = _lib.load()
    Cs, Cg = int(S.shape[1]), int(G.shape[1])
    shape = (K, Cg, Cs) if swap else (K, Cs, Cg)
    in_slot = False
    if like is not None and like.numel() == K * Cs * Cg:
        dW = grad_like(like, shape)
        in_slot = dW._base is not None
    else:
        dW = torch.empty(shape, dtype=torch.float32, device=S.device)
    assert S.dtype == G.dtype and S.shape[0] == R
    assert table is None or (table.dtype == torch.int32 and table.shape == ((K // packed_kl if packed_kl else K), R))
    nbytes = lib.wfs_gather_dw_workspace_bytes(K, R, Cs, Cg)
    ws = torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=S.device)

    def launch():
        defer = _DEFERRED_DW is not None and in_slot and not overlap
        job = _lib.DwJob() if defer else None
        _lib.check(lib.wfs_gather_dw(_lib.ptr(table), kmap, K, identity_k, R, _lib.ptr(S), Cs, _lib.ptr(G), G.shape[0],
                                     Cg, 1 if swap else 0, _lib.ptr(dW), _lib.dtype_code(S), _lib.ptr(ws), ws.numel(),
                                     _lib.ptr(r_dev), ctypes.byref(job) if defer else None, packed_kl, _lib.stream_ptr()))
        if defer and job.nslabs > 0:
            _DEFERRED_DW.append((job, ws))          # second stage pending: flush_deferred_dw()

    if overlap and ACCOUNT is None:
        from . import ops
        main = torch.cuda.current_stream()
        side = ops.side_stream(S.device, 1)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            launch()
            ev = torch.cuda.Event()
            ev.record(side)
        if not _PENDING_SIDE:
            # fallback join at the end of this backward pass (the reducer normally joins earlier, before packing)
            torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)
        _PENDING_SIDE.append((ev, (table, S, G, dW, ws, r_dev)))
    else:
        launch()
    if ACCOUNT is not None:
        _account("gather_dw", (table >> 3).clamp_(min=-1) if packed_kl else table, R, R, Cs, G.shape[0], Cg, K, Cs, Cg,
                 S.element_size())
    return dW


def conv_backward(table, K, identity_k, R, X, dY, W, r_dev=None, like=None, packed_kl=0):
    """(dX, dW) of a conv / SubM layer in one call (include/wfsparse.h wfs_conv_backward): 32 -> 32 layers with 16-bit
    rows run both products in ONE launch.  ``like``: the filter parameter (its gradient goes straight into its slot of a
    registered flat gradient buffer, the slab reduction joins the step's deferred ones)."""
    lib = _lib.load()
    Cin, Cout = int(X.shape[1]), int(dY.shape[1])
    if like is not None and like.numel() == K * Cin * Cout:
        dW = grad_like(like, (K, Cin, Cout))
        in_slot = dW._base is not None
    else:
        dW, in_slot = torch.empty((K, Cin, Cout), dtype=torch.float32, device=X.device), False
    dX = _rows((R, Cin), dY, r_dev)
    ws = torch.empty((max(int(lib.wfs_gather_dw_workspace_bytes(K, R, Cin, Cout)), 1),), dtype=torch.uint8, device=X.device)
    defer = _DEFERRED_DW is not None and in_slot
    job = _lib.DwJob() if defer else None
    _lib.check(lib.wfs_conv_backward(_lib.ptr(table), K, identity_k, R, _lib.ptr(X), _lib.ptr(dY), dY.shape[0], Cin, Cout,
                                     _lib.ptr(W), _lib.ptr(dX), _lib.ptr(dW), _lib.dtype_code(X), _lib.ptr(ws), ws.numel(),
                                     _lib.ptr(r_dev), ctypes.byref(job) if defer else None, packed_kl, _lib.stream_ptr()))
    if defer and job.nslabs > 0:
        _DEFERRED_DW.append((job, ws))
    if ACCOUNT is not None:
        # SURVEY.md 8d "backward": X, dY and dX once each, the rulebook twice (8 B per pair), the filters twice
        dense = (table >> 3).clamp_(min=-1) if packed_kl else table
        pairs = int((dense >= 0).sum().item())
        es = X.element_size()
        fused = Cin == 32 and Cout == 32 and _one_launch_rows(X.dtype) and K <= 27      # one launch
        if fused:
            ACCOUNT.append(dict(kind="conv_backward", pairs=pairs,
                                bytes=2 * R * Cin * es + dY.shape[0] * Cout * es + 2 * pairs * 8 + 2 * K * Cin * Cout * 4,
                                flops=4 * pairs * Cin * Cout))
        else:
            _account("gather_dw", dense, R, R, Cin, dY.shape[0], Cout, K, Cin, Cout, es)
            _account("gather_conv", dense, R, dY.shape[0], Cout, R, Cin, K, Cin, Cout, es)
    return dX, dW


# dW and dX of a 32 -> 32 layer in one launch (WFS_FUSED_CONV_BACKWARD=0: two launches): 16-bit rows, and fp32 rows on the
# three-piece kernels (the library's WFS_SPLIT_BF16, default on; WFS_FUSED_CONV_BACKWARD_F32=0 keeps fp32 on two launches)
FUSED_CONV_BACKWARD = __import__("os").environ.get("WFS_FUSED_CONV_BACKWARD", "1") != "0"
FUSED_CONV_BACKWARD_F32 = (__import__("os").environ.get("WFS_SPLIT_BF16", "1") != "0"
                           and __import__("os").environ.get("WFS_FUSED_CONV_BACKWARD_F32", "1") != "0")


def _one_launch_rows(dtype):
    return dtype in (torch.bfloat16, torch.float16) or (dtype == torch.float32 and FUSED_CONV_BACKWARD_F32)


class SparseConvFunction(Function):
    """features [n_in, Cin], filters [*k, Cin, Cout] fp32, bias [Cout] or None -> [n_out, Cout]."""

    @staticmethod
    def forward(ctx, features, filters, bias, rulebook, mode):
        rb = rulebook
        features = _features_ok(features)
        K = rb.K
        W = filters.detach().reshape(K, filters.shape[-2], filters.shape[-1]).float().contiguous()
        b = None if bias is None else bias.detach().float().contiguous()
        ident = rb.centre_k if rb.subm else -1
        w16 = None
        if mode == INVERSE:
            assert features.shape[0] == rb.M, "inverse conv input must be the coupled conv's output set"
            w16 = filters16(W, features) if takes_wide_path(K, rb.N, features, W.shape[2]) else None
            out = gather_conv(rb.nbr_out, None, K, ident, rb.N, features, W, False, b, rb.n_dev, w16)
        elif rb.has_dup:
            out = scatter_conv(rb.nbr_out, K, ident, rb.N, features, W, False, rb.M, b)
        else:
            assert features.shape[0] == rb.N
            table, kmap = rb.table_by_out()
            w16 = filters16(W, features) if takes_wide_path(K, rb.M, features, W.shape[2]) else None
            out = gather_conv(table, kmap, K, ident, rb.M, features, W, False, b, rb.m_dev, w16)
        ctx.save_for_backward(features, filters, bias)
        ctx.rb, ctx.mode = rb, mode
        ctx.w16 = w16 if ctx.needs_input_grad[0] else None       # the 16-bit filters of a wide layer: dX reads them again
        return out

    @staticmethod
    def backward(ctx, grad_output):
        features, filters, bias = ctx.saved_tensors
        rb, mode = ctx.rb, ctx.mode
        K = rb.K
        dY = grad_output.contiguous()
        if dY.dtype != features.dtype:
            dY = dY.to(features.dtype)
        W = filters.detach().reshape(K, filters.shape[-2], filters.shape[-1]).float().contiguous()
        ident = rb.centre_k if rb.subm else -1
        dX = dW = db = None
        from . import ops as _ops
        ov = _ops.OVERLAP_DW
        if mode == INVERSE:
            if ctx.needs_input_grad[0]:
                if rb.has_dup:
                    dX = scatter_conv(rb.nbr_out, K, ident, rb.N, dY, W, True, rb.M, None)
                else:
                    dX = gather_conv(rb.nbr_in, None, K, ident, rb.M, dY, W, True, None, rb.m_dev, ctx.w16)
            if ctx.needs_input_grad[1]:
                dW = gather_dw(rb.nbr_out, K, ident, rb.N, dY, features, True, None, rb.n_dev, ov, filters)
        elif (FUSED_CONV_BACKWARD and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and not ov and not rb.has_dup
              and features.shape[1] == 32 and dY.shape[1] == 32 and _one_launch_rows(features.dtype)
              and features.is_cuda):
            table, pk = rb.table_by_in(32, 32, features, 3)
            if pk and rb.table_by_in(32, 32, dY, 1)[1] != pk:
                table, pk = rb.nbr_out, 0
            dX, dW = conv_backward(table, K, ident, rb.N, features, dY, W, rb.n_dev, filters, pk)
        else:
            # dW first: with OVERLAP_DW it goes to the side stream and runs beside the dX launched next
            if ctx.needs_input_grad[1]:
                if features.shape[1] == 2 and dY.shape[1] == 32 and K <= 27 and not rb.has_dup:
                    # narrow input, wide output (first layer): keep the wide dY rows stationary
                    table, kmap = rb.table_by_out()
                    dW = gather_dw(table, K, ident, rb.M, dY, features, True, kmap, rb.m_dev, ov, filters)
                else:
                    table, pk = rb.table_by_in(features.shape[1], dY.shape[1], features, 3)
                    dW = gather_dw(table, K, ident, rb.N, features, dY, False, None, rb.n_dev, ov, filters, pk)
            if ctx.needs_input_grad[0]:
                table, pk = rb.table_by_in(dY.shape[1], W.shape[1], dY, 1)
                dX = gather_conv(table, None, K, ident, rb.N, dY, W, True, None, rb.n_dev, ctx.w16, pk)
        if dW is not None:
            dW = dW.reshape(filters.shape).to(filters.dtype)
        if bias is not None and ctx.needs_input_grad[2]:
            # dY has one row per OUTPUT of this product: rb.N rows for an inverse conv, rb.M otherwise
            db = _masked_column_sum(dY, rb.n_dev if mode == INVERSE else rb.m_dev).to(bias.dtype)
        return dX, dW, db, None, None


class PointwiseConvFunction(Function):
    """1 x 1 convolution with >= 128 channels on a side: Y = X . W (+ bias) on the matrix cores of
    csrc/wide.hip instead of spconv's ``torch.mm(features, weight.view(in, out))`` (spconv 1.2.1 conv.py; the hybrid
    net's 2048 -> 1697 layer, reference src/models/SPConvBlocks.py:498)."""

    @staticmethod
    def forward(ctx, features, filters, bias, n_dev):
        features = _features_ok(features)
        R = int(features.shape[0])
        W = filters.detach().reshape(1, filters.shape[-2], filters.shape[-1]).float().contiguous()
        b = None if bias is None else bias.detach().float().contiguous()
        w16 = filters16(W, features) if takes_wide_path(1, R, features, W.shape[2]) else None
        out = gather_conv(None, None, 1, 0, R, features, W, False, b, n_dev, w16)
        ctx.save_for_backward(features, filters, bias)
        ctx.n_dev = n_dev
        ctx.w16 = w16 if ctx.needs_input_grad[0] else None
        return out

    @staticmethod
    def backward(ctx, grad_output):
        features, filters, bias = ctx.saved_tensors
        n_dev = ctx.n_dev
        R = int(features.shape[0])
        dY = grad_output.contiguous()
        if dY.dtype != features.dtype:
            dY = dY.to(features.dtype)
        W = filters.detach().reshape(1, filters.shape[-2], filters.shape[-1]).float().contiguous()
        dX = dW = db = None
        if ctx.needs_input_grad[1]:
            dW = gather_dw(None, 1, 0, R, features, dY, False, None, n_dev, False, filters)
            dW = dW.reshape(filters.shape).to(filters.dtype)
        if ctx.needs_input_grad[0]:
            dX = gather_conv(None, None, 1, 0, R, dY, W, True, None, n_dev, ctx.w16)
        if bias is not None and ctx.needs_input_grad[2]:
            db = _masked_column_sum(dY, n_dev).to(bias.dtype)
        return dX, dW, db, None


def pointwise_conv(features, filters, bias, n_dev=None):
    return PointwiseConvFunction.apply(features, filters, bias, n_dev)


def _dense_map_ok(cell_map, spatial, batch_size, C, features):
    """Shapes wfs_to_dense_mapped covers, and the map must be of THIS dense shape."""
    v = 1
    for s_ in spatial:
        v *= int(s_)
    pack = 1 if features.dtype == torch.float32 else 2
    return (cell_map[3] == v and v % pack == 0 and C % 4 == 0 and 4 <= C <= 128 and 1 <= batch_size <= 65535
            and features.shape[0] > 0)


class ToDenseFunction(Function):
    """SparseConvTensor.dense(): [M, C] -> [B, C, *spatial] (A.1)."""

    @staticmethod
    def forward(ctx, features, indices, spatial_shape, batch_size, unique, m_dev=None, cell_map=None):
        lib = _lib.load()
        features = _features_ok(features)
        indices = indices.contiguous()
        M, C = features.shape
        ndim = indices.shape[1] - 1
        spatial = [int(s) for s in spatial_shape]
        ctx.cell_map = None
        if cell_map is not None and unique and _dense_map_ok(cell_map, spatial, int(batch_size), C, features):
            # the producing conv's cell -> row map: every cell written once, no zero fill, whole runs per channel
            ticket, slot, _keep, V = cell_map
            out = torch.empty([int(batch_size), C] + spatial, dtype=features.dtype, device=features.device)
            _lib.check(lib.wfs_to_dense_mapped(_lib.ptr(features), ticket, slot, M, _lib.ptr(m_dev), int(batch_size), V,
                                               C, _lib.ptr(out), _lib.dtype_code(features), _lib.stream_ptr()))
            ctx.cell_map = cell_map
            ctx.meta = (spatial, int(batch_size), M, C)
            ctx.m_dev = m_dev
            ctx.like = (features.dtype, features.device)
            return out
        out = torch.zeros([int(batch_size), C] + spatial, dtype=features.dtype, device=features.device)
        winner = None
        if not unique:
            cells = int(batch_size)
            for s in spatial:
                cells *= s
            winner = torch.empty((cells,), dtype=torch.int32, device=features.device)
        _lib.check(lib.wfs_to_dense(_lib.ptr(features), _lib.ptr(indices), M, ndim, _lib.i32_array(spatial),
                                    int(batch_size), C, _lib.ptr(out), _lib.ptr(winner), _lib.dtype_code(features),
                                    _lib.ptr(m_dev), _lib.stream_ptr()))
        ctx.save_for_backward(indices)
        ctx.meta = (spatial, int(batch_size), M, C)
        ctx.m_dev = m_dev
        return out

    @staticmethod
    def backward(ctx, grad_output):
        lib = _lib.load()
        spatial, batch_size, M, C = ctx.meta
        if ctx.cell_map is not None:
            ticket, slot, _keep, V = ctx.cell_map
            dtype, device = ctx.like
            dY = grad_output.contiguous()
            if dY.dtype != dtype:
                dY = dY.to(dtype)
            dX = torch.empty((M, C), dtype=dtype, device=device)     # every valid row has exactly one cell
            _lib.check(lib.wfs_to_dense_bwd_mapped(_lib.ptr(dY), ticket, slot, M, _lib.ptr(ctx.m_dev), batch_size, V, C,
                                                   _lib.ptr(dX), _lib.dtype_code(dX), _lib.stream_ptr()))
            return dX, None, None, None, None, None, None
        (indices,) = ctx.saved_tensors
        dY = grad_output.contiguous()
        dX = _rows((M, C), dY, ctx.m_dev)
        _lib.check(lib.wfs_to_dense_bwd(_lib.ptr(dY), _lib.ptr(indices), M, len(spatial), _lib.i32_array(spatial),
                                        batch_size, C, _lib.ptr(dX), _lib.dtype_code(dY), _lib.ptr(ctx.m_dev),
                                        _lib.stream_ptr()))
        return dX, None, None, None, None, None, None


class BatchNormReLUFunction(Function):
    """nn.BatchNorm1d (+ nn.ReLU) over the active rows [N, C] in two launches per direction
    (reference: the plain modules inside spconv.SparseSequential, src/models/SPConvBlocks.py:505-508)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, training, relu, n_dev=None,
                batches_tracked=None):
        lib = _lib.load()
        x = _features_ok(x)
        N, C = x.shape
        y = _rows(tuple(x.shape), x, n_dev)
        for t in (weight, bias, running_mean, running_var):
            assert t is None or (t.dtype == torch.float32 and t.numel() == C and t.is_contiguous())
        save_mean = torch.empty((C,), dtype=torch.float32, device=x.device)
        save_invstd = torch.empty((C,), dtype=torch.float32, device=x.device)
        ws = torch.empty((max(int(lib.wfs_bn_workspace_bytes(N, C)), 1),), dtype=torch.uint8, device=x.device)
        _lib.check(lib.wfs_bn_relu_fwd(_lib.ptr(x), N, C, _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(running_mean),
                                       _lib.ptr(running_var), _lib.ptr(batches_tracked) if training else None,
                                       float(momentum), float(eps), 1 if training else 0,
                                       1 if relu else 0, _lib.ptr(y), _lib.ptr(save_mean), _lib.ptr(save_invstd),
                                       _lib.ptr(ws), ws.numel(), _lib.dtype_code(x), _lib.ptr(n_dev),
                                       _lib.stream_ptr()))
        ctx.save_for_backward(x, weight, bias, save_mean, save_invstd)
        ctx.flags = (bool(training), bool(relu))
        ctx.n_dev = n_dev
        return y

    @staticmethod
    def backward(ctx, grad_output):
        x, weight, bias, save_mean, save_invstd = ctx.saved_tensors
        training, relu = ctx.flags
        dx, dgamma, dbeta = bn_relu_backward(x, grad_output, weight, bias, save_mean, save_invstd, training, relu, ctx.n_dev)
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None


def bn_relu_backward(x, grad_output, weight, bias, save_mean, save_invstd, training, relu, n_dev):
    """(dx, dgamma, dbeta) of y = [relu](BatchNorm1d(x)) over the active rows, given dL/dy (two launches: the sums
    sum(g), sum(g * xhat) with the ReLU mask recomputed from x, then the elementwise pass)."""
    lib = _lib.load()
    N, C = x.shape
    dy = grad_output.contiguous()
    if dy.dtype != x.dtype:
        dy = dy.to(x.dtype)
    dx = _rows(tuple(x.shape), x, n_dev)
    dgamma = grad_like(weight) if weight is not None else None
    dbeta = grad_like(bias) if bias is not None else None
    ws = torch.empty((max(int(lib.wfs_bn_workspace_bytes(N, C)), 1),), dtype=torch.uint8, device=x.device)
    _lib.check(lib.wfs_bn_relu_bwd(_lib.ptr(x), _lib.ptr(dy), N, C, _lib.ptr(weight), _lib.ptr(bias),
                                   _lib.ptr(save_mean), _lib.ptr(save_invstd), 1 if training else 0,
                                   1 if relu else 0, _lib.ptr(dx), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(ws),
                                   ws.numel(), _lib.dtype_code(x), _lib.ptr(n_dev), _lib.stream_ptr()))
    return dx, dgamma, dbeta


def batch_norm_relu(features, bn, relu, n_dev=None):
    """Apply an nn.BatchNorm1d module (and optionally the nn.ReLU that follows it) to [N, C] features
    (``n_dev``: device-side count of valid rows, see include/wfsparse.h "device-side row counts")."""
    training = bn.training or (bn.running_mean is None and bn.running_var is None)
    tracked = bn.num_batches_tracked if (bn.training and bn.track_running_stats) else None   # bumped by the kernel
    return BatchNormReLUFunction.apply(
        features, bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
        bn.running_var if bn.track_running_stats else None, bn.momentum, bn.eps, training, relu, n_dev, tracked)


def can_fuse_batch_norm(bn, features):
    """The fused kernels cover what the reference's nets use: float32 parameters, a fixed momentum,
    fp32 / bf16 / fp16 features on the GPU, any channel count (beyond what one block covers -- 1024 channels, or 256
    when C is not a multiple of 4; the hybrid net starts at 2 T = 2048 -- the library runs channel slices)."""
    c = bn.num_features
    return (type(bn) is torch.nn.BatchNorm1d and bn.momentum is not None and features.is_cuda and features.dim() == 2
            and features.dtype in (torch.float32, torch.bfloat16, torch.float16) and features.shape[0] > 0
            and 1 <= c <= 65536
            and (bn.weight is None or bn.weight.dtype == torch.float32)
            and (bn.training or bn.running_mean is not None))


class SkinnyLinearFunction(Function):
    """nn.Linear with a handful of outputs over a long input row (the PSD head: [B, 35840] -> [B, 3]):
    Y = X W^T + b with X fp32/bf16, W/b/Y fp32 (reference src/models/SPConvNet.py:67-68)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        x = _features_ok(x)
        B, I = x.shape
        O = weight.shape[0]
        w = weight.detach().float().contiguous()
        b = None if bias is None else bias.detach().float().contiguous()
        y = torch.empty((B, O), dtype=torch.float32, device=x.device)
        _lib.check(lib.wfs_head_fwd(_lib.ptr(x), B, I, _lib.ptr(w), _lib.ptr(b), O, _lib.ptr(y), _lib.dtype_code(x),
                                    _lib.stream_ptr()))
        ctx.save_for_backward(x, weight, bias)
        return y

    @staticmethod
    def backward(ctx, grad_output):
        lib = _lib.load()
        x, weight, bias = ctx.saved_tensors
        B, I = x.shape
        O = weight.shape[0]
        g = grad_output.float().contiguous()
        w = weight.detach().float().contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = grad_like(weight, (O, I)) if ctx.needs_input_grad[1] else None
        ws = torch.empty((max(int(lib.wfs_head_workspace_bytes(B, I, O)), 1),), dtype=torch.uint8, device=x.device)
        want_db = bias is not None and ctx.needs_input_grad[2]
        db = grad_like(bias, (O,)) if (want_db and dw is not None) else None
        defer = _DEFERRED_DW is not None and dw is not None and dw._base is not None and dx is not None
        job = _lib.DwJob() if defer else None
        _lib.check(lib.wfs_head_bwd(_lib.ptr(x), _lib.ptr(g), B, I, _lib.ptr(w), O, _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db),
                                    _lib.dtype_code(x), _lib.ptr(ws), ws.numel(), ctypes.byref(job) if defer else None,
                                    _lib.stream_ptr()))
        if defer and job.nslabs > 0:
            _DEFERRED_DW.append((job, ws))          # the sum over the dW partials joins the pass's deferred reductions
        if want_db and db is None:
            db = g.sum(0)
        return dx, (dw.to(weight.dtype) if dw is not None else None), (db.to(bias.dtype) if db is not None else None)


class SparseHeadFunction(Function):
    """ToDense -> view(-1, n_linear) -> nn.Linear(n_linear, n_type) straight off the sparse rows (csrc/shead.hip;
    include/wfsparse.h wfs_sparse_head_fwd): features [M, C] of the last conv, its cell -> row map, fp32 weight
    [O, C * V] / bias [O] of the Linear -> logits [batch, O] fp32.  The dense [batch, C, *spatial] tensor never exists
    (reference src/models/SPConvNet.py:65-68)."""

    @staticmethod
    def forward(ctx, features, weight, bias, cell_map, batch_size, m_dev):
        lib = _lib.load()
        features = _features_ok(features)
        M, C = features.shape
        ticket, slot, _keep, V = cell_map
        O = int(weight.shape[0])
        w = weight.detach().float().contiguous()
        b = None if bias is None else bias.detach().float().contiguous()
        y = torch.empty((int(batch_size), O), dtype=torch.float32, device=features.device)
        ws = torch.empty((int(lib.wfs_sparse_head_workspace_bytes(int(batch_size), V, C, O)),), dtype=torch.uint8,
                         device=features.device)
        _lib.check(lib.wfs_sparse_head_fwd(_lib.ptr(features), ticket, slot, M, _lib.ptr(m_dev), int(batch_size), V, C,
                                           _lib.ptr(w), _lib.ptr(b), O, _lib.ptr(y), _lib.dtype_code(features),
                                           _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
        ctx.save_for_backward(features, weight, bias)
        ctx.cell_map, ctx.batch_size, ctx.m_dev = cell_map, int(batch_size), m_dev
        return y

    @staticmethod
    def backward(ctx, grad_output):
        lib = _lib.load()
        features, weight, bias = ctx.saved_tensors
        M, C = features.shape
        ticket, slot, _keep, V = ctx.cell_map
        O = int(weight.shape[0])
        g = grad_output.float().contiguous()
        w = weight.detach().float().contiguous()
        dx = _rows((M, C), features, ctx.m_dev) if ctx.needs_input_grad[0] else None
        dw = grad_like(weight, (O, C * V)) if ctx.needs_input_grad[1] else None
        want_db = bias is not None and ctx.needs_input_grad[2]
        db = grad_like(bias, (O,)) if (want_db and dw is not None) else None
        ws = torch.empty((int(lib.wfs_sparse_head_workspace_bytes(ctx.batch_size, V, C, O)),), dtype=torch.uint8,
                         device=features.device)
        defer = _DEFERRED_DW is not None and dw is not None and dw._base is not None
        job = _lib.DwJob() if defer else None
        _lib.check(lib.wfs_sparse_head_bwd(_lib.ptr(features), _lib.ptr(g), ticket, slot, M, _lib.ptr(ctx.m_dev),
                                           ctx.batch_size, V, C, _lib.ptr(w), O, _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db),
                                           _lib.dtype_code(features), _lib.ptr(ws), ws.numel(),
                                           ctypes.byref(job) if defer else None, _lib.stream_ptr()))
        if defer and job.nslabs > 0:
            _DEFERRED_DW.append((job, ws))          # the sum over the per-slice dW partials joins the deferred reductions
        if want_db and db is None:
            db = g.sum(0)
        return (dx, (dw.to(weight.dtype) if dw is not None else None), (db.to(bias.dtype) if db is not None else None),
                None, None, None)


# WFS_SPARSE_HEAD=0: the dense route (dense() + the streaming head kernels)
SPARSE_HEAD = __import__("os").environ.get("WFS_SPARSE_HEAD", "1") != "0"


def can_use_sparse_head(x, layers):
    """x: the SparseConvTensor in front of a trailing ToDense; layers: the Linear stack behind the flatten."""
    if not SPARSE_HEAD or len(layers) != 1 or type(layers[0]) is not torch.nn.Linear:
        return False
    lin, f = layers[0], x.features
    cm = getattr(x, "cell_map", None)
    if cm is None or not f.is_cuda or f.dim() != 2 or f.shape[0] == 0 or lin.weight.dtype != torch.float32:
        return False
    if not (x.unique is True or x.n_valid is not None):
        return False
    V = 1
    for s_ in x.spatial_shape:
        V *= int(s_)
    C = int(f.shape[1])
    return bool(cm[3] == V and lin.in_features == C * V and f.dtype in (torch.float32, torch.bfloat16, torch.float16)
                and _lib.load().wfs_sparse_head_ok(int(x.batch_size), V, C, int(lin.out_features), _lib.dtype_code(f)))


def sparse_head(x, linear):
    return SparseHeadFunction.apply(x.features, linear.weight, linear.bias, x.cell_map, x.batch_size, x.n_valid)


def can_use_skinny_linear(linear, x):
    # long rows (I % 8 == 0, >= 1024: streamed) or short ones of any length (<= 4096: scalar kernels)
    return (type(linear) is torch.nn.Linear and x.is_cuda and x.dim() == 2 and linear.out_features <= 8
            and ((x.shape[1] % 8 == 0 and x.shape[1] >= 1024) or x.shape[1] <= 4096)
            and x.dtype in (torch.float32, torch.bfloat16, torch.float16) and linear.weight.dtype == torch.float32)


def skinny_linear(x, linear):
    return SkinnyLinearFunction.apply(x, linear.weight, linear.bias)


class WideLinearFunction(Function):
    """nn.Linear with many outputs on activations [B, I]: y = x W^T + b on the matrix cores (csrc/wide.hip;
    include/wfsparse.h wfs_wide_linear_fwd): 16-bit activations with the fp32 weights rounded to their type as
    operands, or fp32 activations on the exact-fp32 MFMA; fp32 accumulate, fp32 result -- the hybrid net's
    Linear(24150, 269) (reference src/models/SPConvNet.py:40-52)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        x = _features_ok(x)
        B, I = x.shape
        O = weight.shape[0]
        w = weight.detach().float().contiguous()
        b = None if bias is None else bias.detach().float().contiguous()
        y = torch.empty((B, O), dtype=torch.float32, device=x.device)
        ws = torch.empty((int(lib.wfs_wide_linear_workspace_bytes(B, I, O, _lib.dtype_code(x))),), dtype=torch.uint8, device=x.device)
        _lib.check(lib.wfs_wide_linear_fwd(_lib.ptr(x), B, I, _lib.ptr(w), _lib.ptr(b), O, _lib.ptr(y), _lib.dtype_code(x),
                                        _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
        ctx.save_for_backward(x, weight, bias)
        return y

    @staticmethod
    def backward(ctx, grad_output):
        lib = _lib.load()
        x, weight, bias = ctx.saved_tensors
        B, I = x.shape
        O = weight.shape[0]
        g = grad_output.float().contiguous()
        w = weight.detach().float().contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = grad_like(weight, (O, I)) if ctx.needs_input_grad[1] else None
        db = _masked_column_sum(g, None) if (bias is not None and ctx.needs_input_grad[2]) else None
        ws = torch.empty((int(lib.wfs_wide_linear_workspace_bytes(B, I, O, _lib.dtype_code(x))),), dtype=torch.uint8, device=x.device)
        _lib.check(lib.wfs_wide_linear_bwd(_lib.ptr(x), _lib.ptr(g), B, I, _lib.ptr(w), O, _lib.ptr(dx), _lib.ptr(dw),
                                        None, _lib.dtype_code(x), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
        return dx, (dw.to(weight.dtype) if dw is not None else None), (db.to(bias.dtype) if db is not None else None)


def can_use_wide_linear(linear, x):
    return bool(type(linear) is torch.nn.Linear and x.is_cuda and x.dim() == 2 and x.shape[0] > 0
                and x.dtype in (torch.float32, torch.bfloat16, torch.float16) and linear.weight.dtype == torch.float32
                and _lib.load().wfs_wide_linear_ok(x.shape[0], x.shape[1], linear.out_features, _lib.dtype_code(x)))


def wide_linear(x, linear):
    return WideLinearFunction.apply(x, linear.weight, linear.bias)


def head_forward(x, layers):
    """The dense head (reference src/models/SPConvNet.py:40-52: a Sequential of nn.Linear built by LinearBlock) on
    the flattened ToDense output: few-output layers on streamed 16-bit / fp32 rows (skinny_linear), wide layers of
    16-bit rows on the matrix cores (wide_linear), anything else as the torch module in the head's own dtype."""
    for layer in layers:
        if can_use_skinny_linear(layer, x):
            x = skinny_linear(x, layer)
        elif can_use_wide_linear(layer, x):
            x = wide_linear(x, layer)
        else:
            p = next(layer.parameters(), None)
            if p is not None and x.dtype != p.dtype:          # 16-bit activations, fp32 master weights
                x = x.to(p.dtype)
            x = layer(x)
    return x


class CrossEntropyMeanFunction(Function):
    """nn.CrossEntropyLoss(reduction='mean') on [B, C] fp32 logits: loss and d loss / d logits from one launch
    (reference criterion: src/engineering/LitBase.py:38-43, applied at LitPSD.py:102)."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        lib = _lib.load()
        logits = _features_ok(logits)
        target = target.contiguous()
        B, C = logits.shape
        loss = torch.empty((1,), dtype=torch.float32, device=logits.device)
        dlogits = torch.empty_like(logits) if ctx.needs_input_grad[0] else None
        _lib.check(lib.wfs_xent_mean_fwd_bwd(_lib.ptr(logits), _lib.ptr(target), B, C, int(ignore_index),
                                             _lib.ptr(loss), _lib.ptr(dlogits), _lib.stream_ptr()))
        ctx.dlogits = dlogits
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_output):
        unit = _UNIT_LOSS_GRAD.get(grad_output.device)
        if unit is not None and grad_output.data_ptr() == unit.data_ptr():
            return ctx.dlogits, None, None            # d loss / d loss = 1, handed in by unit_loss_grad(): no multiply
        return ctx.dlogits * grad_output, None, None


# ``loss.backward()`` makes autograd fill a fresh ones tensor for d loss / d loss (a launch) and the loss node multiply
# by it (another).  A runner that owns the backward call passes this persistent tensor instead:
#     torch.autograd.backward(loss, grad_tensors=unit_loss_grad(loss.device))
# and the fused cross-entropy recognises it (by address) and returns its gradient unscaled.
_UNIT_LOSS_GRAD = {}


def unit_loss_grad(device):
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if device not in _UNIT_LOSS_GRAD:
        _UNIT_LOSS_GRAD[device] = torch.ones((), dtype=torch.float32, device=device)
    return _UNIT_LOSS_GRAD[device]


def can_fuse_cross_entropy(criterion, logits, target):
    return (type(criterion) is torch.nn.CrossEntropyLoss and criterion.reduction == "mean" and criterion.weight is None
            and getattr(criterion, "label_smoothing", 0.0) == 0.0 and logits.is_cuda and logits.dim() == 2
            and logits.dtype == torch.float32 and target.dtype == torch.int64 and target.dim() == 1
            and logits.shape[0] >= 1 and logits.shape[1] <= 4096)


def cross_entropy_mean(logits, target, ignore_index=-100):
    return CrossEntropyMeanFunction.apply(logits, target, ignore_index)


# 16-bit storage: bf16, and fp16 for the reference's ``half_precision`` / ``use_half`` (float16 features,
# src/datasets/HDF5Dataset.py:227-228).  Both have native kernels (fp32 accumulate, fp32 master filters).
def indice_conv(features, filters, bias, rulebook):
    return SparseConvFunction.apply(features, filters, bias, rulebook, CONV)


def indice_subm_conv(features, filters, bias, rulebook):
    return SparseConvFunction.apply(features, filters, bias, rulebook, SUBM)


def indice_inverse_conv(features, filters, bias, rulebook):
    return SparseConvFunction.apply(features, filters, bias, rulebook, INVERSE)


def to_dense(features, indices, spatial_shape, batch_size, unique, m_dev=None, cell_map=None):
    return ToDenseFunction.apply(features, indices, spatial_shape, batch_size, unique, m_dev, cell_map)

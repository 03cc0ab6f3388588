"""oracle/ref.py -- TEST INFRASTRUCTURE ONLY (numpy/ctypes front end of oracle/spconv_ref.c).

CPU restatement of spconv 1.2.1's ``ops.get_indice_pairs`` / ``indice_conv`` /
``indice_conv_backward`` / ``SparseConvTensor.dense`` as SURVEY.md Appendix A.2-A.4, A.1 restate
them (spconv itself is absent from /root/reference: requirements.txt:15).  PARITY UNPINNED
against the upstream binary -- see the header of spconv_ref.c.

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile oracle/spconv_ref.c -> oracle/libwfo.so with gcc (seconds)."""
    so = os.path.join(_HERE, "libwfo.so")
    src = os.path.join(_HERE, "spconv_ref.c")
    if force or not os.path.exists(so) or (
            os.path.exists(src) and os.path.getmtime(so) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "libwfo.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        i32p = ctypes.POINTER(ctypes.c_int32)
        f32p = ctypes.POINTER(ctypes.c_float)
        L.wfo_rulebook_subm.restype = ctypes.c_int
        L.wfo_rulebook_subm.argtypes = [i32p, ctypes.c_int64, ctypes.c_int, i32p, i32p, i32p, i32p, i32p]
        L.wfo_rulebook_conv.restype = ctypes.c_int64
        L.wfo_rulebook_conv.argtypes = [i32p, ctypes.c_int64, ctypes.c_int, i32p, i32p, i32p, i32p,
                                        i32p, i32p, i32p, i32p]
        L.wfo_rulebook_conv_transpose.restype = ctypes.c_int64
        L.wfo_rulebook_conv_transpose.argtypes = L.wfo_rulebook_conv.argtypes
        L.wfo_indice_conv_fwd.restype = ctypes.c_int
        L.wfo_indice_conv_fwd.argtypes = [f32p, f32p, i32p, i32p, ctypes.c_int64, ctypes.c_int64,
                                          ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, f32p]
        L.wfo_indice_conv_bwd.restype = ctypes.c_int
        L.wfo_indice_conv_bwd.argtypes = [f32p, f32p, f32p, i32p, i32p, ctypes.c_int64,
                                          ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          f32p, f32p]
        L.wfo_to_dense.restype = ctypes.c_int
        L.wfo_to_dense.argtypes = [f32p, i32p, ctypes.c_int64, ctypes.c_int, i32p, ctypes.c_int,
                                   ctypes.c_int, f32p]
        _LIB = L
    return _LIB


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def _listify(v, ndim):
    if isinstance(v, (list, tuple, np.ndarray)):
        v = [int(x) for x in v]
        assert len(v) == ndim
        return v
    return [int(v)] * ndim


def conv_output_shape(spatial, ksize, stride, padding, dilation):
    """A.1: out = (i + 2p - d(k-1) - 1)//s + 1 per dim."""
    out = []
    for i, k, s, p, d in zip(spatial, ksize, stride, padding, dilation):
        out.append((int(i) + 2 * p - d * (k - 1) - 1) // s + 1)
    return out


def deconv_output_shape(spatial, ksize, stride, padding, dilation, out_padding):
    """A.1: out = (i-1)s - 2p + k + op per dim."""
    out = []
    for i, k, s, p, d, op in zip(spatial, ksize, stride, padding, dilation, out_padding):
        out.append((int(i) - 1) * s - 2 * p + k + op)
    return out


def get_indice_pairs(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1,
                     out_padding=0, subm=False, transpose=False):
    """A.2 front door.  Returns (out_indices [M,D+1], indice_pairs [2,K,N], indice_pair_num [K])."""
    indices = _i32(indices)
    N, ndim = indices.shape[0], indices.shape[1] - 1
    ksize, stride = _listify(ksize, ndim), _listify(stride, ndim)
    padding, dilation = _listify(padding, ndim), _listify(dilation, ndim)
    out_padding = _listify(out_padding, ndim)
    spatial_shape = [int(s) for s in spatial_shape]
    for d, s in zip(dilation, stride):
        assert s == 1 or d == 1, "don't support this."
    if transpose:
        assert not subm
        out_shape = deconv_output_shape(spatial_shape, ksize, stride, padding, dilation, out_padding)
    elif subm:
        out_shape = list(spatial_shape)
    else:
        out_shape = conv_output_shape(spatial_shape, ksize, stride, padding, dilation)
    vol = 1
    for s in out_shape:
        vol *= s
    if int(batch_size) * vol >= 2 ** 31:
        raise RuntimeError("batch_size * prod(out_shape) must be < 2^31")
    K = int(np.prod(ksize))
    pairs = np.full((2, K, N), -1, dtype=np.int32)
    num = np.zeros((K,), dtype=np.int32)
    L = lib()
    i32 = ctypes.c_int32
    if subm:
        rc = L.wfo_rulebook_subm(_p(indices, i32), N, ndim, _p(_i32(out_shape), i32),
                                 _p(_i32(ksize), i32), _p(_i32(dilation), i32), _p(pairs, i32),
                                 _p(num, i32))
        if rc != 0:
            raise RuntimeError("wfo_rulebook_subm failed: %d" % rc)
        return indices, pairs, num
    out_idx = np.zeros((max(N * K, 1), ndim + 1), dtype=np.int32)
    fn = L.wfo_rulebook_conv_transpose if transpose else L.wfo_rulebook_conv
    M = fn(_p(indices, i32), N, ndim, _p(_i32(out_shape), i32),
                            _p(_i32(ksize), i32), _p(_i32(stride), i32), _p(_i32(padding), i32),
                            _p(_i32(dilation), i32), _p(out_idx, i32), _p(pairs, i32), _p(num, i32))
    if M < 0:
        raise RuntimeError("wfo_rulebook_conv failed: %d" % M)
    return out_idx[:M].copy(), pairs, num


def indice_conv(features, filters, pairs, pair_num, num_act_out, inverse=False, subm=False):
    """A.4 forward.  filters [*k, Cin, Cout] (any leading kernel dims)."""
    features, filters = _f32(features), _f32(filters)
    Cin, Cout = filters.shape[-2], filters.shape[-1]
    K = pairs.shape[1]
    filters = filters.reshape(K, Cin, Cout)
    out = np.zeros((int(num_act_out), Cout), dtype=np.float32)
    pairs, pair_num = _i32(pairs), _i32(pair_num)
    f32, i32 = ctypes.c_float, ctypes.c_int32
    lib().wfo_indice_conv_fwd(_p(features, f32), _p(filters, f32), _p(pairs, i32), _p(pair_num, i32),
                              features.shape[0], int(num_act_out), K, pairs.shape[2], Cin, Cout,
                              int(bool(inverse)), int(bool(subm)), _p(out, f32))
    return out


def indice_conv_backward(features, filters, out_bp, pairs, pair_num, inverse=False, subm=False):
    """A.4 backward.  Returns (d_features, d_filters [same shape as filters])."""
    features, filters, out_bp = _f32(features), _f32(filters), _f32(out_bp)
    Cin, Cout = filters.shape[-2], filters.shape[-1]
    K = pairs.shape[1]
    fshape = filters.shape
    filters = filters.reshape(K, Cin, Cout)
    din = np.zeros_like(features)
    dfil = np.zeros_like(filters)
    pairs, pair_num = _i32(pairs), _i32(pair_num)
    f32, i32 = ctypes.c_float, ctypes.c_int32
    lib().wfo_indice_conv_bwd(_p(features, f32), _p(filters, f32), _p(out_bp, f32), _p(pairs, i32),
                              _p(pair_num, i32), features.shape[0], out_bp.shape[0], K,
                              pairs.shape[2], Cin, Cout, int(bool(inverse)), int(bool(subm)),
                              _p(din, f32), _p(dfil, f32))
    return din, dfil.reshape(fshape)


def to_dense(features, indices, spatial_shape, batch_size):
    """A.1 SparseConvTensor.dense(channels_first=True) -> [B, C, *spatial]."""
    features, indices = _f32(features), _i32(indices)
    ndim = indices.shape[1] - 1
    spatial_shape = [int(s) for s in spatial_shape]
    C = features.shape[1]
    out = np.zeros([int(batch_size), C] + spatial_shape, dtype=np.float32)
    f32, i32 = ctypes.c_float, ctypes.c_int32
    lib().wfo_to_dense(_p(features, f32), _p(indices, i32), indices.shape[0], ndim,
                       _p(_i32(spatial_shape), i32), int(batch_size), C, _p(out, f32))
    return out

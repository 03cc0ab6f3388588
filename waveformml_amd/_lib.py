"""ctypes binding of libwfsparse.so (include/wfsparse.h).  No CPU fallback: if the HIP library is
missing or a tensor is not on the GPU, calls fail loudly."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# WFS_LIB: another build of the same library (A/B timing of kernel variants, tools/ab_bench.sh)
LIB_PATH = os.environ.get("WFS_LIB") or os.path.join(_HERE, "lib", "libwfsparse.so")

WFS_OK, WFS_EINVAL, WFS_EOVERFLOW, WFS_EHIP, WFS_EWORKSPACE = 0, 1, 2, 3, 4
WFS_F32, WFS_BF16, WFS_F16 = 0, 1, 2
WFS_MAX_DIM = 4
WFS_ABI_VERSION = 6         # include/wfsparse.h: this binding's struct layouts and signatures
TIMER_GATHER_CONV, TIMER_GATHER_DW, TIMER_RULEBOOK, TIMER_CONV_BACKWARD = 0, 1, 2, 3

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
_vp, _i32, _i64, _sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t


class Geometry(ctypes.Structure):
    """struct wfs_geometry"""
    _fields_ = [("ndim", _i32), ("batch_size", _i32), ("subm", _i32), ("K", _i32),
                ("spatial", _i32 * WFS_MAX_DIM), ("out_shape", _i32 * WFS_MAX_DIM),
                ("ksize", _i32 * WFS_MAX_DIM), ("stride", _i32 * WFS_MAX_DIM),
                ("padding", _i32 * WFS_MAX_DIM), ("dilation", _i32 * WFS_MAX_DIM),
                ("transposed", _i32), ("output_padding", _i32 * WFS_MAX_DIM)]


class DwJob(ctypes.Structure):
    """struct wfs_dw_job"""
    _fields_ = [("part", _vp), ("nslabs", _i64), ("per", _i64), ("K", _i32), ("A", _i32), ("B", _i32),
                ("transpose", _i32), ("dW", _vp)]


# name -> (restype, argtypes); mirrors include/wfsparse.h one to one
SIGNATURES = {
    "wfs_abi_version": (ctypes.c_int, []),
    "wfs_last_error": (ctypes.c_char_p, []),
    "wfs_geometry_init": (ctypes.c_int, [ctypes.POINTER(Geometry)]),
    "wfs_rulebook_workspace_bytes": (_sz, [ctypes.POINTER(Geometry), _i64]),
    "wfs_rulebook_plan": (ctypes.c_int, [ctypes.POINTER(Geometry), _vp, _i64, _vp, _vp, _sz, c_i64p, _vp, _vp, _i64,
                                         _vp]),
    "wfs_rulebook_emit": (ctypes.c_int, [ctypes.POINTER(Geometry), _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp,
                                         _vp, _sz, _vp, _vp, _vp]),
    "wfs_indices_check": (ctypes.c_int, [ctypes.POINTER(Geometry), _vp, _i64, _vp, _sz, c_i64p, _vp]),
    "wfs_gather_conv": (ctypes.c_int, [_vp, c_i32p, _i32, _i32, _i64, _vp, _i64, _i32, _vp, _i32, _i32, _i32,
                                       _vp, _vp, _i32, _vp, _i32, _vp]),
    "wfs_gather_packed_ok": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "wfs_unpack_table": (ctypes.c_int, [_vp, _i32, _i32, _i64, _vp, _vp, _vp]),
    "wfs_event_rulebook_conv_ok": (ctypes.c_int, [ctypes.POINTER(Geometry)]),
    "wfs_event_rulebook_conv_packed_kl": (ctypes.c_int, [ctypes.POINTER(Geometry)]),
    "wfs_event_rulebook_conv_state_bytes": (_sz, [_i32]),
    "wfs_event_rulebook_conv": (ctypes.c_int, [ctypes.POINTER(Geometry), _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp,
                                               _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "wfs_wide_conv_ok": (ctypes.c_int, [_i32, _i64, _i64, _i32, _i32, _i32]),
    "wfs_wide_enable": (ctypes.c_int, [_i32]),
    "wfs_wide_conv_workspace_bytes": (_sz, [_i32, _i64, _i64, _i32, _i32, _i32, _i32]),
    "wfs_wide_filters_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "wfs_wide_filters": (ctypes.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "wfs_wide_gather_conv": (ctypes.c_int, [_vp, c_i32p, _i32, _i32, _i64, _vp, _i64, _i32, _vp, _vp, _i32, _i32, _i32,
                                            _vp, _vp, _i32, _vp, _vp, _sz, _vp]),
    "wfs_wide_linear_ok": (ctypes.c_int, [_i64, _i32, _i32, _i32]),
    "wfs_wide_linear_workspace_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "wfs_wide_linear_fwd": (ctypes.c_int, [_vp, _i64, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _sz, _vp]),
    "wfs_wide_linear_bwd": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _sz, _vp]),
    "wfs_event_offsets_ints": (_sz, [_i32]),
    "wfs_event_offsets": (ctypes.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "wfs_event_rulebook_ok": (ctypes.c_int, [ctypes.POINTER(Geometry)]),
    "wfs_event_rulebook_flag_ints": (_sz, [_i32]),
    "wfs_event_rulebook_subm": (ctypes.c_int, [ctypes.POINTER(Geometry), _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "wfs_gather_dw_workspace_bytes": (_sz, [_i32, _i64, _i32, _i32]),
    "wfs_gather_dw": (ctypes.c_int, [_vp, c_i32p, _i32, _i32, _i64, _vp, _i32, _vp, _i64, _i32, _i32, _vp, _i32, _vp,
                                     _sz, _vp, ctypes.POINTER(DwJob), _i32, _vp]),
    "wfs_conv_backward": (ctypes.c_int, [_vp, _i32, _i32, _i64, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _sz,
                                         _vp, ctypes.POINTER(DwJob), _i32, _vp]),
    "wfs_dw_reduce_jobs": (ctypes.c_int, [ctypes.POINTER(DwJob), _i32, _vp]),
    "wfs_scatter_conv": (ctypes.c_int, [_vp, _i32, _i32, _i64, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _i32, _vp]),
    "wfs_bn_workspace_bytes": (_sz, [_i64, _i32]),
    "wfs_column_sum_workspace_bytes": (_sz, [_i32]),
    "wfs_column_sum": (ctypes.c_int, [_vp, _i64, _i32, _vp, _vp, _sz, _i32, _vp, _vp]),
    "wfs_bn_relu_fwd": (ctypes.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_float, _i32,
                                       _i32, _vp, _vp, _vp, _vp, _sz, _i32, _vp, _vp]),
    "wfs_bn_relu_bwd": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _sz,
                                       _i32, _vp, _vp]),
    "wfs_rulebook_cell_map": (ctypes.c_int, [ctypes.POINTER(Geometry), _i64, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                             c_i64p]),
    "wfs_to_dense_mapped": (ctypes.c_int, [_vp, _vp, _vp, _i64, _vp, _i32, _i64, _i32, _vp, _i32, _vp]),
    "wfs_to_dense_bwd_mapped": (ctypes.c_int, [_vp, _vp, _vp, _i64, _vp, _i32, _i64, _i32, _vp, _i32, _vp]),
    "wfs_to_dense": (ctypes.c_int, [_vp, _vp, _i64, _i32, c_i32p, _i32, _i32, _vp, _vp, _i32, _vp, _vp]),
    "wfs_to_dense_bwd": (ctypes.c_int, [_vp, _vp, _i64, _i32, c_i32p, _i32, _i32, _vp, _i32, _vp, _vp]),
    "wfs_sparse_head_ok": (ctypes.c_int, [_i32, _i64, _i32, _i32, _i32]),
    "wfs_sparse_head_workspace_bytes": (_sz, [_i32, _i64, _i32, _i32]),
    "wfs_sparse_head_fwd": (ctypes.c_int, [_vp, _vp, _vp, _i64, _vp, _i32, _i64, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _sz,
                                           _vp]),
    "wfs_sparse_head_bwd": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i32, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _i32,
                                           _vp, _sz, ctypes.POINTER(DwJob), _vp]),
    "wfs_head_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "wfs_head_fwd": (ctypes.c_int, [_vp, _i64, _i64, _vp, _vp, _i32, _vp, _i32, _vp]),
    "wfs_head_bwd": (ctypes.c_int, [_vp, _vp, _i64, _i64, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _sz, ctypes.POINTER(DwJob),
                                    _vp]),
    "wfs_tcn_lds_bytes": (_sz, [_i32, _i32, _i32]),
    "wfs_tcn_taps_fwd": (ctypes.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "wfs_tcn_taps_bwd": (ctypes.c_int, [_vp, _i32, _i32, _vp, _i64, _vp]),
    "wfs_tcn_fwd": (ctypes.c_int, [_vp, _i64, _i32, _vp, _vp, _i32, _i32, _vp, _i32, ctypes.c_float, _vp, _vp]),
    "wfs_tcn_bwd": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _i32, ctypes.c_float, _vp,
                                    _vp]),
    "wfs_xent_mean_fwd_bwd": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp]),
    "wfs_sgd_step": (ctypes.c_int, [_vp, _vp, _vp, _i64, _vp, ctypes.c_float, ctypes.c_float, ctypes.c_float, _i32, _i32,
                                    _vp]),
    "wfs_load_batch": (ctypes.c_int, [_vp, _i64, _i32, c_i32p, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i32,
                                      _vp]),
    "wfs_timing_enable": (ctypes.c_int, [_i32]),
    "wfs_timing_read": (ctypes.c_int, [_i32, ctypes.POINTER(ctypes.c_double), c_i64p]),
}

_LIB = None


def load():
    """Load libwfsparse.so (built by waveformml_amd/csrc/Makefile or __graft_entry__.build())."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libwfsparse.so is missing at %s -- build it with `make -C waveformml_amd/csrc` "
                "(or python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)        # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        got = lib.wfs_abi_version()
        if got != WFS_ABI_VERSION:
            raise ImportError("%s has ABI version %d, this binding was written for %d (struct layouts / signatures "
                              "differ): rebuild with `make -C waveformml_amd/csrc`" % (LIB_PATH, got, WFS_ABI_VERSION))
        # WFS_WIDE_MIN_CHANNELS: A/B of the wide-layer path (csrc/wide.hip) -- 0 = off, >= 8 = its channel threshold
        wide = os.environ.get("WFS_WIDE_MIN_CHANNELS")
        if wide is not None:
            lib.wfs_wide_enable(int(wide))
        _LIB = lib
    return _LIB


def last_error():
    msg = load().wfs_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc):
    if rc != WFS_OK:
        raise RuntimeError("libwfsparse: %s (status %d)" % (last_error(), rc))


def stream_ptr():
    """Raw hipStream_t of torch's current stream on the current device (kernels are launched there)."""
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def ptr(t):
    """Raw device pointer of a contiguous CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("waveformml_amd: tensor must live on the GPU (there is no CPU path); got %s" % t.device)
    if not t.is_contiguous():
        raise RuntimeError("waveformml_amd: tensor must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def dtype_code(t):
    if t.dtype == torch.float32:
        return WFS_F32
    if t.dtype == torch.bfloat16:
        return WFS_BF16
    if t.dtype == torch.float16:
        return WFS_F16
    raise RuntimeError("waveformml_amd: features must be float32, bfloat16 or float16, got %s" % t.dtype)


def i32_array(values):
    return (ctypes.c_int32 * len(values))(*[int(v) for v in values])


def make_geometry(ndim, batch_size, spatial, ksize, stride, padding, dilation, subm, transposed=False,
                  output_padding=None):
    g = Geometry()
    g.ndim, g.batch_size, g.subm = int(ndim), int(batch_size), int(bool(subm))
    g.transposed = int(bool(transposed))
    for i in range(ndim):
        g.spatial[i], g.ksize[i] = int(spatial[i]), int(ksize[i])
        g.stride[i], g.padding[i], g.dilation[i] = int(stride[i]), int(padding[i]), int(dilation[i])
        g.output_padding[i] = int(output_padding[i]) if output_padding is not None else 0
    check(load().wfs_geometry_init(ctypes.byref(g)))
    return g


def timing_enable(on=True):
    check(load().wfs_timing_enable(1 if on else 0))


def timing_read(timer):
    ms, n = ctypes.c_double(0), ctypes.c_int64(0)
    check(load().wfs_timing_read(timer, ctypes.byref(ms), ctypes.byref(n)))
    return ms.value, n.value

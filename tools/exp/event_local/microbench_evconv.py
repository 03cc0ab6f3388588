"""Event-local conv kernels against the tile-parallel ones on the bench workload's layers (run on the GPU box):
results compared, then each timed inside a replayed HIP graph.
usage: python tools/microbench_evconv.py [iters] [events]"""
import ctypes, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
import numpy as np, torch
from waveformml_amd import _lib
from waveformml_amd.psd import synthetic
from waveformml_amd.spconv import ops, functional as Fsp

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 256
DT = torch.bfloat16
dev = torch.device("cuda:0")
lib = _lib.load()
# the experiments' own library (make -C tools/exp/event_local), resolved after the product's symbols
xlib = ctypes.CDLL(os.environ.get("WFS_EVEXP_LIB") or os.path.join(HERE, "libwfs_evexp.so"))
_vp, _i32, _i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
xlib.wfs_slot_table.argtypes = [_vp, _i32, _i32, _i32, _i64, _vp, _vp, _i32, _vp, _vp, _vp]
xlib.wfs_event_conv.argtypes = [_vp, _i32, _i32, _i32, _i64, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp]
xlib.wfs_event_rulebook_conv.argtypes = [ctypes.POINTER(_lib.Geometry), _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp,
                                         _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]
xlib.wfs_event_rulebook_conv_workspace_bytes.argtypes = [_i32]
xlib.wfs_event_rulebook_conv_workspace_bytes.restype = ctypes.c_size_t
for _n in ("wfs_slot_table", "wfs_event_conv", "wfs_event_rulebook_conv"):
    getattr(xlib, _n).restype = ctypes.c_int


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def I64(v):
    return ctypes.c_int64(int(v))
c, f, y = synthetic.generate(NB, 256, 3, seed=1234)
torch.cuda.set_stream(torch.cuda.Stream())
idx = torch.from_numpy(np.ascontiguousarray(c[:, [3, 0, 1, 2]])).to(dev)
rb = ops.build_rulebook(idx, NB, [14, 11, 256], [3] * 3, [1] * 3, [0] * 3, [1] * 3, True)
rb1 = ops.build_rulebook(idx, NB, [14, 11, 256], [3] * 3, [1, 1, 4], [0] * 3, [1] * 3, False, known_unique=True)
N, M1 = rb.N, rb1.M
X = torch.randn(N, 32, device=dev).to(DT)
dY = torch.randn(N, 32, device=dev).to(DT)
dY1 = torch.randn(M1, 32, device=dev).to(DT)
W = torch.randn(27, 32, 32, device=dev) * 0.1
bias = torch.randn(32, device=dev)


def offsets(indices, n):
    out = torch.empty((int(lib.wfs_event_offsets_ints(NB)),), dtype=torch.int32, device=dev)
    _lib.check(lib.wfs_event_offsets(_lib.ptr(indices), n, indices.shape[1] - 1, NB, None, _lib.ptr(out), _lib.stream_ptr()))
    return out


ev_in = offsets(idx, N)
ev_out1 = offsets(rb1.out_indices, M1)
ref = np.searchsorted(c[:, 3], np.arange(NB + 1))
assert (ev_in[:NB + 1].cpu().numpy() == ref).all() and int(ev_in[NB + 1:].sum()) == 0
print("event offsets ok; largest event %d rows, strided outputs %d" % (np.diff(ref).max(), int(np.diff(ev_out1[:NB + 1].cpu().numpy()).max())))


_SLOTS = {}


def slots_of(table, mirror, ident, R, ev_o, ev_i):
    key = (table.data_ptr(), mirror, ident, ev_o.data_ptr(), ev_i.data_ptr())
    if key not in _SLOTS:
        s = torch.empty((R, 32), dtype=torch.int16, device=dev)
        _lib.check(xlib.wfs_slot_table(_lib.ptr(table), mirror, 27, ident, R, _lib.ptr(ev_o), _lib.ptr(ev_i), NB, None,
                                      _lib.ptr(s), _lib.stream_ptr()))
        _SLOTS[key] = s
    return _SLOTS[key]


def evconv(table, mirror, ident, R, ev_o, ev_i, Xin, tr, b=None):
    Y = torch.empty((R, 32), dtype=DT, device=dev)
    sl = slots_of(table, mirror, ident, R, ev_o, ev_i)
    _lib.check(xlib.wfs_event_conv(_lib.ptr(table), mirror, 27, ident, R, _lib.ptr(sl), 0, _lib.ptr(ev_o), _lib.ptr(ev_i), NB, _lib.ptr(Xin),
                                  _lib.ptr(W), 1 if tr else 0, _lib.ptr(b), _lib.ptr(Y), _lib.dtype_code(Xin), None,
                                  _lib.stream_ptr()))
    return Y


def timeit(name, fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    print("%-40s %8.1f us" % (name, a.elapsed_time(b) / (iters * reps) * 1e3), flush=True)


t, km = rb.table_by_out()
cases = [
    ("subm fwd", lambda: Fsp.gather_conv(t, km, 27, rb.centre_k, N, X, W, False, bias),
     lambda: evconv(t, 1, rb.centre_k, N, ev_in, ev_in, X, False, bias)),
    ("subm dX", lambda: Fsp.gather_conv(rb.nbr_out, None, 27, rb.centre_k, N, dY, W, True, None),
     lambda: evconv(rb.nbr_out, 0, rb.centre_k, N, ev_in, ev_in, dY, True)),
    ("conv s4 fwd", lambda: Fsp.gather_conv(rb1.nbr_in, None, 27, -1, M1, X, W, False, None),
     lambda: evconv(rb1.nbr_in, 0, -1, M1, ev_out1, ev_in, X, False)),
    ("conv s4 dX", lambda: Fsp.gather_conv(rb1.nbr_out, None, 27, -1, N, dY1, W, True, None),
     lambda: evconv(rb1.nbr_out, 0, -1, N, ev_in, ev_out1, dY1, True)),
]
ONLY = os.environ.get("EV_ONLY") is not None      # knock-out libraries: timings only (their results are wrong)
for name, old, new in ([] if ONLY else cases):
    a, b = old().float(), new().float()
    torch.cuda.synchronize()
    err = float((a - b).abs().max()) / float(a.abs().max())
    print("%-12s max |tile-parallel - event-local| / scale = %.3g   %s" % (name, err, "BIT-EQUAL" if torch.equal(a, b) else ""))
    assert err < 1e-2, name
if ONLY:
    for name, old, new in cases:
        timeit(name + "  event-local", new)
    sys.exit(0)
# a row set that is not grouped by event: the same launch falls back to tile-parallel gathers
bad = ev_in.clone()
bad[NB + 1] = 1
a = Fsp.gather_conv(t, km, 27, rb.centre_k, N, X, W, False, bias).float()
b = evconv(t, 1, rb.centre_k, N, bad, bad, X, False, bias).float()
assert torch.equal(a, b), "fallback path differs"
print("fallback (flagged offsets) ok")
for name, old, new in cases:
    timeit(name + "  tile-parallel", old)
    timeit(name + "  event-local", new)
timeit("event offsets (N rows)", lambda: offsets(idx, N))


def mk_slots():
    s = torch.empty((N, 32), dtype=torch.int16, device=dev)
    _lib.check(xlib.wfs_slot_table(_lib.ptr(t), 1, 27, rb.centre_k, N, _lib.ptr(ev_in), _lib.ptr(ev_in), NB, None, _lib.ptr(s),
                                  _lib.stream_ptr()))
    return s


timeit("slot table (subm, N rows)", mk_slots)

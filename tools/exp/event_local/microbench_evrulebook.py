"""Event-local rulebook builds against rulebook.hip's chip-wide ones on the bench batch (run on the GPU box): tables
compared bit for bit, then both timed inside replayed HIP graphs.  usage: python tools/microbench_evrulebook.py [iters] [events]"""
import ctypes, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
import numpy as np, torch
from waveformml_amd import _lib
from waveformml_amd.psd import synthetic
from waveformml_amd.spconv import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 256
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
N = idx.shape[0]
nv = torch.tensor([N], dtype=torch.int64, device=dev)
SP = [14, 11, 256]


def offsets(indices, n):
    out = torch.empty((int(lib.wfs_event_offsets_ints(NB)),), dtype=torch.int32, device=dev)
    _lib.check(lib.wfs_event_offsets(_lib.ptr(indices), n, indices.shape[1] - 1, NB, None, _lib.ptr(out), _lib.stream_ptr()))
    return out


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
    print("%-44s %8.1f us" % (name, a.elapsed_time(b) / (iters * reps) * 1e3), flush=True)


ev = offsets(idx, N)
# ---- SubM
rb = ops.build_rulebook(idx, NB, SP, [3] * 3, [1] * 3, [0] * 3, [1] * 3, True)
g = _lib.make_geometry(3, NB, SP, [3] * 3, [1] * 3, [0] * 3, [1] * 3, True)
flags = torch.full((int(lib.wfs_event_rulebook_flag_ints(NB)),), 7, dtype=torch.int32, device=dev)
nbr = torch.full((27, N), -7, dtype=torch.int32, device=dev)
slots = torch.zeros((N, 32), dtype=torch.int16, device=dev)


def ev_subm(with_slots=True):
    _lib.check(lib.wfs_event_rulebook_subm(ctypes.byref(g), _lib.ptr(idx), N, _lib.ptr(nv), _lib.ptr(ev), _lib.ptr(nbr),
                                           _lib.ptr(slots) if with_slots else None, _lib.ptr(flags), _lib.stream_ptr()))


ER_ONLY = os.environ.get("ER_ONLY")
ev_subm()
torch.cuda.synchronize()
assert ER_ONLY or int(flags.abs().sum()) == 0, flags.tolist()
assert torch.equal(nbr, rb.nbr_out), "SubM nbr_out differs"
ref_slots = torch.empty((N, 32), dtype=torch.int16, device=dev)
_lib.check(xlib.wfs_slot_table(_lib.ptr(rb.nbr_out), 0, 27, -1, N, _lib.ptr(ev), _lib.ptr(ev), NB, None, _lib.ptr(ref_slots),
                              _lib.stream_ptr()))
assert torch.equal(slots, ref_slots), "SubM slots differ"
print("SubM: event-local nbr_out and slot records BIT-EQUAL to the chip-wide build (N %d)" % N)
timeit("subm build, chip-wide (3 launches)", lambda: ops.build_rulebook(idx, NB, SP, [3] * 3, [1] * 3, [0] * 3, [1] * 3, True, n_dev=nv))
timeit("subm build, event-local + slots", lambda: ev_subm(True))
timeit("subm build, event-local", lambda: ev_subm(False))
timeit("event offsets", lambda: offsets(idx, N))

# ---- strided conv, two layers (the PSD net's k3 s(1,1,4) stack)
def ev_conv(geo, indices, n, n_dev, ev_i, m_cap, cell=False, slots=False):
    K = int(geo.K)
    out = dict(nbr_out=torch.full((K, n), -7, dtype=torch.int32, device=dev),
               nbr_in=torch.full((K, m_cap), -7, dtype=torch.int32, device=dev),
               out_indices=torch.full((m_cap, 4), -7, dtype=torch.int32, device=dev),
               out_ev=torch.full((int(lib.wfs_event_offsets_ints(NB)),), -7, dtype=torch.int32, device=dev),
               info=torch.zeros((4,), dtype=torch.int64, device=dev), m_dev=torch.zeros((1,), dtype=torch.int64, device=dev),
               overflow=torch.zeros((1,), dtype=torch.int32, device=dev), flags=torch.zeros((4,), dtype=torch.int32, device=dev))
    vo = int(np.prod([geo.out_shape[i] for i in range(3)]))
    if cell:
        out["ticket"] = torch.full((NB * vo,), 7, dtype=torch.int32, device=dev)
        out["rowmap"] = torch.full((NB * vo,), -7, dtype=torch.int32, device=dev)
    if slots:
        out["slots"] = torch.full((n, 32), -1, dtype=torch.int16, device=dev)
        out["slots_fwd"] = torch.full((m_cap, 32), -1, dtype=torch.int16, device=dev)
    ws = torch.empty((int(xlib.wfs_event_rulebook_conv_workspace_bytes(NB)),), dtype=torch.uint8, device=dev)

    def run():
        _lib.check(xlib.wfs_event_rulebook_conv(ctypes.byref(geo), _lib.ptr(indices), n, _lib.ptr(n_dev), _lib.ptr(ev_i),
                                               _lib.ptr(out["nbr_out"]), _lib.ptr(out["nbr_in"]), _lib.ptr(out["out_indices"]),
                                               m_cap, _lib.ptr(out["out_ev"]), _lib.ptr(out["info"]), _lib.ptr(out["m_dev"]),
                                               _lib.ptr(out["overflow"]), _lib.ptr(out["flags"]), _lib.ptr(out.get("ticket")),
                                               _lib.ptr(out.get("rowmap")), _lib.ptr(out.get("slots")), _lib.ptr(out.get("slots_fwd")),
                                               _lib.ptr(ws), ws.numel(),
                                               _lib.stream_ptr()))
    out["run"] = run
    return out


rb1 = ops.build_rulebook(idx, NB, SP, [3] * 3, [1, 1, 4], [0] * 3, [1] * 3, False, known_unique=True)
if ER_ONLY:
    g1 = _lib.make_geometry(3, NB, SP, [3] * 3, [1, 1, 4], [0] * 3, [1] * 3, False)
    c1 = ev_conv(g1, idx, N, nv, ev, rb1.M + 1000, slots=True)
    timeit("conv s4 layer 1 build, event-local (2)", c1["run"])
    sys.exit(0)
g1 = _lib.make_geometry(3, NB, SP, [3] * 3, [1, 1, 4], [0] * 3, [1] * 3, False)
M1 = rb1.M
cap1 = M1 + 1000
c1 = ev_conv(g1, idx, N, nv, ev, cap1, slots=True)
c1["run"]()
torch.cuda.synchronize()
assert c1["flags"].tolist() == [0, 0, 0, 0], c1["flags"].tolist()
assert int(c1["info"][0]) == M1 and int(c1["m_dev"][0]) == M1 and int(c1["overflow"][0]) == 0, (int(c1["info"][0]), M1)
assert torch.equal(c1["out_indices"][:M1], rb1.out_indices), "conv1 out_indices differ"
assert torch.equal(c1["nbr_out"], rb1.nbr_out), "conv1 nbr_out differs"
assert torch.equal(c1["nbr_in"][:, :M1], rb1.nbr_in), "conv1 nbr_in differs"
ev1_ref = offsets(rb1.out_indices, M1)
assert torch.equal(c1["out_ev"][:NB + 1 + 64], ev1_ref[:NB + 1 + 64]), "conv1 output event offsets differ"
ref_s = torch.empty((N, 32), dtype=torch.int16, device=dev)
_lib.check(xlib.wfs_slot_table(_lib.ptr(rb1.nbr_out), 0, 27, -1, N, _lib.ptr(ev), _lib.ptr(ev1_ref), NB, None, _lib.ptr(ref_s),
                              _lib.stream_ptr()))
assert torch.equal(c1["slots"], ref_s), "conv1 dX slot records differ"
ref_f = torch.empty((M1, 32), dtype=torch.int16, device=dev)
_lib.check(xlib.wfs_slot_table(_lib.ptr(rb1.nbr_in), 0, 27, -1, M1, _lib.ptr(ev1_ref), _lib.ptr(ev), NB, None, _lib.ptr(ref_f),
                              _lib.stream_ptr()))
assert torch.equal(c1["slots_fwd"][:M1], ref_f), "conv1 forward slot records differ"
print("conv s4 layer 1: event-local tables BIT-EQUAL (M %d)" % M1)
SP1 = rb1.out_spatial_shape
rb2 = ops.build_rulebook(rb1.out_indices, NB, SP1, [3] * 3, [1, 1, 4], [0] * 3, [1] * 3, False, known_unique=True)
g2 = _lib.make_geometry(3, NB, SP1, [3] * 3, [1, 1, 4], [0] * 3, [1] * 3, False)
M2 = rb2.M
idx1 = c1["out_indices"]
c2 = ev_conv(g2, idx1, cap1, c1["m_dev"], c1["out_ev"], M2 + 500, cell=True)
c2["run"]()
torch.cuda.synchronize()
assert c2["flags"].tolist() == [0, 0, 0, 0] and int(c2["info"][0]) == M2, (c2["flags"].tolist(), int(c2["info"][0]), M2)
assert torch.equal(c2["out_indices"][:M2], rb2.out_indices) and torch.equal(c2["nbr_out"][:, :M1], rb2.nbr_out)
assert torch.equal(c2["nbr_in"][:, :M2], rb2.nbr_in)
vo2 = int(np.prod(rb2.out_spatial_shape))
dense_ref = torch.full((NB * vo2,), -1, dtype=torch.int32, device=dev)
oi = rb2.out_indices.long()
lin = ((oi[:, 0] * rb2.out_spatial_shape[0] + oi[:, 1]) * rb2.out_spatial_shape[1] + oi[:, 2]) * rb2.out_spatial_shape[2] + oi[:, 3]
dense_ref[lin] = torch.arange(M2, device=dev, dtype=torch.int32)
assert torch.equal(c2["rowmap"], dense_ref) and torch.equal(c2["ticket"] != -1, dense_ref >= 0), "cell map differs"
print("conv s4 layer 2: event-local tables and cell map BIT-EQUAL (M %d)" % M2)
timeit("conv s4 layer 1 build, chip-wide (6 launches)", lambda: ops.build_rulebook(idx, NB, SP, [3] * 3, [1, 1, 4], [0] * 3, [1] * 3, False, n_dev=nv, out_capacity=cap1))
timeit("conv s4 layer 1 build, event-local (2)", c1["run"])
m1d = c1["m_dev"]
timeit("conv s4 layer 2 build, chip-wide", lambda: ops.build_rulebook(idx1, NB, SP1, [3] * 3, [1, 1, 4], [0] * 3, [1] * 3, False, n_dev=m1d, out_capacity=M2 + 500))
timeit("conv s4 layer 2 build, event-local + cell map", c2["run"])

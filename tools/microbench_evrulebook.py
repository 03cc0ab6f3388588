"""Event-local rulebook builds against rulebook.hip's chip-wide ones on the bench batch (run on the GPU box): tables
compared bit for bit, then both timed inside replayed HIP graphs.  usage: python tools/microbench_evrulebook.py [iters] [events]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveformml_amd import _lib
from waveformml_amd.psd import synthetic
from waveformml_amd.spconv import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
lib = _lib.load()
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
flags = torch.zeros((4,), dtype=torch.int32, device=dev)
nbr = torch.full((27, N), -7, dtype=torch.int32, device=dev)
slots = torch.zeros((N, 32), dtype=torch.int16, device=dev)


def ev_subm(with_slots=True):
    _lib.check(lib.wfs_event_rulebook_subm(ctypes.byref(g), _lib.ptr(idx), N, _lib.ptr(nv), _lib.ptr(ev), _lib.ptr(nbr),
                                           _lib.ptr(slots) if with_slots else None, _lib.ptr(flags), _lib.stream_ptr()))


if os.environ.get("ER_ONLY"):
    timeit("subm build, event-local + slots", lambda: ev_subm(True))
    sys.exit(0)
ev_subm()
torch.cuda.synchronize()
assert flags.tolist() == [0, 0, 0, 0], flags.tolist()
assert torch.equal(nbr, rb.nbr_out), "SubM nbr_out differs"
ref_slots = torch.empty((N, 32), dtype=torch.int16, device=dev)
_lib.check(lib.wfs_slot_table(_lib.ptr(rb.nbr_out), 0, 27, -1, N, _lib.ptr(ev), _lib.ptr(ev), NB, None, _lib.ptr(ref_slots),
                              _lib.stream_ptr()))
assert torch.equal(slots, ref_slots), "SubM slots differ"
print("SubM: event-local nbr_out and slot records BIT-EQUAL to the chip-wide build (N %d)" % N)
timeit("subm build, chip-wide (3 launches)", lambda: ops.build_rulebook(idx, NB, SP, [3] * 3, [1] * 3, [0] * 3, [1] * 3, True, n_dev=nv))
timeit("subm build, event-local + slots", lambda: ev_subm(True))
timeit("subm build, event-local", lambda: ev_subm(False))
timeit("event offsets", lambda: offsets(idx, N))

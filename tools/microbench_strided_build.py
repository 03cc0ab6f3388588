"""Micro-benchmark of the strided layers' rulebook builds and of the products that read their tables, on the bench
workload's geometry (run on the GPU box):  python tools/microbench_strided_build.py [iters] [events] [bf16|f32]
Event-local build (csrc/evconv.hip, one launch) against the chip-wide build (rulebook.hip, six launches); dX / dW through
the packed [9, N] table against the dense [27, N] one.  Every timing is GPU time per call inside a replayed HIP graph."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from waveformml_amd.psd import synthetic
from waveformml_amd.spconv import functional as Fsp
from waveformml_amd.spconv import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 256
DT = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else torch.bfloat16
dev = torch.device("cuda:0")
c, f, y = synthetic.generate(NB, 256, 3, seed=1234)
torch.cuda.set_stream(torch.cuda.Stream())
idx = torch.from_numpy(np.ascontiguousarray(c[:, [3, 0, 1, 2]])).to(dev)
GEO = ([3] * 3, [1, 1, 4], [0] * 3, [1] * 3)
e1 = ops.build_rulebook(idx, NB, [14, 11, 256], *GEO, False, known_unique=True)
e2 = ops.build_rulebook(e1.out_indices, NB, e1.out_spatial_shape, *GEO, False, known_unique=True)
N, M1, M2 = e1.N, e1.M, e2.M
P1, P2 = int((e1.nbr_out >= 0).sum()), int((e2.nbr_out >= 0).sum())
print("events %d  N %d  M1 %d  P1 %d  M2 %d  P2 %d" % (NB, N, M1, P1, M2, P2), flush=True)


def timeit(name, fn, nbytes=None, reps=10):
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
    us = a.elapsed_time(b) / (iters * reps) * 1e3
    extra = "  %.0f GB/s algorithmic" % (nbytes / us / 1e3) if nbytes else ""
    print("%-52s %8.1f us%s" % (name, us, extra), flush=True)
    return us


nv = torch.tensor([N], dtype=torch.int64, device=dev)
cap1, cap2 = int(M1 * 1.19), int(M2 * 1.19)
alg1 = N * 16 + 8 * P1 + M1 * 16
alg2 = M1 * 16 + 8 * P2 + M2 * 16
last = {}
for on in (True, False):
    ops.EVENT_LOCAL_CONV = on
    tag = "event-local" if on else "chip-wide "
    s1, s2 = {}, {}

    def b1():
        last[on] = ops.build_rulebook(idx, NB, [14, 11, 256], *GEO, False, n_dev=nv, out_capacity=cap1, flags=s1,
                                      want_cell_map=False)
        return last[on]

    timeit("strided build, layer 1 (%s)" % tag, b1, alg1)
    r1 = b1()

    def b2():
        return ops.build_rulebook(r1.out_indices, NB, r1.out_spatial_shape, *GEO, False, n_dev=r1.m_dev,
                                  out_capacity=cap2, events=r1.events_out, flags=s2, want_cell_map=True)

    timeit("strided build, layer 2 + cell map (%s)" % tag, b2, alg2)
ops.EVENT_LOCAL_CONV = True

rb = last[True]
torch.cuda.synchronize()
assert int(rb.m_dev) == M1 and torch.equal(rb.out_indices[:M1], e1.out_indices)
ES = 2 if DT != torch.float32 else 4
X = torch.randn(N, 32, device=dev).to(DT)
dY1 = torch.randn(rb.M, 32, device=dev).to(DT)
W = torch.randn(27, 32, 32, device=dev) * 0.1
by1 = N * 32 * ES + M1 * 32 * ES + P1 * 8 + 27 * 4096
dense = rb.nbr_out
tp, pk = rb.table_by_in(32, 32, X, 1)
timeit("conv s4 fwd 32->32 (dense nbr_in)", lambda: Fsp.gather_conv(rb.nbr_in, None, 27, -1, rb.M, X, W, False, None, rb.m_dev), by1)
timeit("conv s4 dX, dense [27, N] table", lambda: Fsp.gather_conv(dense, None, 27, -1, N, dY1, W, True, None, nv), by1)
timeit("conv s4 dX, packed [9, N] table", lambda: Fsp.gather_conv(tp, None, 27, -1, N, dY1, W, True, None, nv, None, pk), by1)
timeit("conv s4 dW, dense [27, N] table", lambda: Fsp.gather_dw(dense, 27, -1, N, X, dY1, False, None, nv), by1)
timeit("conv s4 dW, packed [9, N] table", lambda: Fsp.gather_dw(tp, 27, -1, N, X, dY1, False, None, nv, False, None, pk), by1)

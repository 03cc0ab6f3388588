"""The shape-generic MFMA conv kernels (gather_conv.hip k_gconv_mfma / k_gdw_mfma) on the reference's 2-D layer shapes
(GEP.json: 252 -> 158 and 158 -> 64 channels, 3 x 3, a few hundred rows) against the dense 128 x 128-tile
products of wide.hip (the round-2 library-GEMM route these replaced: profiles/r02_microbench_generic_mfma.txt).  usage: python tools/microbench_generic.py [events] [dtype] [wide]      (wide: the hybrid net's 1697 -> 1021 -> 345 layers)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveformml_amd.psd import synthetic
from waveformml_amd.spconv import ops, functional as Fsp

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 256
DT = {"f32": torch.float32, "bf16": torch.bfloat16}[sys.argv[2] if len(sys.argv) > 2 else "f32"]
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream())
c, f, y = synthetic.generate(NB, 150, 3, seed=1, layout="2d")
idx = torch.from_numpy(np.ascontiguousarray(c[:, [2, 0, 1]])).to(dev)
rb = ops.build_rulebook(idx, NB, [14, 11], [3, 3], [1, 1], [0, 0], [1, 1], False, known_unique=True)
N, M, K = rb.N, rb.M, rb.K
print("rows in %d out %d pairs %d" % (N, M, int((rb.nbr_out >= 0).sum())))


def timeit(name, fn, reps=10, iters=20):
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
    print("%-46s %8.1f us" % (name, a.elapsed_time(b) / (iters * reps) * 1e3), flush=True)


WIDE = len(sys.argv) > 3 and sys.argv[3] == "wide"
SHAPES = ((1697, 1021), (1021, 345), (300, 252)) if WIDE else ((252, 158), (158, 64), (64, 64), (130, 138))
from waveformml_amd import _lib
lib = _lib.load()
# the hybrid net's second 3 x 3 layer works on the OUTPUT set of the first one
rb2 = ops.build_rulebook(rb.out_indices, NB, [12, 9], [3, 3], [1, 1], [0, 0], [1, 1], False, known_unique=True)
print("second layer: rows in %d out %d pairs %d" % (rb2.N, rb2.M, int((rb2.nbr_out >= 0).sum())))
for (ci, co) in SHAPES:
    r = rb2 if (ci, co) == (1021, 345) else rb
    X = torch.randn(r.N, ci, device=dev).to(DT)
    dY = torch.randn(r.M, co, device=dev).to(DT)
    W = torch.randn(K, ci, co, device=dev) * 0.05
    routes = [("libwfsparse wide (128 x 128-tile MFMA)", 1, 0), ("libwfsparse 32 x 32-tile MFMA", 0, 0)]
    for route, wide, lim in routes:
        lib.wfs_wide_enable(wide)
        timeit("%d->%d fwd  %s" % (ci, co, route), lambda: Fsp.gather_conv(r.nbr_in, None, K, -1, r.M, X, W, False, None))
        timeit("%d->%d dX   %s" % (ci, co, route), lambda: Fsp.gather_conv(r.nbr_out, None, K, -1, r.N, dY, W, True, None))
        timeit("%d->%d dW   %s" % (ci, co, route), lambda: Fsp.gather_dw(r.nbr_out, K, -1, r.N, X, dY, False))
    lib.wfs_wide_enable(1)
if WIDE:
    # the hybrid net's 1 x 1 layer (2048 -> 1697): spconv's torch.mm against the wide path
    R = rb.N
    X = torch.randn(R, 2048, device=dev).to(DT)
    dY = torch.randn(R, 1697, device=dev).to(DT)
    W = torch.randn(1, 2048, 1697, device=dev) * 0.02
    Wh = W[0].to(DT)
    timeit("2048->1697 1x1 fwd  libwfsparse wide", lambda: Fsp.gather_conv(None, None, 1, 0, R, X, W, False, None))
    timeit("2048->1697 1x1 dX   libwfsparse wide", lambda: Fsp.gather_conv(None, None, 1, 0, R, dY, W, True, None))
    timeit("2048->1697 1x1 dW   libwfsparse wide", lambda: Fsp.gather_dw(None, 1, 0, R, X, dY, False))
    timeit("2048->1697 1x1 fwd  torch.mm (+ filter cast)", lambda: torch.mm(X, W[0].to(DT)))
    timeit("2048->1697 1x1 dX   torch.mm (+ filter cast)", lambda: torch.mm(dY, W[0].to(DT).t()))
    timeit("2048->1697 1x1 dW   torch.mm", lambda: torch.mm(X.t(), dY).float())

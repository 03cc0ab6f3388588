"""The shape-generic MFMA conv kernels (gather_conv.hip k_gconv_mfma / k_gdw_mfma) on the reference's 2-D layer shapes
(GEP.json: 252 -> 158 and 158 -> 64 channels, 3 x 3, a few hundred rows) against the route they replace (rows gathered
with torch index kernels into [R, K * C] + one library GEMM).  usage: python tools/microbench_generic.py [events] [dtype] [wide]      (wide: the hybrid net's 1697 -> 1021 -> 345 layers)"""
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


SHAPES = ((1697, 1021), (1021, 345)) if (len(sys.argv) > 3 and sys.argv[3] == "wide") else ((252, 158), (158, 64), (64, 64), (130, 138))
for (ci, co) in SHAPES:
    X = torch.randn(N, ci, device=dev).to(DT)
    dY = torch.randn(M, co, device=dev).to(DT)
    W = torch.randn(K, ci, co, device=dev) * 0.05
    for route, lim in (("libwfsparse MFMA", 1 << 30), ("torch gather + library GEMM", 8)):
        Fsp.GEMM_ROUTE_MIN_CHANNELS = lim
        timeit("%d->%d fwd  %s" % (ci, co, route), lambda: Fsp.gather_conv(rb.nbr_in, None, K, -1, M, X, W, False, None))
        timeit("%d->%d dX   %s" % (ci, co, route), lambda: Fsp.gather_conv(rb.nbr_out, None, K, -1, N, dY, W, True, None))
        timeit("%d->%d dW   %s" % (ci, co, route), lambda: Fsp.gather_dw(rb.nbr_out, K, -1, N, X, dY, False))

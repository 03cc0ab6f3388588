"""GPU time of the k_gemm16 launches alone (HIP-graph replay of the three products of the 1697 -> 1021 layer, minus the
same graphs with the staging kernels only is not possible from outside: so the whole op is timed and the staging part
is constant across knock-outs).  env WFS_WIDE_MODE = wfs_wide_enable bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from waveformml_amd import _lib
from waveformml_amd.psd import synthetic
from waveformml_amd.spconv import ops, functional as Fsp
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream())
c, f, y = synthetic.generate(256, 150, 3, seed=1, layout="2d")
idx = torch.from_numpy(np.ascontiguousarray(c[:, [2, 0, 1]])).to(dev)
rb = ops.build_rulebook(idx, 256, [14, 11], [3, 3], [1, 1], [0, 0], [1, 1], False, known_unique=True)
K, ci, co = rb.K, 1697, 1021
X = torch.randn(rb.N, ci, device=dev).to(torch.bfloat16)
dY = torch.randn(rb.M, co, device=dev).to(torch.bfloat16)
W = torch.randn(K, ci, co, device=dev) * 0.05
_lib.load().wfs_wide_enable(int(os.environ.get("WFS_WIDE_MODE", "1")))


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
    print("%-10s %8.1f us" % (name, a.elapsed_time(b) / (iters * reps) * 1e3), flush=True)


timeit("fwd", lambda: Fsp.gather_conv(rb.nbr_in, None, K, -1, rb.M, X, W, False, None))
timeit("dX", lambda: Fsp.gather_conv(rb.nbr_out, None, K, -1, rb.N, dY, W, True, None))
timeit("dW", lambda: Fsp.gather_dw(rb.nbr_out, K, -1, rb.N, X, dY, False))

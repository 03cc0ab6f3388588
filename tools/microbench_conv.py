"""Micro-benchmark of the conv kernels on the bench workload's layers (run on the GPU box).
usage: python tools/microbench_conv.py [iters] [f32|bf16] [events] [samples]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveformml_amd.psd import synthetic
from waveformml_amd.spconv import ops, functional as Fsp

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
DT = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 256
TS = int(sys.argv[4]) if len(sys.argv) > 4 else 256
dev = torch.device("cuda:0")
c, f, y = synthetic.generate(NB, TS, 3, seed=1234)
torch.cuda.set_stream(torch.cuda.Stream())
idx = torch.from_numpy(np.ascontiguousarray(c[:, [3, 0, 1, 2]])).to(dev)
rb = ops.build_rulebook(idx, NB, [14, 11, TS], [3] * 3, [1] * 3, [0] * 3, [1] * 3, True)
rb1 = ops.build_rulebook(idx, NB, [14, 11, TS], [3] * 3, [1, 1, 4], [0] * 3, [1] * 3, False, known_unique=True)
N, M1 = rb.N, rb1.M
X = torch.randn(N, 32, device=dev).to(DT)
dY = torch.randn(N, 32, device=dev).to(DT)
dY1 = torch.randn(M1, 32, device=dev).to(DT)
X2 = torch.randn(N, 2, device=dev).to(DT)
ES = X.element_size()
W = torch.randn(27, 32, 32, device=dev) * 0.1
W2 = torch.randn(27, 2, 32, device=dev) * 0.1


def timeit(name, fn, nbytes=None, reps=10):
    """`reps` launches captured into a HIP graph and replayed: the number is GPU time per launch including the
    in-graph launch gap, free of Python / ctypes overhead (which is ~15 us per call here)."""
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
    print("%-34s %8.1f us%s" % (name, us, extra), flush=True)


P = int((rb.nbr_out >= 0).sum())
P1 = int((rb1.nbr_out >= 0).sum())
print("N %d  P %d  M1 %d  P1 %d" % (N, P, M1, P1))
t, km = rb.table_by_out()
by = N * 32 * ES * 2 + P * 8 + 27 * 4096
timeit("subm fwd 32->32", lambda: Fsp.gather_conv(t, km, 27, rb.centre_k, N, X, W, False, None), by)
timeit("subm dX 32->32", lambda: Fsp.gather_conv(rb.nbr_out, None, 27, rb.centre_k, N, dY, W, True, None), by)
timeit("subm dW 32x32", lambda: Fsp.gather_dw(rb.nbr_out, 27, rb.centre_k, N, X, dY, False), by)
by1 = N * 32 * ES + M1 * 32 * ES + P1 * 8 + 27 * 4096
timeit("conv s4 fwd 32->32", lambda: Fsp.gather_conv(rb1.nbr_in, None, 27, -1, M1, X, W, False, None), by1)
timeit("conv s4 dX", lambda: Fsp.gather_conv(rb1.nbr_out, None, 27, -1, N, dY1, W, True, None), by1)
timeit("conv s4 dW", lambda: Fsp.gather_dw(rb1.nbr_out, 27, -1, N, X, dY1, False), by1)
timeit("subm fwd 2->32", lambda: Fsp.gather_conv(t, km, 27, rb.centre_k, N, X2, W2, False, None), N * 136 + P * 8)
timeit("subm dW 32x2 (dY stationary)", lambda: Fsp.gather_dw(t, 27, rb.centre_k, N, dY, X2, True, km), N * 34 * ES + P * 8)
nv = torch.tensor([N], dtype=torch.int64, device=dev)
timeit("rulebook subm (device counts)", lambda: ops.build_rulebook(idx, NB, [14, 11, TS], [3] * 3, [1] * 3, [0] * 3, [1] * 3, True, n_dev=nv))
timeit("rulebook conv s4 (device counts)", lambda: ops.build_rulebook(idx, NB, [14, 11, TS], [3] * 3, [1, 1, 4], [0] * 3, [1] * 3, False, n_dev=nv, out_capacity=int(M1 * 1.25)))
bn = torch.nn.BatchNorm1d(32).to(dev)
Xg = X.clone().requires_grad_(True)
timeit("bn+relu fwd", lambda: Fsp.batch_norm_relu(X, bn, True), N * 32 * ES * 3)
yb = Fsp.batch_norm_relu(Xg, bn, True)
timeit("bn+relu bwd", lambda: torch.autograd.grad(yb, Xg, dY, retain_graph=True), N * 32 * ES * 3)

# --- per-offset cost: synthetic tables with exactly 1 / 10 / 27 active offsets in every tile
ar = torch.arange(N, device=dev, dtype=torch.int32)
for nact in (1, 10, 27):
    tb = torch.full((27, N), -1, dtype=torch.int32, device=dev)
    for k in range(nact):
        tb[k] = ar
    timeit("fwd 32->32, %2d active offsets/tile" % nact, lambda: Fsp.gather_conv(tb, None, 27, -1, N, X, W, False, None), N * 32 * ES * 2 + nact * N * 8)
    timeit("dW 32x32,   %2d active offsets/tile" % nact, lambda: Fsp.gather_dw(tb, 27, -1, N, X, dY, False), N * 32 * ES * 2 + nact * N * 8)
# measured tile unions of the real tables
def unions(tab, R):
    nt = (R + 31) // 32
    pad = torch.full((27, nt * 32), -1, dtype=torch.int32, device=dev)
    pad[:, :R] = tab
    return float((pad.view(27, nt, 32) >= 0).any(2).sum()) / nt
print("active offsets per 32-row tile: subm %.2f  conv1 nbr_in %.2f  conv1 nbr_out %.2f" % (unions(rb.nbr_out, N), unions(rb1.nbr_in, M1), unions(rb1.nbr_out, N)))

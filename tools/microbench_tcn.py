"""Fused TemporalConvNet (wfs_tcn_fwd / wfs_tcn_bwd) vs the torch composition on the GPU, C5-sized rows.
usage: python tools/microbench_tcn.py [rows] [L] [levels] [k] [f32|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from waveformml_amd.psd.tcn import TemporalConvNet

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
levels = int(sys.argv[3]) if len(sys.argv) > 3 else 3
k = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dt = torch.bfloat16 if (len(sys.argv) > 5 and sys.argv[5] == "bf16") else torch.float32
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream())
net = TemporalConvNet(1, [1] * levels, kernel_size=k, dropout=0.0).to(dev)
x = torch.randn(N, 1, L, device=dev).to(dt).requires_grad_(True)
g = torch.randn(N, 1, L, device=dev).to(dt)
es = x.element_size()


def timeit(name, fn, nbytes, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / iters * 1e3
    print("%-36s %9.1f us   %6.0f GB/s algorithmic" % (name, us, nbytes / us / 1e3), flush=True)


def step(fused):
    x.grad = None
    y = net(x) if fused else net.network(x if dt == torch.float32 else x.float()).to(dt)
    y.backward(g)


print("rows %d  L %d  levels %d  k %d  %s" % (N, L, levels, k, dt))
with torch.no_grad():
    timeit("fused forward", lambda: net(x), 2 * N * L * es)
    timeit("torch forward", lambda: net.network(x if dt == torch.float32 else x.float()), 2 * N * L * es)
timeit("fused forward + backward", lambda: step(True), 5 * N * L * es)
timeit("torch forward + backward", lambda: step(False), 5 * N * L * es)

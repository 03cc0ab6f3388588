"""Rulebook build times at the bench geometry (256 events x 256 samples): the per-layer builds of rulebook.hip (SubM +
two strided layers) against the event-parallel chain (rulebook_chain.hip), device-count mode, HIP-event timing of
back-to-back launches.   usage: python tools/microbench_rulebook.py [events] [samples] [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from waveformml_amd.psd import synthetic
from waveformml_amd.spconv import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream())
c, f, y = synthetic.generate(B, T, 3, seed=1234)
idx = torch.from_numpy(np.ascontiguousarray(c[:, [3, 0, 1, 2]])).to(dev)
ne = np.bincount(c[:, 3], minlength=B)
print("voxels %d; rows per event: mean %.0f median %.0f p90 %.0f max %d" % (len(c), ne.mean(), np.median(ne), np.percentile(ne, 90), ne.max()))
n = idx.shape[0]
n_dev = torch.tensor([n], dtype=torch.int64, device=dev)
specs = [([3] * 3, [1] * 3, [0] * 3, [1] * 3, True), ([3] * 3, [1, 1, 4], [0] * 3, [1] * 3, False),
         ([3] * 3, [1, 1, 4], [0] * 3, [1] * 3, False)]


def per_layer():
    rb0 = ops._build_rulebook(idx, B, [14, 11, T], [3] * 3, [1] * 3, [0] * 3, [1] * 3, True, known_unique=True, n_dev=n_dev)
    rb1 = ops._build_rulebook(idx, B, [14, 11, T], [3] * 3, [1, 1, 4], [0] * 3, [1] * 3, False, known_unique=True, n_dev=n_dev,
                              out_capacity=int(1.2 * n))
    rb2 = ops._build_rulebook(rb1.out_indices, B, rb1.out_spatial_shape, [3] * 3, [1, 1, 4], [0] * 3, [1] * 3, False,
                              known_unique=True, n_dev=rb1.m_dev, out_capacity=int(0.5 * n))
    return rb0, rb1, rb2


def chain():
    return ops.build_rulebook_chain(idx, B, [14, 11, T], specs, n_dev=n_dev, capacities=[None, int(1.2 * n), int(0.5 * n)],
                                    cell_maps=[False, False, True])


for name, fn in (("per-layer (rulebook.hip, 15 launches)", per_layer), ("chain (rulebook_chain.hip, 2 launches + memset)", chain)):
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-50s %8.1f us per build" % (name, e0.elapsed_time(e1) / iters * 1e3))
rbs = chain()
print("M3 %d M4 %d" % (int(rbs[1].m_dev), int(rbs[2].m_dev)))

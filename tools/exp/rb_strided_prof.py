"""The strided (1, 1, 4) rulebook build alone, device-count mode, repeated (for rocprofv3 --kernel-trace --stats / --pmc).
usage: python tools/exp/rb_strided_prof.py [events]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from waveformml_amd.psd import synthetic
from waveformml_amd.spconv import ops
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream())
c, f, y = synthetic.generate(NB, 256, 3, seed=1234)
idx = torch.from_numpy(np.ascontiguousarray(c[:, [3, 0, 1, 2]])).to(dev)
nv = torch.tensor([idx.shape[0]], dtype=torch.int64, device=dev)
rb = ops.build_rulebook(idx, NB, [14, 11, 256], [3] * 3, [1, 1, 4], [0] * 3, [1] * 3, False, known_unique=True)
cap = int(rb.M * 1.25)
for _ in range(20):
    ops.build_rulebook(idx, NB, [14, 11, 256], [3] * 3, [1, 1, 4], [0] * 3, [1] * 3, False, n_dev=nv, out_capacity=cap)
torch.cuda.synchronize()
print("N", idx.shape[0], "M", rb.M)

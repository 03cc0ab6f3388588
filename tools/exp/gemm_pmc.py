"""One wide product repeated (for rocprofv3 --pmc of k_gemm16): python tools/exp/gemm_pmc.py [which] [mode]
which: fwd | dx | dw of the hybrid net's 1697 -> 1021 layer; mode = wfs_wide_enable bits (1 default, 7 = 2 register stages)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from waveformml_amd import _lib
from waveformml_amd.psd import synthetic
from waveformml_amd.spconv import ops, functional as Fsp
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
c, f, y = synthetic.generate(256, 150, 3, seed=1, layout="2d")
idx = torch.from_numpy(np.ascontiguousarray(c[:, [2, 0, 1]])).to(dev)
rb = ops.build_rulebook(idx, 256, [14, 11], [3, 3], [1, 1], [0, 0], [1, 1], False, known_unique=True)
K = rb.K
ci, co = 1697, 1021
X = torch.randn(rb.N, ci, device=dev).to(torch.bfloat16)
dY = torch.randn(rb.M, co, device=dev).to(torch.bfloat16)
W = torch.randn(K, ci, co, device=dev) * 0.05
_lib.load().wfs_wide_enable(mode)
for _ in range(5):
    if which == "fwd":
        Fsp.gather_conv(rb.nbr_in, None, K, -1, rb.M, X, W, False, None)
    elif which == "dx":
        Fsp.gather_conv(rb.nbr_out, None, K, -1, rb.N, dY, W, True, None)
    else:
        Fsp.gather_dw(rb.nbr_out, K, -1, rb.N, X, dY, False)
torch.cuda.synchronize()

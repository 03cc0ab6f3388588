"""One 32 -> 32 conv op of the PSD batch repeated (for rocprofv3 --pmc): python tools/exp/conv_pmc.py [fwd|dx|dw] [f32|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from waveformml_amd.psd import synthetic
from waveformml_amd.spconv import ops, functional as Fsp
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
DT = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
dev = torch.device("cuda:0")
c, f, y = synthetic.generate(256, 256, 3, seed=1234)
idx = torch.from_numpy(np.ascontiguousarray(c[:, [3, 0, 1, 2]])).to(dev)
rb = ops.build_rulebook(idx, 256, [14, 11, 256], [3] * 3, [1] * 3, [0] * 3, [1] * 3, True)
N = rb.N
X = torch.randn(N, 32, device=dev).to(DT)
dY = torch.randn(N, 32, device=dev).to(DT)
W = torch.randn(27, 32, 32, device=dev) * 0.1
t, km = rb.table_by_out()
for _ in range(5):
    if which == "fwd":
        Fsp.gather_conv(t, km, 27, rb.centre_k, N, X, W, False, None)
    elif which == "dx":
        Fsp.gather_conv(rb.nbr_out, None, 27, rb.centre_k, N, dY, W, True, None)
    else:
        Fsp.gather_dw(rb.nbr_out, 27, rb.centre_k, N, X, dY, False)
torch.cuda.synchronize()

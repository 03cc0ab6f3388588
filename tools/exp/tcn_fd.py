import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from waveformml_amd.psd.tcn import FusedTCNFunction
DEV = "cuda:0"
rng = np.random.default_rng(5)
levels, k, N, L = 3, 3, 9, 300
seed = torch.tensor([1234567], dtype=torch.int64, device=DEV)
taps0 = torch.from_numpy(rng.standard_normal((levels, 2, k)).astype(np.float32) * 0.6).to(DEV)
bias0 = torch.from_numpy(rng.standard_normal((levels, 2)).astype(np.float32) * 0.3).to(DEV)
x0 = torch.from_numpy(rng.standard_normal((N, L)).astype(np.float32)).to(DEV)
g = torch.from_numpy(rng.standard_normal((N, L)).astype(np.float32)).to(DEV)
for p in (0.0, 0.3):
    taps, bias, x = taps0.clone().requires_grad_(True), bias0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
    f = lambda xx, tt, bb: FusedTCNFunction.apply(xx, tt, bb, p, seed if p > 0 else None)
    f(x, taps, bias).backward(g)
    with torch.no_grad():
        for h in (2e-3, 2e-4):
            out = []
            for i in range(6):
                d = torch.zeros(taps.numel(), device=DEV); d[i] = h; d = d.reshape(taps.shape)
                fd = float((g.double() * (f(x, taps + d, bias).double() - f(x, taps - d, bias).double())).sum() / (2 * h))
                out.append((round(fd, 3), round(float(taps.grad.reshape(-1)[i]), 3)))
            print("p", p, "h", h, out)

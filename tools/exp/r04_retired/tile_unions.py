import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from waveformml_amd.psd import synthetic
from waveformml_amd.spconv import ops
dev = torch.device("cuda:0")
c, f, y = synthetic.generate(256, 256, 3, seed=1234)
idx = torch.from_numpy(np.ascontiguousarray(c[:, [3, 0, 1, 2]])).to(dev)
rb1 = ops.build_rulebook(idx, 256, [14, 11, 256], [3] * 3, [1, 1, 4], [0] * 3, [1] * 3, False, known_unique=True)
t = rb1.nbr_out.cpu().numpy() >= 0          # [27, N]
N = t.shape[1]
print("pairs per row %.2f" % (t.sum() / N))
print("first rows (b,x,y,t):", c[:12, [3,0,1,2]].tolist())
for TR in (16, 32):
    for il in (1, 2, 4):
        per = TR * il
        nsb = (N + per - 1) // per
        pad = np.zeros((27, nsb * per), bool); pad[:, :N] = t
        # row = (sb*TR + i)*il + c  -> reshape [27, sb, i, c]
        v = pad.reshape(27, nsb, TR, il)
        u = v.any(axis=2)            # [27, sb, c]: union over the tile's rows
        print("TR %d il %d: active offsets per tile %.2f" % (TR, il, u.sum() / (nsb * il)))

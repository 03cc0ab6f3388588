import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import rand_coords
import waveformml_amd.spconv as sp
from waveformml_amd import _lib
DEV="cuda:0"
lib=_lib.load()
dtype = torch.float16
res = {}
for wide in (1,0):
    lib.wfs_wide_enable(wide)
    rng = np.random.default_rng(31)
    shape, B, n = (14, 11), 24, 420
    idx = rand_coords(rng, B, shape, n)
    idx = np.ascontiguousarray(idx[np.argsort(idx[:, 0], kind="stable")])
    feat = torch.from_numpy(rng.standard_normal((n, 300)).astype(np.float32)).to(dtype)
    torch.manual_seed(4)
    net = sp.SparseSequential(sp.SparseConv2d(300, 264, 1, 1, 0, 1, 1, True), torch.nn.BatchNorm1d(264), torch.nn.ReLU(),
                              sp.SparseConv2d(264, 130, 3, 1, 0, 1, 1, True)).to(DEV)
    fg = feat.to(DEV).requires_grad_(True)
    x = sp.SparseConvTensor(fg, torch.from_numpy(idx).to(DEV), list(shape), B)
    inter = {}
    x1 = net[0](x); inter["y1"] = x1.features; x1.features.retain_grad()
    f2 = net[2](net[1](x1.features)); inter["y2"] = f2; f2.retain_grad()
    x1.features = f2
    x3 = net[3](x1); inter["y3"] = x3.features
    g = torch.from_numpy(rng.standard_normal(tuple(x3.features.shape)).astype(np.float32))
    x3.features.backward(g.to(DEV).to(dtype))
    inter["g_y2"] = f2.grad; inter["g_y1"] = inter["y1"].grad; inter["g_x"] = fg.grad
    for n_, p in net.named_parameters(): inter["g_" + n_] = p.grad
    res[wide] = {k: v.detach().float().cpu().numpy() for k, v in inter.items()}
for k in res[1]:
    a, b = res[1][k], res[0][k]
    print("%-12s rel l2 wide vs narrow %.3e   max %.3e" % (k, np.linalg.norm(a-b)/np.linalg.norm(b), np.abs(a-b).max()/np.abs(b).max()))

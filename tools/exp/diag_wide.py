import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import rand_coords
import waveformml_amd.spconv as sp
from waveformml_amd import _lib
from oracle import spconv as osp
DEV="cuda:0"
lib=_lib.load()
for dtype in (torch.bfloat16, torch.float16):
  for wide in (1,0):
    lib.wfs_wide_enable(wide)
    rng = np.random.default_rng(31)
    shape, B, n = (14, 11), 24, 420
    idx = rand_coords(rng, B, shape, n)
    idx = np.ascontiguousarray(idx[np.argsort(idx[:, 0], kind="stable")])
    feat = torch.from_numpy(rng.standard_normal((n, 300)).astype(np.float32)).to(dtype)
    torch.manual_seed(4)
    ref = osp.SparseSequential(osp.SparseConv2d(300, 264, 1, 1, 0, 1, 1, True), torch.nn.BatchNorm1d(264), torch.nn.ReLU(),
                               osp.SparseConv2d(264, 130, 3, 1, 0, 1, 1, True))
    net = sp.SparseSequential(sp.SparseConv2d(300, 264, 1, 1, 0, 1, 1, True), torch.nn.BatchNorm1d(264), torch.nn.ReLU(),
                              sp.SparseConv2d(264, 130, 3, 1, 0, 1, 1, True)).to(DEV)
    net.load_state_dict(ref.state_dict())
    fr = feat.float().requires_grad_(True)
    fg = feat.to(DEV).requires_grad_(True)
    yr = ref(osp.SparseConvTensor(fr, torch.from_numpy(idx), list(shape), B))
    yg = net(sp.SparseConvTensor(fg, torch.from_numpy(idx).to(DEV), list(shape), B))
    def rel(a,b):
        a, b = a.detach().float().cpu().numpy(), b.detach().float().numpy()
        return float(np.abs(a-b).max()/np.abs(b).max()), float(np.linalg.norm(a-b)/np.linalg.norm(b))
    print(dtype, "wide" if wide else "narrow", "fwd", rel(yg.features, yr.features))
    g = torch.from_numpy(rng.standard_normal(tuple(yr.features.shape)).astype(np.float32))
    yr.features.backward(g)
    yg.features.backward(g.to(DEV).to(dtype))
    print("   dX", rel(fg.grad, fr.grad))
    for (name, a), (_n, b) in zip(net.named_parameters(), ref.named_parameters()):
        print("  ", name, rel(a.grad, b.grad))

import faulthandler, os, sys, time, ctypes
faulthandler.dump_traceback_later(60, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from waveformml_amd import _lib
from waveformml_amd.psd import synthetic
from test_gpu_parity import _c2_module
DEV = "cuda:0"
def P(*a): print(*a, flush=True)
lib = _lib.load()
class Wrap:
    def __init__(self, name, fn): self.name, self.fn = name, fn
    def __call__(self, *a):
        r = self.fn(*a)
        if self.name.startswith("wfs_") and "bytes" not in self.name and self.name not in ("wfs_last_error", "wfs_geometry_init"):
            torch.cuda.synchronize(); P("  ok", self.name)
        return r
for name in list(_lib.SIGNATURES):
    setattr(lib, name, Wrap(name, getattr(lib, name)))
T, B = 64, 24
mod = _c2_module(T, 32 * 10 * 7 * 4).to(DEV)
cap = 8192
coords = torch.zeros((cap, 4), dtype=torch.int32, device=DEV); feats = torch.zeros((cap, 2), device=DEV)
nv = torch.zeros((1,), dtype=torch.int64, device=DEV)
for m in mod.modules():
    if hasattr(m, "subm") and not m.subm: m.out_capacity = 8192
for s in (5, 6):
    c, f, y = synthetic.generate(B, T, 3, seed=s)
    n = len(c); coords[:n] = torch.from_numpy(c).to(DEV); feats[:n] = torch.from_numpy(f).to(DEV); nv.fill_(n)
    P("batch", s, "n", n)
    loss = mod.training_step(([coords, feats, nv], torch.from_numpy(y).to(DEV)), 0)
    torch.cuda.synchronize(); P(" fwd done", float(loss))
    loss.backward(); torch.cuda.synchronize(); P(" bwd done")
P("done")

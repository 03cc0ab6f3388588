"""Where does the C4 fp32 step (seeds of tests/test_gpu_fullsize.py) lose the 3rd SubM layer's filter gradient?"""
import copy, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from waveformml_amd.psd import synthetic
from waveformml_amd.psd.config import DictionaryUtility
from waveformml_amd.psd.lit import LitPSD
from waveformml_amd.spconv import functional as Fsp
cfg = json.load(open(os.path.join(ROOT, "config", "psd_c4_deep_fp16.json")))
torch.manual_seed(21)
gpu = LitPSD(DictionaryUtility.to_object(copy.deepcopy(cfg)))
rc = copy.deepcopy(cfg)
rc["net_config"]["imports"] = ["oracle.spconv" if m == "waveformml_amd.spconv" else m for m in rc["net_config"]["imports"]]
c64 = LitPSD(DictionaryUtility.to_object(rc)).double()
c64.load_state_dict(gpu.state_dict())
gpu = gpu.to("cuda:0"); gpu.train(); c64.train()
c, f, y = synthetic.generate(64, 512, 3, seed=99)
print("rows", len(c), "rows % 32 =", len(c) % 32)
# capture the dW call of layer 6
calls = []
orig = Fsp.gather_dw
def spy(table, K, identity_k, R, S, G, swap, *a, **kw):
    out = orig(table, K, identity_k, R, S, G, swap, *a, **kw)
    calls.append((table, K, identity_k, R, S.detach().clone(), G.detach().clone(), swap, out.detach().clone()))
    return out
Fsp.gather_dw = spy
lg = gpu.training_step(([torch.from_numpy(c).cuda(), torch.from_numpy(f).cuda()], torch.from_numpy(y).cuda()), 0)
l64 = c64.training_step(([torch.from_numpy(c), torch.from_numpy(f).double()], torch.from_numpy(y)), 0)
lg.backward(); l64.backward()
torch.set_num_threads(16)
names = [n for n, _ in gpu.model.named_parameters()]
for (n, a), t in zip(gpu.model.named_parameters(), c64.model.parameters()):
    if "weight" in n and a.dim() == 5:
        tr = t.grad.numpy(); ga = a.grad.cpu().numpy().astype(np.float64)
        err = np.abs(ga - tr) / np.abs(tr).max()
        k = np.unravel_index(np.argmax(err), err.shape)
        print("%-22s max err %.3e at %s; entries > 1e-5: %d of %d" % (n, err.max(), k, int((err > 1e-5).sum()), err.size))
# recompute every captured dW in fp64 from the kernel's own inputs
print("dW calls:", len(calls))
for i, (table, K, ident, R, S, G, swap, out) in enumerate(calls):
    if S.shape[1] != 32 or G.shape[1] != 32:
        continue
    Sd, Gd = S.double(), G.double()
    ref = torch.zeros(K, 32, 32, dtype=torch.float64, device=S.device)
    for k in range(K):
        nb = torch.arange(R, device=S.device) if k == ident else table[k].long()
        ok = nb >= 0
        ref[k] = Sd[:R][ok].t() @ Gd[nb[ok]]
    got = out.reshape(K, 32, 32).double()
    if swap:
        got = got.transpose(1, 2)
    e = (got - ref).abs() / ref.abs().max()
    print("call %d: R %d, kernel vs fp64 from its own inputs: max %.3e, entries > 1e-5: %d" % (i, R, float(e.max()), int((e > 1e-5).sum())))

# ---- BatchNorm backward of every 32-channel layer, recomputed in fp64 from the kernels' own inputs
print("BatchNorm backward, kernel vs fp64 on the same (x, dy):")
from waveformml_amd import _lib
lib = _lib.load()
mods = list(gpu.model.sparseModel._modules.values())
# forward again to get the raw conv outputs: hook every conv and BN
feats = {}
import waveformml_amd.spconv as sp
hooks = []
for i, m in enumerate(mods):
    if isinstance(m, torch.nn.BatchNorm1d):
        hooks.append(m.register_forward_hook(lambda mod, inp, out, i=i: None))
xs = {}
orig_bn = Fsp.batch_norm_relu
def spy_bn(features, bn, relu, n_dev=None, stats=None):
    out = orig_bn(features, bn, relu, n_dev, stats)
    xs[id(bn)] = features.detach().clone()
    out.register_hook(lambda g, key=id(bn): xs.__setitem__(("dy", key), g.detach().clone()))
    return out
Fsp.batch_norm_relu = spy_bn
gpu.zero_grad()
lg = gpu.training_step(([torch.from_numpy(c).cuda(), torch.from_numpy(f).cuda()], torch.from_numpy(y).cuda()), 0)
lg.backward()
for i, m in enumerate(mods):
    if not isinstance(m, torch.nn.BatchNorm1d) or id(m) not in xs:
        continue
    x, dy = xs[id(m)].double(), xs[("dy", id(m))].double()
    N = x.shape[0]
    mean, var = x.mean(0), x.var(0, unbiased=False)
    inv = (var + m.eps).rsqrt()
    xh = (x - mean) * inv
    ga, be = m.weight.detach().double(), m.bias.detach().double()
    g = dy * ((ga * xh + be) > 0)
    k1, k2 = g.mean(0), (g * xh).mean(0)
    dx_ref = ga * inv * (g - k1 - xh * k2)
    # the kernel's own result on the same inputs
    x32, dy32 = xs[id(m)], xs[("dy", id(m))]
    sm = x32.float().mean(0); sv = (x32.float().var(0, unbiased=False) + m.eps).rsqrt()
    dx, dga, dbe = Fsp.bn_relu_backward(x32, dy32, m.weight.detach(), m.bias.detach(), mean.float(), inv.float(), True, True, None)
    e = (dx.double() - dx_ref).abs().max(0).values / dx_ref.abs().max()
    print("  BN %2d: max err of dx %.3e (worst channel %d, invstd there %.3e, max invstd %.3e); dgamma err %.3e" % (
        i, float(e.max()), int(e.argmax()), float(inv[int(e.argmax())]), float(inv.max()),
        float((dga.double() - (g * xh).sum(0)).abs().max() / (g * xh).sum(0).abs().max())))

# ---- dX launches (gather_conv with the transposed filters) against fp64 from their own inputs
print("dX, kernel vs fp64 on the same inputs:")
Fsp.batch_norm_relu = orig_bn
gcalls = []
orig_gc = Fsp.gather_conv
def spy_gc(table, kmap, K, identity_k, R, X, W, transpose_w, bias, *a, **kw):
    out = orig_gc(table, kmap, K, identity_k, R, X, W, transpose_w, bias, *a, **kw)
    if X.shape[1] == 32 and out.shape[1] == 32:
        gcalls.append((table, kmap, K, identity_k, R, X.detach().clone(), W.detach().clone(), transpose_w, out.detach().clone()))
    return out
Fsp.gather_conv = spy_gc
gpu.zero_grad()
lg = gpu.training_step(([torch.from_numpy(c).cuda(), torch.from_numpy(f).cuda()], torch.from_numpy(y).cuda()), 0)
lg.backward()
for i, (table, kmap, K, ident, R, X, W, tr, out) in enumerate(gcalls):
    Xd, Wd = X.double(), W.double()
    ref = torch.zeros(R, 32, dtype=torch.float64, device=X.device)
    for k in range(K):
        kk = kmap[k] if kmap is not None else k
        nb = torch.arange(R, device=X.device) if k == ident else table[kk].long()
        ok = nb >= 0
        Wk = Wd[k].t() if tr else Wd[k]
        ref[ok] += Xd[nb[ok]] @ Wk
    e = (out.double() - ref).abs()
    rowscale = ref.abs().max()
    print("  %s call %2d: R %6d max abs err / max|ref| %.3e ; worst row's own scale %.3e vs global %.3e" % (
        "dX " if tr else "fwd", i, R, float(e.max() / rowscale), float(ref[int(e.max(1).values.argmax())].abs().max()), float(rowscale)))

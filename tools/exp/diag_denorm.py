import sys, numpy as np, torch
sys.path.insert(0, ".")
from waveformml_amd.spconv import functional as Fsp
DEV = "cuda:0"
rng = np.random.default_rng(0)
K, R, XR, Cx, Cy = 9, 420, 1900, 130, 264
X = rng.standard_normal((XR, Cx)).astype(np.float32)
t = rng.integers(0, XR, size=(K, R)).astype(np.int32); t[rng.random((K, R)) > 0.5] = -1
for scale in (0.05, 0.003, 0.0003):
    W = (rng.uniform(-1, 1, (K, Cy, Cx)) * scale).astype(np.float32)
    for dtype in (torch.float16, torch.bfloat16):
        Y = Fsp.gather_conv(torch.from_numpy(t).to(DEV), None, K, -1, R, torch.from_numpy(X).to(DEV).to(dtype), torch.from_numpy(W).to(DEV), True, None)
        Xr = torch.from_numpy(X).to(dtype).double().numpy()
        Wr = torch.from_numpy(W).to(dtype).double().numpy()
        Wf = Wr.copy()
        if dtype == torch.float16:
            Wf[np.abs(Wf) < 2.0 ** -14] = 0
        def ref(Wm):
            out = np.zeros((R, Cy))
            for k in range(K):
                ok = t[k] >= 0
                out[ok] += Xr[t[k][ok]] @ Wm[k].T
            return out
        got = Y.double().cpu().numpy()
        for name, Wm in (("rounded", Wr), ("denormals flushed", Wf), ("fp32 filters", W.astype(np.float64))):
            want = ref(Wm)
            print("scale %g %s vs %-18s rel l2 %.2e" % (scale, str(dtype)[6:], name, np.linalg.norm(got - want) / np.linalg.norm(want)))

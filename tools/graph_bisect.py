"""Which pieces of the step survive HIP-graph capture?  Each piece runs in its own process.
usage: python tools/graph_bisect.py            (driver)   |   python tools/graph_bisect.py <piece>"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PIECES = ["noop", "subm_rulebook", "conv_rulebook", "gconv_f32", "gconv_bf16", "gconv_c2", "bn", "dense", "dw_f32",
          "dw_bf16", "forward", "fwd_bwd", "full"]

def run(piece):
    import faulthandler
    faulthandler.dump_traceback_later(40, exit=True)
    import numpy as np, torch
    from waveformml_amd.psd import synthetic
    from waveformml_amd.spconv import ops, functional as Fsp
    import waveformml_amd.spconv as sp
    dev = torch.device("cuda:0")
    c, f, y = synthetic.generate(16, 64, 3, seed=1)
    n = len(c); cap = n + 100
    idx = torch.zeros((cap, 4), dtype=torch.int32, device=dev); idx[:n] = torch.from_numpy(np.ascontiguousarray(c[:, [3, 0, 1, 2]])).to(dev)
    nv = torch.tensor([n], dtype=torch.int64, device=dev)
    X = torch.randn(cap, 32, device=dev); W = torch.randn(27, 32, 32, device=dev) * 0.1
    X2 = torch.randn(cap, 2, device=dev); W2 = torch.randn(27, 2, 32, device=dev)
    rb = ops.build_rulebook(idx, 16, [14, 11, 64], [3] * 3, [1] * 3, [0] * 3, [1] * 3, True, n_dev=nv)
    t, km = rb.table_by_out()
    bn = torch.nn.BatchNorm1d(32).to(dev)
    def body():
        if piece == "noop":
            return X * 2
        if piece == "subm_rulebook":
            return ops.build_rulebook(idx, 16, [14, 11, 64], [3] * 3, [1] * 3, [0] * 3, [1] * 3, True, n_dev=nv).nbr_out
        if piece == "conv_rulebook":
            return ops.build_rulebook(idx, 16, [14, 11, 64], [3] * 3, [1, 1, 4], [0] * 3, [1] * 3, False, n_dev=nv).nbr_in
        if piece == "gconv_f32":
            return Fsp.gather_conv(t, km, 27, 13, cap, X, W, False, None, nv)
        if piece == "gconv_bf16":
            return Fsp.gather_conv(t, km, 27, 13, cap, X.bfloat16(), W, False, None, nv)
        if piece == "gconv_c2":
            return Fsp.gather_conv(t, km, 27, 13, cap, X2, W2, False, None, nv)
        if piece == "bn":
            return Fsp.batch_norm_relu(X, bn, True, nv)
        if piece == "dense":
            return Fsp.ToDenseFunction.apply(X, idx, [14, 11, 64], 16, True, nv)
        if piece == "dw_f32":
            return Fsp.gather_dw(rb.nbr_out, 27, 13, cap, X, X, False, None, nv)
        if piece == "dw_bf16":
            return Fsp.gather_dw(rb.nbr_out, 27, 13, cap, X.bfloat16(), X.bfloat16(), False, None, nv)
        raise SystemExit("unknown piece")
    if piece.startswith("class"):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import json, copy
        from waveformml_amd.psd.config import DictionaryUtility
        from waveformml_amd.psd.lit import LitPSD
        from waveformml_amd.psd.ddp import FlatGradAllReducer
        from waveformml_amd.psd import graph as G
        cfg = json.load(open(os.path.join(ROOT, "config", "psd_c2_3d.json")))
        cfg["system_config"]["n_samples"] = 64
        cfg["net_config"]["algorithm"][-1] = [32 * 10 * 7 * 4, 3]
        mod = LitPSD(DictionaryUtility.to_object(cfg)).to(dev)
        red = FlatGradAllReducer(mod.model.parameters(), world_size=1)
        mod.optimizer_parameters = red.optimizer_parameters()
        opt = mod.configure_optimizers()[0][0]
        batch = ([torch.from_numpy(c).to(dev), torch.from_numpy(f).to(dev)], torch.from_numpy(y).to(dev))
        if piece == "class_defaultcap":
            G._round_up_orig = G._round_up
        step = G.GraphedTrainStep(mod, opt, red, batch, headroom=(1.25 if piece != "class_defaultcap" else 3.0))
        print("captured", flush=True)
        loss = step(batch); step.check()
        print("OK", piece, float(loss))
        return
    if piece in ("forward", "fwd_bwd", "full"):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import json, copy
        from waveformml_amd.psd.config import DictionaryUtility
        from waveformml_amd.psd.lit import LitPSD
        from waveformml_amd.psd.ddp import FlatGradAllReducer
        cfg = json.load(open(os.path.join(ROOT, "config", "psd_c2_3d.json")))
        cfg["system_config"]["n_samples"] = 64
        cfg["net_config"]["algorithm"][-1] = [32 * 10 * 7 * 4, 3]
        mod = LitPSD(DictionaryUtility.to_object(cfg)).to(dev)
        red = FlatGradAllReducer(mod.model.parameters(), world_size=1)
        mod.optimizer_parameters = red.optimizer_parameters()
        opt = mod.configure_optimizers()[0][0]
        coords = torch.zeros((cap, 4), dtype=torch.int32, device=dev); coords[:n] = torch.from_numpy(c).to(dev)
        feats = torch.zeros((cap, 2), device=dev); feats[:n] = torch.from_numpy(f).to(dev)
        labels = torch.from_numpy(y).to(dev)
        def body():
            red.reset()
            loss = mod.training_step(([coords, feats, nv], labels), 0)
            if piece == "forward":
                return loss.detach()
            loss.backward()
            red.pack_all()
            if piece == "full":
                opt.step()
            return loss.detach()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            body()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = body()
    for _ in range(3):
        g.replay(); torch.cuda.synchronize()
    print("OK", piece, float(out.float().abs().sum()) if torch.is_tensor(out) else "")

if len(sys.argv) > 1:
    run(sys.argv[1])
else:
    for p in PIECES:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), p], capture_output=True, text=True, timeout=180)
        last = (r.stdout.strip().splitlines() or [""])[-1]
        err = [l for l in r.stderr.splitlines() if "Error" in l or "error" in l][-2:]
        print("%-14s rc=%4d  %s %s" % (p, r.returncode, last, " | ".join(err)[:200]), flush=True)

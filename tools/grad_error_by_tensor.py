"""What 16-bit rows do to the gradients and to training (verdict r2 item 6; run on the GPU box).

1. Per-tensor relative L2 error  ||g_16bit - g_fp64|| / ||g_fp64||  of every parameter gradient of one training step,
   bf16 and fp16 rows on the GPU against the CPU restatement run in FLOAT64 on the same (rounded) inputs, for
   BASELINE configs[1] (C2: 256 events x 256 samples) and configs[3] (C4: 64 events x 512 samples, eight conv layers).
2. 200 captured training steps on 8 cycling batches, bf16 rows against fp32 rows from the same initial weights: the two
   loss curves side by side (what "3.8e-3 on the logits" means for training).

-> one JSON object (profiles/r03_grad_error_by_tensor.json).  tests/test_gpu_fullsize.py's GRAD_REL_L2 bounds are 1.5 x
the maxima measured here.
usage: python tools/grad_error_by_tensor.py [steps]"""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from waveformml_amd.psd import synthetic
from waveformml_amd.psd.config import DictionaryUtility

DEV = "cuda:0"
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def grad_errors(cfg_path, events, samples, seed, dtype):
    from test_gpu_fullsize import _pair
    with open(os.path.join(ROOT, "config", cfg_path)) as fh:
        cfg = json.load(fh)
    torch.manual_seed(1234)
    gpu, cpu = _pair(cfg, DictionaryUtility.to_object)
    c, f, y = synthetic.generate(events, samples, 3, seed=seed)
    fin = torch.from_numpy(f).to(dtype)
    truth = cpu.make_twin().double()
    truth.load_state_dict(cpu.state_dict())
    truth.train()
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    lt = truth.training_step(([torch.from_numpy(c), fin.double()], torch.from_numpy(y)), 0)
    lt.backward()
    lg = gpu.training_step(([torch.from_numpy(c).to(DEV), fin.to(DEV)], torch.from_numpy(y).to(DEV)), 0)
    lg.backward()
    out = {}
    gmax = max(float(p.grad.abs().max()) for p in truth.model.parameters() if p.grad is not None)
    for (name, a), b in zip(gpu.model.named_parameters(), truth.model.parameters()):
        if b.grad is None:
            continue
        t = b.grad.double()
        if float(t.abs().max()) < 1e-7 * gmax:
            continue                                  # zero in exact arithmetic (a bias in front of a BatchNorm)
        out[name] = float((a.grad.double().cpu() - t).norm() / t.norm())
    return {"voxels": int(len(c)), "loss_16bit": float(lg), "loss_fp64": float(lt), "rel_l2": out,
            "max_rel_l2": max(out.values())}


def loss_curves(n_steps):
    from waveformml_amd.psd.ddp import FlatGradAllReducer
    from waveformml_amd.psd.graph import GraphedTrainStep
    from waveformml_amd.psd.lit import LitPSD
    with open(os.path.join(ROOT, "config", "psd_c2_3d.json")) as fh:
        cfg = json.load(fh)
    batches = [synthetic.generate(256, 256, 3, seed=1234 + i) for i in range(8)]
    big = max(range(8), key=lambda i: len(batches[i][0]))
    curves = {}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        torch.manual_seed(0)
        mod = LitPSD(DictionaryUtility.to_object(copy.deepcopy(cfg))).to(DEV)
        red = FlatGradAllReducer(mod.model.parameters(), world_size=1)
        mod.optimizer_parameters = red.optimizer_parameters()
        opt = mod.configure_optimizers()
        opt = opt[0][0] if isinstance(opt, tuple) else opt
        dev_b = [([torch.from_numpy(c).to(DEV), torch.from_numpy(f).to(DEV).to(dt)], torch.from_numpy(y).to(DEV))
                 for c, f, y in batches]
        start = red.flat_param.detach().clone()
        bufs = [t.detach().clone() for t in mod.buffers()]
        step = GraphedTrainStep(mod, opt, red, dev_b[big])
        with torch.no_grad():                   # undo the capture's calibration / warm-up steps (as Trainer._capture)
            red.flat_param.copy_(start)
            for t, q in zip(mod.buffers(), bufs):
                t.copy_(q)
            for st in opt.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
        losses = [float(step(dev_b[i % 8])) for i in range(n_steps)]
        step.check()
        curves[name] = losses
    a, b = np.asarray(curves["f32"]), np.asarray(curves["bf16"])
    tail = slice(max(0, n_steps - 40), n_steps)
    return {"steps": n_steps, "batches": 8, "loss_f32_every_10": [round(float(v), 5) for v in a[::10]],
            "loss_bf16_every_10": [round(float(v), 5) for v in b[::10]],
            "first_step_rel_diff": float(abs(a[0] - b[0]) / abs(a[0])),
            "max_abs_diff": float(np.abs(a - b).max()), "mean_abs_diff": float(np.abs(a - b).mean()),
            "mean_loss_last_40": {"f32": float(a[tail].mean()), "bf16": float(b[tail].mean())},
            "final_loss": {"f32": float(a[-1]), "bf16": float(b[-1])}}


def main():
    out = {"grad_rel_l2_vs_fp64": {}}
    for tag, cfgp, ev, T, seed in (("C2_256x256", "psd_c2_3d.json", 256, 256, 1234), ("C4_64x512", "psd_c4_deep_fp16.json", 64, 512, 99)):
        for dn, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
            r = grad_errors(cfgp, ev, T, seed, dt)
            out["grad_rel_l2_vs_fp64"]["%s_%s" % (tag, dn)] = r
            print(tag, dn, "max rel L2 %.4f" % r["max_rel_l2"], flush=True)
    out["bound_1p5x_max"] = {k: round(1.5 * v["max_rel_l2"], 3) for k, v in out["grad_rel_l2_vs_fp64"].items()}
    out["loss_curves_bf16_vs_f32"] = loss_curves(steps)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

"""How well-conditioned are the whole-net gradients at bench size?  C2 (256 events x 256 samples), one training step:
GPU fp32 rows vs the CPU oracle in fp32 (the reference's arithmetic) vs the SAME oracle in fp64 (ground truth).
Prints, per parameter, max|a - b| / max|truth| for (gpu, f64), (cpu32, f64), (gpu, cpu32).
usage: python tools/exp/grad_conditioning.py [events] [samples] [config]"""
import copy, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from waveformml_amd.psd import synthetic
from waveformml_amd.psd.config import DictionaryUtility
from waveformml_amd.psd.lit import LitPSD

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "config", "psd_c2_3d.json")
cfg = json.load(open(path))
assert cfg["system_config"]["n_samples"] == T
torch.manual_seed(1234)
gpu = LitPSD(DictionaryUtility.to_object(copy.deepcopy(cfg)))
rc = copy.deepcopy(cfg)
rc["net_config"]["imports"] = ["oracle.spconv" if m == "waveformml_amd.spconv" else m for m in rc["net_config"]["imports"]]
c32 = LitPSD(DictionaryUtility.to_object(copy.deepcopy(rc)))
c32.load_state_dict(gpu.state_dict())
c64 = LitPSD(DictionaryUtility.to_object(copy.deepcopy(rc))).double()
c64.load_state_dict(gpu.state_dict())
gpu = gpu.to("cuda:0")
for m in (gpu, c32, c64):
    m.train()
c, f, y = synthetic.generate(B, T, 3, seed=1234)
torch.set_num_threads(16)
lg = gpu.training_step(([torch.from_numpy(c).cuda(), torch.from_numpy(f).cuda()], torch.from_numpy(y).cuda()), 0)
l32 = c32.training_step(([torch.from_numpy(c), torch.from_numpy(f)], torch.from_numpy(y)), 0)
l64 = c64.training_step(([torch.from_numpy(c), torch.from_numpy(f).double()], torch.from_numpy(y)), 0)
print("loss gpu %.9f cpu32 %.9f cpu64 %.12f" % (lg.item(), l32.item(), l64.item()))
lg.backward(); l32.backward(); l64.backward()
print("%-28s %12s %12s %12s" % ("parameter", "gpu-f64", "cpu32-f64", "gpu-cpu32"))
for (n, a), b, t in zip(gpu.model.named_parameters(), c32.model.parameters(), c64.model.parameters()):
    tr = t.grad.numpy()
    s = max(float(np.abs(tr).max()), 1e-300)
    ga, gb = a.grad.cpu().numpy().astype(np.float64), b.grad.numpy().astype(np.float64)
    print("%-28s %12.3e %12.3e %12.3e" % (n, np.abs(ga - tr).max() / s, np.abs(gb - tr).max() / s, np.abs(ga - gb).max() / s))

"""Per-tensor fp32 gradient errors of the hybrid net (the test_c5_hybrid_net_at_config_size setup) against the fp64 oracle,
beside the fp32 CPU oracle's own errors.  env WFS_WIDE_MIN_CHANNELS=0 -> 32 x 32-tile kernels."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_fullsize as T
from waveformml_amd.psd import synthetic
from waveformml_amd.psd.config import load_config
cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "gep_config.json")))
cfg["system_config"]["n_samples"] = 1024
cfg["net_config"]["hparams"]["n_dil"] = 3
cfg["net_config"]["hparams"]["wf_params"]["dropout"] = 0.0
torch.manual_seed(21)
gpu, cpu = T._pair(cfg, load_config)
with torch.no_grad():
    for p in gpu.model.waveformLayer.parameters():
        p.copy_(torch.randn_like(p) * 0.5)
cpu.load_state_dict({k: v.cpu() for k, v in gpu.state_dict().items()})
c, f, y = synthetic.generate(64, 1024, 3, seed=3, layout="2d")
try:
    rep = T._one_step(gpu, cpu, c, f, y, torch.float32, 3e-5, 1.0)
except AssertionError as e:
    print("assert:", e); rep = []
for name, e_gpu, e_ref, e_pair, l2_gpu, l2_ref in rep:
    print("%-40s max-norm: gpu vs fp64 %.2e  cpu-fp32 vs fp64 %.2e  gpu vs cpu-fp32 %.2e   L2: gpu %.2e  cpu-fp32 %.2e" % (name, e_gpu, e_ref, e_pair, l2_gpu, l2_ref))

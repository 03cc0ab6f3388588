"""How much of the fp16 path's gradient error on the deep C4 stack is fp16 underflow?  Same step as
tests/test_gpu_parity.py::test_c4_deep_stack_config_matches_the_cpu_path[f16], backward run at several loss scales;
prints, per scale, the worst parameter-gradient error relative to that tensor's scale."""
import copy, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from waveformml_amd.psd import synthetic
from waveformml_amd.psd.config import DictionaryUtility
from waveformml_amd.psd.lit import LitPSD

T, B = int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 24
cfg = json.load(open(os.path.join(ROOT, "config", "psd_c4_deep_fp16.json")))
cfg["system_config"]["n_samples"] = T
cfg["net_config"]["algorithm"][-1] = [32 * 10 * 7 * (((T - 3) // 4 + 1 - 3) // 4 + 1), 3]
torch.manual_seed(21)
gpu = LitPSD(DictionaryUtility.to_object(copy.deepcopy(cfg)))
ref_cfg = copy.deepcopy(cfg)
ref_cfg["net_config"]["imports"] = ["oracle.spconv" if m == "waveformml_amd.spconv" else m for m in ref_cfg["net_config"]["imports"]]
cpu = LitPSD(DictionaryUtility.to_object(ref_cfg))
cpu.load_state_dict(gpu.state_dict())
gpu = gpu.to("cuda:0")
gpu.train(), cpu.train()
c, f, y = synthetic.generate(B, T, 3, seed=99)
fin = torch.from_numpy(f).half()
cpu.training_step(([torch.from_numpy(c), fin.float()], torch.from_numpy(y)), 0).backward()
for dt in (torch.float16, torch.bfloat16):
    for scale in (1.0, 64.0, 1024.0, 65536.0):
        gpu.zero_grad()
        loss = gpu.training_step(([torch.from_numpy(c).cuda(), fin.cuda().to(dt)], torch.from_numpy(y).cuda()), 0)
        (loss * scale).backward()
        worst, name_w = 0.0, ""
        for (name, a), b in zip(gpu.model.named_parameters(), cpu.model.parameters()):
            g = a.grad.float().cpu() / scale
            err = float((g - b.grad).abs().max() / b.grad.abs().max())
            if err > worst:
                worst, name_w = err, name
        if scale == 1.0:
            for (name, a), b in zip(gpu.model.named_parameters(), cpu.model.parameters()):
                print("   %-28s rel L2 %.3e" % (name, float((a.grad.float().cpu() - b.grad).norm() / b.grad.norm().clamp_min(1e-30))))
        print("%s scale %8.0f  worst grad error %.3e of scale (%s)  finite %s" % (
            dt, scale, worst, name_w, all(bool(torch.isfinite(p.grad).all()) for p in gpu.model.parameters())))

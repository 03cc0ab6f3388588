"""Inference loops, eager vs captured (psd/graph.GraphedEvalStep): test_loop over device-resident 256-event batches of
the 3-D C2 net (bf16 rows) and the occlusion sweep of the 2-D GEP net (fp32, T = 150, 60 occluded columns).
usage: python tools/bench_eval.py"""
import copy, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from waveformml_amd.psd import synthetic
from waveformml_amd.psd.config import DictionaryUtility, load_config
from waveformml_amd.psd.evaluate import occlusion_sweep, test_loop
from waveformml_amd.psd.lit import LitPSD
from waveformml_amd.spconv import ops

dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream())
ops.ASSUME_VALID_UNIQUE_INDICES = True
out = {}
cfg = json.load(open(os.path.join(ROOT, "config", "psd_c2_3d.json")))
torch.manual_seed(0)
mod = LitPSD(DictionaryUtility.to_object(copy.deepcopy(cfg))).to(dev)
batches = []
for s in range(12):
    c, f, y = synthetic.generate(256, 256, 3, seed=500 + s)
    batches.append(([torch.from_numpy(c).to(dev), torch.from_numpy(f).to(dev).bfloat16()], torch.from_numpy(y).to(dev)))
for capture in (False, True):
    test_loop(mod, batches[:2], dev, capture=capture)
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = test_loop(mod, batches * 16, dev, capture=capture)          # one loop of 192 batches (one capture)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / (16 * len(batches))
    out["test_loop_c2_%s_ms_per_batch" % ("captured" if capture else "eager")] = round(dt * 1e3, 3)
    out["test_loop_c2_%s_events_per_s" % ("captured" if capture else "eager")] = round(256 / dt)
gcfg = json.load(open(os.path.join(ROOT, "tests", "golden", "gep_config.json")))
gep = LitPSD(load_config(copy.deepcopy(gcfg))).to(dev)
c, f, y = synthetic.generate(256, 150, 3, seed=4, layout="2d")
batch = ([torch.from_numpy(c).to(dev), torch.from_numpy(f).to(dev)], torch.from_numpy(y).to(dev))
idxs = [None] + list(range(1, 300, 5))
for capture in (False, True):
    occlusion_sweep(gep, batch, idxs[:3], capture=capture)
    torch.cuda.synchronize()
    t = time.perf_counter()
    occlusion_sweep(gep, batch, idxs, capture=capture)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / len(idxs)
    out["occlusion_sweep_gep_%s_ms_per_index" % ("captured" if capture else "eager")] = round(dt * 1e3, 3)
print(json.dumps(out))

"""Host-side cost of replaying the captured training step: time of graph.replay() without synchronising, against the
GPU time of the step.  usage: python tools/exp/replay_host_time.py"""
import os, sys, time, json, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from waveformml_amd.psd import synthetic
from waveformml_amd.psd.config import DictionaryUtility
from waveformml_amd.psd.ddp import FlatGradAllReducer
from waveformml_amd.psd.graph import GraphedTrainStep
from waveformml_amd.psd.lit import LitPSD
from waveformml_amd.spconv import ops
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(dev))
ops.ASSUME_VALID_UNIQUE_INDICES = True
ops.PREFETCH_RULEBOOKS = os.environ.get("WFS_PREFETCH", "1") != "0"
cfg = json.load(open(os.path.join(ROOT, "config", "psd_c2_3d.json")))
torch.manual_seed(0)
mod = LitPSD(DictionaryUtility.to_object(copy.deepcopy(cfg))).to(dev)
red = FlatGradAllReducer(mod.model.parameters(), world_size=1)
mod.optimizer_parameters = red.optimizer_parameters()
opt = mod.configure_optimizers()[0][0]
c, f, y = synthetic.generate(256, 256, 3, seed=1234)
batch = ([torch.from_numpy(c).to(dev), torch.from_numpy(f).to(dev).to(torch.bfloat16)], torch.from_numpy(y).to(dev))
step = GraphedTrainStep(mod, opt, red, batch)
for _ in range(5):
    step(batch)
torch.cuda.synchronize()
g = step.graph
hs = []
for _ in range(50):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    hs.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    g.replay()
torch.cuda.synchronize()
per = (time.perf_counter() - t0) / 200
print("host time of one replay() call on an idle GPU: median %.1f us (min %.1f); back-to-back replays: %.1f us per step"
      % (np.median(hs) * 1e6, min(hs) * 1e6, per * 1e6))

"""cProfile of the training step's host side (run on the GPU box): python tools/host_profile.py"""
import cProfile, copy, json, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from waveformml_amd.psd import synthetic
from waveformml_amd.psd.config import DictionaryUtility
from waveformml_amd.psd.ddp import FlatGradAllReducer
from waveformml_amd.psd.lit import LitPSD

dev = torch.device("cuda:0")
cfg = bench.load_cfg(os.path.join(ROOT, "config", "psd_c2_3d.json"), 256)
module = LitPSD(DictionaryUtility.to_object(copy.deepcopy(cfg))).to(dev).train()
opt = module.configure_optimizers()[0][0]
reducer = FlatGradAllReducer(module.model.parameters())
c, f, y = synthetic.generate(256, 256, 3, seed=1234)
batch = ([torch.from_numpy(c).to(dev), torch.from_numpy(f).to(dev)], torch.from_numpy(y).to(dev))

def step():
    reducer.reset()
    loss = module.training_step(batch, 0)
    loss.backward()
    reducer.finish()
    opt.step()

for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host-only %.3f ms/step, with final sync %.3f ms/step" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)

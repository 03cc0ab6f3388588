"""Where the HOST spends a step of Trainer(capture=True) on host-resident batches: cProfile of the second and third epoch.
usage: python tools/exp/trainer_host_profile.py"""
import copy, cProfile, json, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from waveformml_amd.psd import data
from waveformml_amd.psd.config import DictionaryUtility
from waveformml_amd.psd.lit import LitPSD
from waveformml_amd.psd.trainer import Trainer
from waveformml_amd.spconv import ops
cfg = json.load(open(os.path.join(ROOT, "config", "psd_c2_3d.json")))
ops.ASSUME_VALID_UNIQUE_INDICES = True
ops.PREFETCH_RULEBOOKS = True
torch.manual_seed(0)
mod = LitPSD(DictionaryUtility.to_object(copy.deepcopy(cfg)))
ds = data.SyntheticPulseDataset(120, 256, 256, n_type=3, layout="3d", seed=4242)
batches = [b for b in data.make_loader(ds, 1, shuffle=False, pin_memory=True)]
tr = Trainer(max_epochs=1, device="cuda:0", feature_dtype=torch.bfloat16, capture=True, check_every=25, log_every=0)
tr.fit(mod, batches)                       # capture + first epoch
tr.max_epochs = 4
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
hist = tr.fit(mod, batches)
pr.disable()
torch.cuda.synchronize()
print("ms per step over the profiled epochs:", [round(h["train_seconds"] / h["steps"] * 1e3, 3) for h in hist])
pstats.Stats(pr).sort_stats("tottime").print_stats(18)

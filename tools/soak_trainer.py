"""Soak run of the captured training loop: Trainer(capture=True) over many DIFFERENT synthetic 256-event batches (their
voxel counts vary by ~3 %), bf16 rows.  Prints steps/s, the loss trajectory, eager fallbacks and overflow checks.
usage: python tools/soak_trainer.py [n_batches] [epochs]"""
import copy, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from waveformml_amd.psd import data
from waveformml_amd.psd.config import DictionaryUtility
from waveformml_amd.psd.lit import LitPSD
from waveformml_amd.psd.trainer import Trainer
from waveformml_amd.spconv import ops

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 120
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = json.load(open(os.path.join(ROOT, "config", "psd_c2_3d.json")))
ops.ASSUME_VALID_UNIQUE_INDICES = True
ops.PREFETCH_RULEBOOKS = True
torch.manual_seed(0)
mod = LitPSD(DictionaryUtility.to_object(copy.deepcopy(cfg)))
ds = data.SyntheticPulseDataset(n_batches, 256, 256, n_type=3, layout="3d", seed=4242)
t0 = time.perf_counter()
batches = [b for b in data.make_loader(ds, 1, shuffle=False, pin_memory=True)]
sizes = [int(b[0][0].shape[0]) for b in batches]
print("generated %d batches in %.1f s; voxels min %d max %d first %d" % (len(batches), time.perf_counter() - t0, min(sizes), max(sizes), sizes[0]), flush=True)
tr = Trainer(max_epochs=epochs, device="cuda:0", feature_dtype=torch.bfloat16, capture=True, check_every=25, log_every=0)
t0 = time.perf_counter()
hist = tr.fit(mod, batches)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"steps": epochs * n_batches, "seconds": round(dt, 2), "steps_per_s_incl_capture_and_h2d": round(epochs * n_batches / dt, 1),
                  "n_cap": tr.last_capacity, "eager_fallbacks": tr.eager_fallbacks, "history": hist}))

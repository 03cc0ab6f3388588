import faulthandler, os, sys, time
faulthandler.dump_traceback_later(50, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from waveformml_amd.psd import synthetic
from waveformml_amd.psd.ddp import FlatGradAllReducer
from waveformml_amd.psd.graph import GraphedTrainStep
from test_gpu_parity import _c2_module
DEV = "cuda:0"
def P(*a): print(*a, flush=True)
T, B = 64, 24
batches = []
for s in (5, 6, 7):
    c, f, y = synthetic.generate(B, T, 3, seed=s)
    batches.append(([torch.from_numpy(c).to(DEV), torch.from_numpy(f).to(DEV)], torch.from_numpy(y).to(DEV)))
def make():
    mod = _c2_module(T, 32 * 10 * 7 * 4).to(DEV)
    red = FlatGradAllReducer(mod.model.parameters(), world_size=1)
    mod.optimizer_parameters = red.optimizer_parameters()
    opt = mod.configure_optimizers()[0][0]
    return mod, red, opt
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
mod_g, red_g, opt_g = make()
step = GraphedTrainStep(mod_g, opt_g, red_g, batches[0], warmup=2)
P("captured; caps", step.n_cap, [m.out_capacity for m in step._convs], "N", [b[0][0].shape[0] for b in batches])
if mode in ("both", "eager"):
    mod_e, red_e, opt_e = make()
    for i in range(2):
        red_e.reset(); mod_e.training_step(batches[0], 0).backward(); red_e.finish(); opt_e.step()
        torch.cuda.synchronize(); P("eager", i)
seq = [batches[0]] * 3 if "X" not in mode else batches
if "T" in mode:
    with torch.cuda.stream(step.stream):
        for i, b in enumerate(seq):
            lg = step(b); torch.cuda.synchronize(); P("call", i, float(lg))
            step.check(); P("check", i)
            x = torch.ones(1000, device=DEV) * 3; P("eager op on the same non-default stream", float(x.sum()))
    P("done"); sys.exit(0)
if "S" in mode:
    for i, b in enumerate(seq):
        lg = step(b); torch.cuda.synchronize(); P("call", i, float(lg))
        step.check(); P("check", i)
    P("done"); sys.exit(0)
for i, b in enumerate(seq):
    if "L" in mode:
        step._load(b); torch.cuda.synchronize(); P("loaded", i)
    step.graph.replay(); torch.cuda.synchronize(); P("replay", i, float(step.loss))
    if "C" in mode:
        step.check(); P("check", i)
P("done")

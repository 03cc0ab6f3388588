"""The reference's per-segment classifier (config/examples/IoniClassifierCNN.json: SPConvPreserveNet, six conv -> inverse
conv layers 130 -> 138 -> 146 -> 154 -> 104 -> 54 -> 5 on the 14 x 11 grid, one logit row per active segment) as a
training step on the GPU (eager and as a captured HIP graph; every layer in libwfsparse's shape-generic MFMA kernels) beside the CPU restatement on
the host cores.  A parity case with a timing, not the headline bench.   usage: python tools/bench_ioni.py [events] [steps] [f32|bf16|f16]"""
import copy, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from bench import host_cores
from test_segment_callers import IONI
from waveformml_amd.psd import synthetic
from waveformml_amd.psd.config import load_config
from waveformml_amd.psd.litseg import LitSegClassifier

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dtype = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[3] if len(sys.argv) > 3 else "f32"]
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream())


def build(module):
    cfg = copy.deepcopy(IONI)
    cfg["net_config"]["imports"] = [module if m in ("oracle.spconv", "waveformml_amd.spconv") else m
                                    for m in cfg["net_config"]["imports"]]
    return LitSegClassifier(load_config(cfg))


torch.manual_seed(0)
gpu = build("waveformml_amd.spconv")
cpu = build("oracle.spconv")
cpu.load_state_dict(gpu.state_dict())
gpu = gpu.to(dev)
c, f, _ = synthetic.generate(B, 65, 5, seed=11, layout="2d")          # [rows, 130] waveform rows on the 14 x 11 grid
rng = np.random.default_rng(1)
y = rng.integers(0, 5, len(c))
cg, fg, yg = torch.from_numpy(c).to(dev), torch.from_numpy(f).to(dev).to(dtype), torch.from_numpy(y).to(dev)
cc, fc, yc = torch.from_numpy(c), torch.from_numpy(f), torch.from_numpy(y)
og = torch.optim.SGD(gpu.model.parameters(), lr=0.02, momentum=0.98, nesterov=True)
oc = torch.optim.SGD(cpu.model.parameters(), lr=0.02, momentum=0.98, nesterov=True)


def step(mod, opt, batch):
    opt.zero_grad(set_to_none=True)
    loss = mod.training_step(batch, 0)
    loss.backward()
    opt.step()
    return loss


lg0 = step(gpu, og, ([cg, fg], yg)).item()
lc0 = step(cpu, oc, ([cc, fc], yc)).item()
for _ in range(3):
    step(gpu, og, ([cg, fg], yg))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step(gpu, og, ([cg, fg], yg))
torch.cuda.synchronize()
gpu_ms = (time.perf_counter() - t0) / steps * 1e3
# the same step captured into a HIP graph (psd/graph.GraphedTrainStep: labels padded per row, flat parameters, FlatSGD)
from waveformml_amd.psd.ddp import FlatGradAllReducer
from waveformml_amd.psd.graph import GraphedTrainStep
from waveformml_amd.spconv import ops
ops.ASSUME_VALID_UNIQUE_INDICES = True
torch.manual_seed(0)
gmod = build("waveformml_amd.spconv").to(dev)
red = FlatGradAllReducer(gmod.model.parameters())
gmod.optimizer_parameters = red.optimizer_parameters()
gopt = gmod.configure_optimizers()
gopt = gopt[0][0] if isinstance(gopt, tuple) else gopt
gstep = GraphedTrainStep(gmod, gopt, red, ([cg, fg], yg))
for _ in range(3):
    gstep(([cg, fg], yg))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    gstep(([cg, fg], yg))
torch.cuda.synchronize()
graph_ms = (time.perf_counter() - t0) / steps * 1e3
gstep.check()
torch.set_num_threads(host_cores())
n_cpu = max(3, steps // 10)
t0 = time.perf_counter()
for _ in range(n_cpu):
    step(cpu, oc, ([cc, fc], yc))
cpu_ms = (time.perf_counter() - t0) / n_cpu * 1e3
print(json.dumps({"config": "IoniClassifierCNN.json (SPConvPreserveNet, %s rows)" % str(dtype).split(".")[-1], "events": B, "rows": int(len(c)),
                  "rel_loss_diff_first_step": abs(lg0 - lc0) / abs(lc0), "gpu_eager_ms_per_step": round(gpu_ms, 3),
                  "gpu_graph_ms_per_step": round(graph_ms, 3), "gpu_events_per_s": round(B / graph_ms * 1e3), "cpu_ms_per_step": round(cpu_ms, 2),
                  "cpu_events_per_s": round(B / cpu_ms * 1e3), "cpu_threads": host_cores()}))

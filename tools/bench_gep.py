"""BASELINE.json configs[0] ("C1": the reference's own CPU-runnable example, config/examples/GEP.json -- 2-D SparseConv2d net
300 -> 252 -> 158 -> 64 channels on the 14 x 11 grid, LinearBlock head, T = 150, batch 32, fp32): training-step time of this
library on the GPU (eager and HIP-graph replay) beside the CPU restatement on the host cores.  A parity case with a
timing, not the headline bench.   usage: python tools/bench_gep.py [batch] [steps] [T] [n_dil] [dropout] [dtype]
With T = 1024 and n_dil = 3 this is BASELINE.json configs[4] ("C5"): the hybrid net, TemporalConvNet front end over the
[N, 2T] waveform rows + the sparse 2-D stack the reference's block generator derives for 2048 input channels."""
import copy, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import host_cores
from waveformml_amd.psd import synthetic
from waveformml_amd.psd.config import load_config
from waveformml_amd.psd.ddp import FlatGradAllReducer
from waveformml_amd.psd.graph import GraphedTrainStep
from waveformml_amd.psd.lit import LitPSD
from waveformml_amd.spconv import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
T = int(sys.argv[3]) if len(sys.argv) > 3 else 150
n_dil = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dropout = float(sys.argv[5]) if len(sys.argv) > 5 else 0.2
dtype = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[6] if len(sys.argv) > 6 else "f32"]
cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "gep_config.json")))
cfg["system_config"]["n_samples"] = T
cfg["net_config"]["hparams"]["n_dil"] = n_dil
cfg["net_config"]["hparams"]["wf_params"]["dropout"] = dropout
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream())
ops.ASSUME_VALID_UNIQUE_INDICES = True
torch.manual_seed(0)
gpu = LitPSD(load_config(copy.deepcopy(cfg))).to(dev)
cpu_cfg = copy.deepcopy(cfg)
cpu_cfg["net_config"]["imports"] = ["oracle.spconv" if m == "waveformml_amd.spconv" else m for m in cpu_cfg["net_config"]["imports"]]
cpu = LitPSD(load_config(cpu_cfg))
cpu.load_state_dict({k: v.cpu() for k, v in gpu.state_dict().items()})
c, f, y = synthetic.generate(B, T, 3, seed=1, layout="2d")
batch = ([torch.from_numpy(c).to(dev), torch.from_numpy(f).to(dev).to(dtype)], torch.from_numpy(y).to(dev))
cbatch = ([torch.from_numpy(c), torch.from_numpy(f)], torch.from_numpy(y))
lg, lc = gpu.training_step(batch, 0), cpu.training_step(cbatch, 0)
rel = abs(lg.item() - lc.item()) / abs(lc.item())

reducer = FlatGradAllReducer(gpu.model.parameters())
gpu.optimizer_parameters = reducer.optimizer_parameters()
opt = gpu.configure_optimizers()
opt = opt[0][0] if isinstance(opt, tuple) else opt


def eager():
    reducer.reset()
    loss = gpu.training_step(batch, 0)
    loss.backward()
    reducer.finish()
    opt.step()


for _ in range(3):
    eager()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(steps):
    eager()
torch.cuda.synchronize()
eager_ms = (time.perf_counter() - t) / steps * 1e3
graph_ms = None
try:
    g = GraphedTrainStep(gpu, opt, reducer, batch)
    for _ in range(3):
        g(batch)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        g(batch)
    torch.cuda.synchronize()
    graph_ms = (time.perf_counter() - t) / steps * 1e3
    g.check()
except Exception as e:                      # noqa: BLE001
    print("graph capture not available for this net: %r" % (e,), file=sys.stderr)

torch.set_num_threads(host_cores())            # the cgroup's CPU share, not the host's core count
copt = cpu.configure_optimizers()
copt = copt[0][0] if isinstance(copt, tuple) else copt
n_cpu = max(3, steps // 10)
t = time.perf_counter()
for _ in range(n_cpu):
    copt.zero_grad()
    cpu.training_step(cbatch, 0).backward()
    copt.step()
cpu_ms = (time.perf_counter() - t) / n_cpu * 1e3
print(json.dumps({"config": "GEP.json hparams (2-D, T=%d, n_dil=%d, wf dropout %g, %s rows)" % (T, n_dil, dropout, str(dtype).split(".")[-1]), "batch": B, "rows": int(c.shape[0]), "rel_loss_diff_first_step": rel,
                  "gpu_eager_ms_per_step": round(eager_ms, 3), "gpu_graph_ms_per_step": None if graph_ms is None else round(graph_ms, 3),
                  "gpu_events_per_s": round(B / ((graph_ms or eager_ms) * 1e-3)), "cpu_ms_per_step": round(cpu_ms, 2),
                  "cpu_events_per_s": round(B / (cpu_ms * 1e-3)), "cpu_threads": torch.get_num_threads()}))

"""Experiment: take the strided stages' rulebook builds OUT of the captured step and run them as their own captured
graph S on a second (lower-priority) stream one step AHEAD, beside the previous step's main graph M (rulebooks depend
only on the batch's indices, which the prefetcher has a step early).  Timing only: the same batch is replayed every
step, so S rewrites the tables M reads with the values they already hold (no double buffering here).

  A  the product's captured step (builds on a branch inside the one graph), replay only
  B  M (no strided builds inside) + S on a second stream, pipelined one step ahead, S at lower priority
  C  the same with equal priorities
  D  S then M on ONE stream (no overlap at all)

usage: python tools/exp/pipeline_rulebooks.py [dtype]"""
import copy, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from waveformml_amd import _lib
from waveformml_amd.psd import synthetic
from waveformml_amd.psd.config import DictionaryUtility
from waveformml_amd.psd.ddp import FlatGradAllReducer
from waveformml_amd.psd.graph import GraphedTrainStep
from waveformml_amd.psd.lit import LitPSD
from waveformml_amd.spconv import modules, ops

DT = sys.argv[1] if len(sys.argv) > 1 else "bf16"
fdtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[DT]
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
lo_hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("stream priority range:", lo_hi, flush=True)
_lib.load()
ops.ASSUME_VALID_UNIQUE_INDICES = True
ops.PREFETCH_RULEBOOKS = True
cfg = bench.load_cfg(os.path.join(ROOT, "config", "psd_c2_3d.json"), 256)
hb = [synthetic.generate(256, 256, cfg["system_config"]["n_type"], seed=1234 + 7919 * i) for i in range(8)]
batches = [([torch.from_numpy(c).to(dev), torch.from_numpy(f).to(dev).to(fdtype)], torch.from_numpy(y).to(dev))
           for (c, f, y) in hb]
example = max(batches, key=lambda b: b[0][0].shape[0])


def make(stream):
    torch.cuda.set_stream(stream)
    torch.manual_seed(1234)
    module = LitPSD(DictionaryUtility.to_object(copy.deepcopy(cfg))).to(dev)
    module.train()
    reducer = FlatGradAllReducer(module.model.parameters())
    module.optimizer_parameters = reducer.optimizer_parameters()
    opt = module.configure_optimizers()
    optimizer = opt[0][0] if isinstance(opt, tuple) else opt
    return GraphedTrainStep(module, optimizer, reducer, example)


def timed(fn, n=300, reps=5):
    fn(20)
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn(n)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / n * 1e3)
    return float(np.median(out)), min(out)


HIGH = torch.cuda.Stream(dev, priority=-1)
# ---- A: the product
gA = make(HIGH)
a = timed(lambda n: [gA.graph.replay() for _ in range(n)])
print("A  one graph, builds on a branch inside      %.4f ms/step (min %.4f)" % a, flush=True)

# ---- M + S
STATE = {"plan": None, "S": None}
orig = modules.SparseSequential._prefetch_rulebooks
LOW = None


def patched(mods, x, owner=None):
    if STATE["plan"] is None:
        already = {id(v.rulebook) for v in x.indice_dict.values() if hasattr(v, "rulebook")}
        torch.cuda.synchronize()
        S = torch.cuda.CUDAGraph()
        side = ops.side_stream
        ops.side_stream = lambda d: torch.cuda.current_stream()
        try:
            orig(mods, x)                      # once eagerly: the layers' sticky flags and state buffers must exist
            x.prefetched = None
            torch.cuda.synchronize()
            with torch.cuda.graph(S, stream=LOW):
                orig(mods, x)
        finally:
            ops.side_stream = side
        plan = {k: rb for k, rb in x.prefetched.items() if id(rb) not in already}
        for rb in plan.values():
            rb.ready = None
        S.replay()
        torch.cuda.synchronize()
        STATE.update(plan=plan, S=S)
        print("S holds %d rulebooks" % len({id(rb) for rb in plan.values()}), flush=True)
    x.prefetched = STATE["plan"]


def pipelined(M, S, low):
    evS, evM = torch.cuda.Event(), torch.cuda.Event()
    with torch.cuda.stream(low):
        S.replay()
        evS.record(low)

    def run(n):
        for _ in range(n):
            HIGH.wait_event(evS)            # this step's rulebooks (launched a step ago)
            evM.record(HIGH)
            low.wait_event(evM)             # the next step's builds start with this step
            with torch.cuda.stream(low):
                S.replay()
                evS.record(low)
            M.replay()
    return run


for tag, prio in (("B", 0), ("C", -1)):
    LOW = torch.cuda.Stream(dev, priority=prio)
    STATE.update(plan=None, S=None)
    modules.SparseSequential._prefetch_rulebooks = staticmethod(patched)
    try:
        g = make(HIGH)
    finally:
        modules.SparseSequential._prefetch_rulebooks = staticmethod(orig)
    M, S = g.graph, STATE["S"]
    m = timed(lambda n: [M.replay() for _ in range(n)])
    print("%s  M alone (no strided builds at all)         %.4f ms/step (min %.4f)" % ((tag,) + m), flush=True)
    torch.cuda.set_stream(LOW)
    s = timed(lambda n: [S.replay() for _ in range(n)])
    torch.cuda.set_stream(HIGH)
    print("%s  S alone                                     %.4f ms/step (min %.4f)" % ((tag,) + s), flush=True)
    b = timed(pipelined(M, S, LOW))
    print("%s  M + S pipelined, S priority %2d             %.4f ms/step (min %.4f)" % ((tag, prio) + b), flush=True)
    if tag == "B":
        d = timed(lambda n: [(S.replay(), M.replay()) for _ in range(n)])
        print("D  S then M on one stream                      %.4f ms/step (min %.4f)" % d, flush=True)
    loss = float(g.loss.float().item())
    print("   (loss after the runs: %.4f)" % loss, flush=True)

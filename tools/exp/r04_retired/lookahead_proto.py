"""Experiment: the NEXT batch's rulebooks (SubM + strided builds) built beside the current batch's step as a ROOT branch of
the same captured graph -- no fork edge behind the first conv, no join in front of the strided convs: the main chain reads
rulebooks that the previous replay left behind.  Timing only (the same batch sits in both buffer sets).

  A  the product's captured step (builds of THIS batch on a branch inside the graph), replay only
  L  main chain on prebuilt rulebooks + root branch rebuilding the other buffer set's rulebooks
  M  main chain on prebuilt rulebooks, nothing else (the floor)

usage: python tools/exp/lookahead_proto.py [dtype]"""
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
import waveformml_amd.spconv as spconv

DT = sys.argv[1] if len(sys.argv) > 1 else "bf16"
fdtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[DT]
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
_lib.load()
ops.ASSUME_VALID_UNIQUE_INDICES = True
ops.PREFETCH_RULEBOOKS = True
cfg = bench.load_cfg(os.path.join(ROOT, "config", "psd_c2_3d.json"), 256)
hb = [synthetic.generate(256, 256, cfg["system_config"]["n_type"], seed=1234 + 7919 * i) for i in range(8)]
batches = [([torch.from_numpy(c).to(dev), torch.from_numpy(f).to(dev).to(fdtype)], torch.from_numpy(y).to(dev))
           for (c, f, y) in hb]
example = max(batches, key=lambda b: b[0][0].shape[0])
main = torch.cuda.Stream(dev)
torch.cuda.set_stream(main)
torch.manual_seed(1234)
module = LitPSD(DictionaryUtility.to_object(copy.deepcopy(cfg))).to(dev)
module.train()
reducer = FlatGradAllReducer(module.model.parameters())
module.optimizer_parameters = reducer.optimizer_parameters()
opt = module.configure_optimizers()
optimizer = opt[0][0] if isinstance(opt, tuple) else opt
g = GraphedTrainStep(module, optimizer, reducer, example)
net = module.model


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


a = timed(lambda n: [g.graph.replay() for _ in range(n)])
print("A  product graph, replay only                  %.4f ms/step (min %.4f)" % a, flush=True)


def build_plan(indices, events, n_valid):
    """every layer's rulebook on the current stream, no joins"""
    st = spconv.SparseConvTensor(g.feats, indices, net.spatial_size, net.batch_size_hint or g.n_events)
    st.n_valid, st.events = n_valid, events
    side = ops.side_stream
    ops.side_stream = lambda d, i=0: torch.cuda.current_stream()
    try:
        modules.SparseSequential._prefetch_rulebooks(list(net.sparseModel._modules.values()), st)
    finally:
        ops.side_stream = side
    plan = st.prefetched
    for rb in plan.values():
        rb.ready = None
    return plan


# buffer set B: the same batch once more
indices_b, events_b, n_valid_b = g.indices.clone(), g.events.clone(), g.n_valid.clone()
torch.cuda.synchronize()
plan_a = build_plan(g.indices, g.events, g.n_valid)
plan_b = build_plan(indices_b, events_b, n_valid_b)
torch.cuda.synchronize()
uniq_b = []
for rb in plan_b.values():
    if all(rb is not u for u in uniq_b):
        uniq_b.append(rb)
print("plan: %d layers, %d rulebooks" % (len(plan_b), len(uniq_b)), flush=True)
net.rulebook_plans = {id(g.coords): plan_a}
for _ in range(2):
    g._body()
torch.cuda.synchronize()
M = torch.cuda.CUDAGraph()
with torch.cuda.graph(M, stream=main):
    g._body()
m = timed(lambda n: [M.replay() for _ in range(n)])
print("M  main chain on prebuilt rulebooks            %.4f ms/step (min %.4f)" % m, flush=True)
side = torch.cuda.Stream(dev)
L = torch.cuda.CUDAGraph()
with torch.cuda.graph(L, stream=main):
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for rb in uniq_b:
            rb.rebuild()
    loss = g._body()
    main.wait_stream(side)
l = timed(lambda n: [L.replay() for _ in range(n)])
print("L  main chain + root branch (next rulebooks)   %.4f ms/step (min %.4f)" % l, flush=True)
# with the hand-over launch in front, as a real step has it
l2 = timed(lambda n: [(g._load(example), L.replay()) for _ in range(n)])
print("L  ... with the hand-over launch per step      %.4f ms/step (min %.4f)" % l2, flush=True)
a2 = timed(lambda n: [g(example) for _ in range(n)])
print("A  ... product step with its hand-over launch  %.4f ms/step (min %.4f)" % a2, flush=True)
print("loss %.4f" % float(loss.float().item()))

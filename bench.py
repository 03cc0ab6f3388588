#!/usr/bin/env python
"""bench.py -- waveforms/sec of the LitPSD sparse-conv training step on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md 8d "C2"): SubMConv3d PSD net on the 14x11 PMT grid x
256-sample waveforms, Cin 2 -> 32 (+ two 32->32 SubM layers sharing the rulebook, two strided
SparseConv3d k3 s(1,1,4), BatchNorm+ReLU after each, ToDense, Linear -> 3 classes), 256 synthetic
events per rank per step, bf16 feature rows with fp32 accumulation and fp32 master weights.
One step = rulebook builds + forward + backward + gradient exchange + SGD step on a batch that is
already resident in HBM.  value = events (waveform readouts) per second over all ranks.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

At N = 1 the same JSON line also carries the exact-fp32 path of the same step ("f32_path": throughput and
the logits/loss diff against the CPU oracle, the 1e-5 parity bar) and the CPU baseline.
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md "Chip-level parameters")
F32_MFMA_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32, dense (same guide, "Peak FP32 (matrix)")
_T0 = time.perf_counter()


def log(msg):
    """Progress on stderr (the JSON line on stdout stays alone)."""
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %8.2fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def host_cores():
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="events per rank per step")
    ap.add_argument("--samples", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["f32", "bf16", "f16"], help="storage type of the feature rows")
    ap.add_argument("--config", default=os.path.join(ROOT, "config", "psd_c2_3d.json"))
    ap.add_argument("--cpu-steps", type=int, default=12, help="timed CPU-baseline steps (0 = skip the CPU leg)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true",
                    help="run the step eagerly instead of replaying it as one captured HIP graph")
    ap.add_argument("--batches", type=int, default=8,
                    help="distinct resident batches (different geometry each) the timed loop cycles through")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed --steps region is run this many times; the line reports the median region")
    return ap.parse_args()


def load_cfg(path, samples):
    with open(path) as f:
        cfg = json.load(f)
    cfg["system_config"]["n_samples"] = samples
    return cfg


def workload_name(config_path):
    """The default config is BASELINE.json configs[1]; any other config is named by its file."""
    base = os.path.basename(config_path)
    return "SubMConv3d PSD net" if base == "psd_c2_3d.json" else "SubMConv3d PSD net (%s)" % base


class Env(object):
    pass


def measure(env, args, dtype, steps, warmup, roofline):
    """Builds the module + batch for `dtype`, times `steps` steps, returns (json fields, extras)."""
    from waveformml_amd import _lib
    from waveformml_amd.psd import synthetic
    from waveformml_amd.psd.config import DictionaryUtility
    from waveformml_amd.psd.ddp import FlatGradAllReducer, broadcast_parameters
    from waveformml_amd.psd.lit import LitPSD
    from waveformml_amd.spconv import functional as Fsp
    dev, world, rank = env.dev, env.world, env.rank
    torch.manual_seed(1234)
    module = LitPSD(DictionaryUtility.to_object(copy.deepcopy(env.cfg))).to(dev)
    module.train()
    broadcast_parameters(module)
    init_state = {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}
    # WFS_BENCH_ONE_RANK_RCCL=1 (a rehearsal, not a bench line): one rank with a real RCCL communicator and the whole
    # bucket / hook / in-graph all-reduce machinery of an N-rank step -- what that structure costs before any link does
    force = env.world == 1 and os.environ.get("WFS_BENCH_ONE_RANK_RCCL") == "1"
    reducer = FlatGradAllReducer(module.model.parameters(), exchange=True if force else None)   # flat parameter + gradient buffers
    module.optimizer_parameters = reducer.optimizer_parameters()
    opt = module.configure_optimizers()
    optimizer = opt[0][0] if isinstance(opt, tuple) else opt

    # synthetic batches, resident in HBM before the timed region (weak scaling: fixed events per rank).  The timed loop
    # cycles through `--batches` DIFFERENT batches (own seed each: other hit patterns, other voxel counts), so no step
    # finds the previous step's gather tables or rows in L2 / Infinity Cache by construction of the benchmark.
    fdtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[dtype]
    nb = max(1, args.batches)
    host_batches = [synthetic.generate(args.batch, args.samples, env.cfg["system_config"]["n_type"], seed=1234 + 7919 * i,
                                       rank=rank) for i in range(nb)]
    batches = [([torch.from_numpy(c_).to(dev), torch.from_numpy(f_).to(dev).to(fdtype)], torch.from_numpy(y_).to(dev))
               for (c_, f_, y_) in host_batches]
    c, f, y = host_batches[0]
    batch = batches[0]
    (coords, feats), labels = batch
    # the captured step is sized on the LARGEST of the resident batches (psd/graph.py adds its headroom on top)
    example = max(batches, key=lambda b: b[0][0].shape[0])

    def eager_step(b=None):
        reducer.reset()
        loss = module.training_step(batch if b is None else b, 0)
        loss.backward()
        reducer.finish()
        optimizer.step()
        return loss

    step = eager_step

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # parity snapshot on the un-trained weights.  Train mode, so BatchNorm normalises with batch statistics over
    # the active voxels (in eval mode the fresh running stats leave the activations at ~1e-6 and the logits are just
    # the head's bias).
    with torch.no_grad():
        logits0 = module.model([coords.clone(), feats.clone()]).float()
        loss0 = float(module.criterion(logits0, labels).item())
        logits0 = logits0.cpu()
    log("[%s] model + batch ready: %d voxels" % (dtype, coords.shape[0]))

    mode, gstep = "eager", None
    if not args.no_graph:
        # the whole step (rulebook builds, forward, backward, gradient packing, optimizer) captured once as a HIP
        # graph over capacity-padded buffers with device-side row counts, replayed per step (psd/graph.py)
        from waveformml_amd.psd.graph import GraphedTrainStep
        try:
            gstep = GraphedTrainStep(module, optimizer, reducer, example)
            step = gstep
            mode = "hipgraph"
        except Exception as e:              # noqa: BLE001 -- report and fall back, the number is then an eager one
            log("graph capture failed (%s: %s); running eagerly" % (type(e).__name__, e))
    it = 0
    for _ in range(warmup):
        step(batches[it % nb])
        it += 1
    regions = []
    for _rep in range(max(1, args.repeats)):
        # one timed region = EXACTLY `steps` steps between two barrier + synchronize fences; max over ranks
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step(batches[it % nb])
            it += 1
        fence()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        regions.append(float(t.item()))
    if gstep is not None:
        gstep.check()                       # raises if a captured capacity was exceeded
    elapsed = float(np.median(regions))
    per_step = [r / steps * 1e3 for r in regions]
    log("[%s] timed regions (%s): median %.3f ms/step (min %.3f, max %.3f over %d regions of %d steps, %d batches)"
        % (dtype, mode, elapsed / steps * 1e3, min(per_step), max(per_step), len(regions), steps, nb))
    voxels = [int(b[0][0].shape[0]) for b in batches]
    out = {"value": args.batch * world * steps / elapsed, "ms_per_step": elapsed / steps * 1e3, "execution": mode,
           "final_loss": float(loss.item()), "active_voxels_per_rank": int(np.mean(voxels)),
           "timing": {"regions": len(regions), "steps_per_region": steps, "statistic": "median region",
                      "ms_per_step_min": min(per_step), "ms_per_step_max": max(per_step),
                      "ms_per_step_all": per_step, "distinct_batches": nb,
                      "voxels_min": min(voxels), "voxels_max": max(voxels)}}

    if roofline:
        # per-kernel HIP-event timing.  Event pairs cannot be re-recorded inside a graph replay, so the kernels are
        # timed in an eager pass of the same step right after the timed region (same kernels, same shapes) on every
        # rank (the step contains collectives); profiles/ holds the rocprofv3 kernel trace of the graph replay itself.
        nprof = max(5, min(steps, 20))
        eager_step()
        torch.cuda.synchronize()
        _lib.timing_enable(True)
        for _ in range(nprof):
            eager_step()
        torch.cuda.synchronize()
        timers = {name: _lib.timing_read(tid) for name, tid in (("gather_conv", _lib.TIMER_GATHER_CONV),
                                                                ("gather_dw", _lib.TIMER_GATHER_DW),
                                                                ("rulebook", _lib.TIMER_RULEBOOK),
                                                                ("conv_backward", _lib.TIMER_CONV_BACKWARD))}
        _lib.timing_enable(False)
        Fsp.ACCOUNT = []
        eager_step()
        torch.cuda.synchronize()
        acct, Fsp.ACCOUNT = Fsp.ACCOUNT, None
        per_kind = {}
        for a in acct:
            d = per_kind.setdefault(a["kind"], {"bytes": 0, "flops": 0, "launches": 0})
            d["bytes"] += a["bytes"]
            d["flops"] += a["flops"]
            d["launches"] += 1
        # the dominant conv kernel class by time: forward / dX launches, dW launches, or -- 32 -> 32 layers with 16-bit
        # rows, round 4 -- the one-launch backward (dW + dX, priced with SURVEY 8d's backward bytes)
        dom = max(("gather_conv", "gather_dw", "conv_backward"), key=lambda k: timers[k][0])
        ms, n = timers[dom]
        by = per_kind.get(dom, {"bytes": 0, "flops": 0, "launches": 1})
        avg_ms = ms / max(n, 1)
        bytes_per_launch = by["bytes"] / max(by["launches"], 1)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = pmc_traffic(dom, dtype)
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic[0], "traffic_source": traffic[1],
                           "avg_launch_us": avg_ms * 1e3,
                           "launches_per_step": by["launches"], "algorithmic_bytes_per_launch": bytes_per_launch,
                           "tflops": by["flops"] / max(by["launches"], 1) / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0,
                           "timed_in": "eager pass after the timed region (HIP events on the launch stream)",
                           "per_step_ms": {k: timers[k][0] / nprof for k in timers}}
        # every conv kernel class of the step, same pricing (the headline object above is the dominant one)
        classes = {}
        for kind in ("gather_conv", "gather_dw", "conv_backward"):
            k_ms, k_n = timers[kind]
            kb = per_kind.get(kind)
            if not k_n or not kb:
                continue
            k_us = k_ms / k_n * 1e3
            k_bytes = kb["bytes"] / max(kb["launches"], 1)
            classes[kind] = {"launches_per_step": kb["launches"], "avg_launch_us": k_us,
                             "algorithmic_bytes_per_launch": k_bytes, "achieved": k_bytes / k_us / 1e3,
                             "frac": k_bytes / k_us / 1e3 / HBM_PEAK_GBS}
            kr_us, _src = rocprof_launch_us(kind, dtype)
            if kr_us:
                classes[kind]["rocprof_avg_launch_us"] = kr_us
                classes[kind]["rocprof_frac"] = k_bytes / kr_us / 1e3 / HBM_PEAK_GBS
        out["roofline"]["classes"] = classes
        rp_us, rp_src = rocprof_launch_us(dom, dtype)
        if rp_us:
            out["roofline"]["rocprof"] = {"avg_launch_us": rp_us, "source": rp_src,
                                          "achieved": bytes_per_launch / (rp_us * 1e-6) / 1e9,
                                          "frac": bytes_per_launch / (rp_us * 1e-6) / 1e9 / HBM_PEAK_GBS}
        if dtype == "f32":
            # the exact-fp32 path contracts on v_mfma_f32_32x32x2_f32, 1/16 of the bf16 matrix rate: next to the HBM
            # view, its launches priced against the fp32 MATRIX peak (MI355X_MICROARCH.md "Peak FP32 (matrix)")
            tf = out["roofline"]["tflops"]
            out["roofline"]["mfma"] = {"bound": "mfma", "achieved": tf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                       "frac": tf / F32_MFMA_PEAK_TFLOPS,
                                       "flops_per_launch": by["flops"] / max(by["launches"], 1)}
            if rp_us:       # the same flops over the rocprof launch time of the graph replay
                rtf = by["flops"] / max(by["launches"], 1) / (rp_us * 1e-6) / 1e12
                out["roofline"]["mfma"]["rocprof"] = {"achieved": rtf, "frac": rtf / F32_MFMA_PEAK_TFLOPS,
                                                      "avg_launch_us": rp_us}
    extras = {"init_state": init_state, "batch_np": (c, f, y), "logits0": logits0, "loss0": loss0}
    # orderly end of this measurement: hooks off, device idle, the captured graph -- RCCL nodes included --
    # destroyed NOW, not by the garbage collector inside the next capture or after the process group is gone
    if gstep is not None:
        gstep.close()
    reducer.remove()
    torch.cuda.synchronize()
    return out, extras


def pmc_traffic(kind, dtype):
    """HBM bytes per launch of a kernel class from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    in separate runs; FETCH_SIZE doubled as MI355X_MICROARCH.md "HBM" prescribes for 16-B-per-lane reads on gfx950),
    launch-weighted over the kernels of that class.  None when no PMC summary for this dtype is committed."""
    path = None
    for rnd in ("r04", "r03", "r02", "r01"):        # the newest committed summary
        cand = os.path.join(ROOT, "profiles", "%s_pmc_hbm_traffic_%s.json" % (rnd, dtype))
        if os.path.exists(cand):
            path = cand
            break
    if path is None:
        return None, None
    with open(path) as f:
        pmc = json.load(f)
    prefixes = {"gather_conv": ("k_gconv", "k_gather_conv"), "gather_dw": ("k_gdw", "k_gather_dw", "k_slab_reduce"),
                "conv_backward": ("k_bwd32",)}[kind]
    tot, launches = 0.0, 0
    for name, rec in pmc.items():
        if name.startswith(prefixes):
            kb = rec["fetch_kb_corrected_x2"] + rec["write_size_kb_per_launch"]
            if name.startswith("k_slab_reduce"):          # the reduce pass belongs to its dW launch
                tot += kb * rec["launches"]
                continue
            tot += kb * rec["launches"]
            launches += rec["launches"]
    if launches == 0:
        return None, None
    return tot / launches * 1024.0, "profiles/%s (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, FETCH x2)" % os.path.basename(path)


_KIND_PREFIXES = {"gather_conv": ("k_gconv", "k_gather_conv"), "gather_dw": ("k_gdw", "k_gather_dw"),
                  "conv_backward": ("k_bwd32",)}


def rocprof_launch_us(kind, dtype):
    """Average launch duration of a kernel class in the committed rocprofv3 --kernel-trace --stats summary of this same
    command's graph replay (profiles/rNN_hipgraph_<dtype>_kernel_stats.csv), launch-weighted: the cross-check of
    ``avg_launch_us`` (HIP events, eager pass).  (None, None) when no summary for this dtype is committed."""
    import csv
    import re
    for rnd in ("r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", "%s_hipgraph_%s_kernel_stats.csv" % (rnd, dtype))
        if not os.path.exists(path):
            continue
        tot, calls = 0.0, 0
        with open(path) as f:
            for r in csv.DictReader(f):
                name = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Name"])
                if name.startswith(_KIND_PREFIXES[kind]):
                    tot += float(r["TotalDurationNs"])
                    calls += int(r["Calls"])
        if calls:
            return tot / calls / 1e3, "profiles/%s (rocprofv3 --kernel-trace --stats of the graph replay)" % os.path.basename(path)
    return None, None


def large_batch_roofline(env, dtype, events=2048, samples=256):
    """The dominant kernel class at 2048 events per rank (673 k voxels): the SubM 32 -> 32 forward launch alone, inside a
    replayed graph, priced with the same algorithmic bytes.  What the gather / table organisation itself
    allows once a launch's fixed latency chain is amortised over 8x the rows."""
    from waveformml_amd.psd import synthetic
    from waveformml_amd.spconv import functional as Fsp
    from waveformml_amd.spconv import ops
    dev = env.dev
    fdtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[dtype]
    c, _f, _y = synthetic.generate(events, samples, 3, seed=1234)
    idx = torch.from_numpy(np.ascontiguousarray(c[:, [3, 0, 1, 2]])).to(dev)
    rb = ops.build_rulebook(idx, events, [14, 11, samples], [3] * 3, [1] * 3, [0] * 3, [1] * 3, True)
    N = rb.N
    X = torch.randn(N, 32, device=dev).to(fdtype)
    W = torch.randn(27, 32, 32, device=dev) * 0.1
    table, kmap = rb.table_by_out()
    pairs = int((rb.nbr_out >= 0).sum().item())
    es = X.element_size()
    nbytes = 2 * N * 32 * es + pairs * 8 + 27 * 32 * 32 * 4

    def fn():
        return Fsp.gather_conv(table, kmap, 27, rb.centre_k, N, X, W, False, None)

    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    reps, iters = 10, 10
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / (iters * reps) * 1e3
    g.reset()
    achieved = nbytes / us / 1e3
    return {"bound": "hbm", "kernel": "gather_conv (SubM 32->32 forward, one launch)", "events_per_rank": events,
            "active_voxels": N, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "avg_launch_us": us, "algorithmic_bytes_per_launch": nbytes,
            "timed_in": "%d launches inside a replayed HIP graph, %d replays (torch events on the launch stream)" % (reps, iters),
            "note": "knock-outs at this size: profiles/r04_gconv32_large_batch_knockouts.txt"}


def cpu_reference(env, extras, n_steps):
    """The CPU restatement of the reference's cpuonly path (oracle/spconv.py: per-offset gather -> torch.mm ->
    scatter-add under the SparseSequential loop, fp32) on the same batch and the same initial weights: its logits
    and loss (parity reference) and, if n_steps > 0, its training-step time on the host cores."""
    from waveformml_amd.psd.config import DictionaryUtility
    from waveformml_amd.psd.lit import LitPSD
    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = copy.deepcopy(env.cfg)
    cfg["net_config"]["imports"] = ["oracle.spconv" if m == "waveformml_amd.spconv" else m
                                    for m in cfg["net_config"]["imports"]]
    ref = LitPSD(DictionaryUtility.to_object(cfg))
    ref.load_state_dict(extras["init_state"])
    c, f, y = extras["batch_np"]
    batch = ([torch.from_numpy(c), torch.from_numpy(f)], torch.from_numpy(y))
    ref.train()
    with torch.no_grad():
        logits = ref.model([batch[0][0].clone(), batch[0][1].clone()])
        loss = float(ref.criterion(logits, batch[1]).item())
    base = None
    if n_steps > 0:
        log("cpu_baseline: %d host threads (os.cpu_count()=%s)" % (cores, os.cpu_count()))
        opt = ref.configure_optimizers()
        optimizer = opt[0][0] if isinstance(opt, tuple) else opt

        def step():
            optimizer.zero_grad()
            ls = ref.training_step(([batch[0][0].clone(), batch[0][1]], batch[1]), 0)
            ls.backward()
            optimizer.step()

        step()                                   # warm-up
        times = []
        budget = time.perf_counter() + 40.0      # bounded sample: stop after ~40 s whatever n_steps says
        for i in range(n_steps):
            t0 = time.perf_counter()
            step()
            times.append(time.perf_counter() - t0)
            if time.perf_counter() > budget:
                break
        med = float(np.median(times))
        log("cpu_baseline: %d steps, median %.3f s" % (len(times), med))
        base = {"value": len(y) / med, "unit": "events/s", "cores": cores, "kind": "port",
                "sample": "%d timed training steps (after 1 warm-up) on the same %d-event batch, median; restatement "
                          "of spconv 1.2.1's Native CPU algorithm, fp32, torch threads = %d = min(CPU affinity mask, "
                          "cgroup cpu.max quota, 32) -- this process's share of the host (os.cpu_count() = %d counts the "
                          "whole node; per-offset [n_k x 32] x [32 x 32] products do not scale past 32 threads)"
                          % (len(times), len(y), cores, os.cpu_count() or 0),
                "ms_per_step": med * 1e3}
    return logits, loss, base


def parity(cpu_logits, cpu_loss, gpu_logits, gpu_loss):
    scale = float(cpu_logits.abs().max())
    d = float((cpu_logits - gpu_logits).abs().max())
    return {"max_abs_logit_diff": d, "logit_scale": scale, "max_rel_logit_diff": d / max(scale, 1e-30),
            "loss_cpu": cpu_loss, "loss_gpu": gpu_loss, "rel_loss_diff": abs(cpu_loss - gpu_loss) / max(abs(cpu_loss), 1e-30)}


def visible_gpus():
    """Number of GPU agents the kernel driver exposes, read from sysfs (KFD topology: a node with SIMDs is a GPU), cut
    to the devices ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES leave visible; None when the
    topology cannot be read (then the children find out).  No HIP / torch.cuda call is made."""
    import glob
    nodes = glob.glob(os.path.join(os.environ.get("WFS_KFD_TOPOLOGY", "/sys/class/kfd/kfd/topology"), "nodes", "*", "properties"))
    if not nodes:
        return None
    n = 0
    for path in nodes:
        try:
            with open(path) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        except (OSError, ValueError):
            return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def launch_ranks(args):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start the N ranks here, one child process per
    GPU, BEFORE this process makes any GPU call (a process that has initialised the GPU must not exec or fork GPU work).
    Children get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT exactly as torch.distributed.run would set
    them; rank 0's stdout (the JSON line) is this process's stdout; any failing child fails the run."""
    import socket
    import subprocess
    n = args.gpus
    if not os.environ.get("WFS_REHEARSAL_ONE_GPU"):
        have = visible_gpus()                     # from sysfs: this parent must not touch the GPU runtime before it forks
        if have is not None and have < n:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (WFS_REHEARSAL_ONE_GPU=1 runs N ranks on one "
                             "card over gloo as a dry run of the N-rank code path)" % (n, have))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        # poll instead of waiting rank by rank: one dead rank must not leave the others hanging in a collective
        live = list(procs)
        while live:
            time.sleep(0.2)
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in live:
                        q.terminate()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    raise SystemExit(rc)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
    if os.environ.get("WFS_WATCHDOG"):         # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["WFS_WATCHDOG"]), exit=True)
    # stdout carries ONE line, the JSON: whatever libraries print there (gloo announces its connections on stdout) goes
    # to stderr from here on, the JSON goes to the descriptor saved now
    json_out = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    env = Env()
    env.world = int(os.environ.get("WORLD_SIZE", "1"))
    env.rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != env.world:
        raise SystemExit("--gpus %d but WORLD_SIZE %d" % (args.gpus, env.world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU path)")
    if os.environ.get("WFS_REHEARSAL_ONE_GPU"):      # N ranks on ONE card over gloo: a dry run of the N-rank code path
        local_rank = 0
    torch.cuda.set_device(local_rank)
    env.dev = torch.device("cuda", local_rank)
    # everything runs on an ordinary stream: on ROCm 7.2 eager work on the legacy default stream between two
    # HIP-graph replays hangs the next replay (psd/graph.py "Stream discipline")
    torch.cuda.set_stream(torch.cuda.Stream(env.dev))
    if env.world > 1 or os.environ.get("WFS_BENCH_ONE_RANK_RCCL") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if os.environ.get("WFS_REHEARSAL_ONE_GPU"):
            dist.init_process_group("gloo", rank=env.rank, world_size=env.world)
        else:
            dist.init_process_group("nccl", rank=env.rank, world_size=env.world, device_id=env.dev)   # RCCL over xGMI

    from waveformml_amd import _lib
    from waveformml_amd.spconv import ops as _ops
    _lib.load()
    env.cfg = load_cfg(args.config, args.samples)
    # synthetic events are distinct in-range sites by construction (psd/synthetic.py): skip the index validation
    # read-backs, exactly as spconv (which never validates) does
    _ops.ASSUME_VALID_UNIQUE_INDICES = True
    _ops.PREFETCH_RULEBOOKS = os.environ.get("WFS_PREFETCH", "1") != "0"    # strided layers' rulebooks on a side stream beside the first layers
    # _ops.OVERLAP_DW stays off: dW and dX each fill the CUs' LDS; side by side they just take twice as long
    if os.environ.get("WFS_OVERLAP_DW") == "1":
        _ops.OVERLAP_DW = True

    single = env.world == 1
    f32 = None
    if single and args.dtype != "f32" and args.cpu_steps > 0:
        # the exact-fp32 path of the same step first: it is what the 1e-5 parity bar applies to
        f32, f32_extras = measure(env, args, "f32", min(args.steps, 30), args.warmup, not args.no_roofline)
    main_out, extras = measure(env, args, args.dtype, args.steps, args.warmup, not args.no_roofline)

    if env.rank == 0:
        result = {
            "metric": "waveforms/sec (LitPSD sparse-conv training step)", "value": main_out["value"], "unit": "events/s",
            "n_gpus": env.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_out["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s, 14x11 PMT grid x %d samples, Cin=2 Cout=32, %d events/rank/step, "
                                   "rulebook rebuilt every step" % (workload_name(args.config), args.samples, args.batch),
                       "active_voxels_per_rank": main_out["active_voxels_per_rank"],
                       "global_batch": args.batch * env.world, "parallelism": "dp%d" % env.world,
                       "final_loss": main_out["final_loss"], "execution": main_out["execution"]},
        }
        result["timing"] = main_out["timing"]
        if "roofline" in main_out:
            result["roofline"] = main_out["roofline"]
            if single:
                result["roofline_large_batch"] = large_batch_roofline(env, args.dtype)
        if single and args.cpu_steps > 0:
            ref_extras = f32_extras if f32 is not None else extras
            cpu_logits, cpu_loss, base = cpu_reference(env, ref_extras, args.cpu_steps)
            result["cpu_baseline"] = base
            if f32 is not None:
                result["parity"] = parity(cpu_logits, cpu_loss, f32_extras["logits0"], f32_extras["loss0"])
                result["parity"]["path"] = ("f32 storage; 32 -> 32 products as exact three-piece bf16 splits on the bf16 matrix cores, "
                                            "fp32 accumulation (WFS_SPLIT_BF16=0: fp32 matrix instructions): the 1e-5 bar")
                result["parity_%s" % args.dtype] = parity(cpu_logits, cpu_loss, extras["logits0"], extras["loss0"])
                result["f32_path"] = {k: f32[k] for k in ("value", "ms_per_step", "execution", "timing") if k in f32}
                if "roofline" in f32:
                    result["f32_path"]["roofline"] = f32["roofline"]
            else:
                result["parity"] = parity(cpu_logits, cpu_loss, extras["logits0"], extras["loss0"])
        print(json.dumps(result), file=json_out, flush=True)
    if dist.is_available() and dist.is_initialized():
        # every captured step was closed in measure(); drain the device, meet the other ranks, only then take the
        # communicator down
        torch.cuda.synchronize()
        if env.world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

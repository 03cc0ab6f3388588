"""GPU parity of the callers that keep the input's row set, through the HIP operators, against the CPU
restatement of spconv (oracle.spconv) holding the SAME weights on the SAME seeded inputs:

  * SPConvPreserveNet + LitSegClassifier at config/examples/IoniClassifierCNN.json's shape: 130 -> 138 -> 146 -> 154 -> 104
    -> 54 -> 5 channels, every layer a SparseConv2d followed by the SparseInverseConv2d of the same rulebook;
  * SparseConv2DPreserve versions 1 / 2 (SubMConv2d stacks sharing rulebooks by indice key);
  * SparseConv2DBlock versions 1-3 inside SPConvNet;
  * LitEZ / SparseConv2DForEZ versions 0-3, with and without the single-ended-only loss.

fp32 bar: loss 1e-5 relative; gradients 1e-4 of the tensor's max (the chains are 4-12 BatchNorm layers deep; per-kernel
parity at 1e-5 is tests/test_gpu_parity.py)."""
import copy
import os

import numpy as np
import pytest
import torch

from test_segment_callers import IONI, ez_config, segment_rows

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(got, want, rtol, what):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    scale = float(np.abs(want).max()) if want.size else 1.0
    np.testing.assert_allclose(got, want, rtol=rtol, atol=rtol * max(scale, 1e-30), err_msg=what)


def _swap_imports(cfg, module):
    out = copy.deepcopy(cfg)
    out["net_config"]["imports"] = [module if m in ("oracle.spconv", "waveformml_amd.spconv") else m
                                    for m in out["net_config"]["imports"]]
    return out


def _pair(cls, cfg):
    """(module on the GPU over waveformml_amd.spconv, module on the CPU over oracle.spconv) with equal weights."""
    from waveformml_amd.psd.config import load_config
    gpu = cls(load_config(_swap_imports(cfg, "waveformml_amd.spconv")))
    cpu = cls(load_config(_swap_imports(cfg, "oracle.spconv")))
    cpu.load_state_dict(gpu.state_dict())
    return gpu.to(DEV), cpu


def _compare_step(gpu, cpu, c, f, target, loss_rtol=1e-5, grad_rtol=1e-4):
    lc = cpu.training_step(([c, f.clone()], target), 0)
    lg = gpu.training_step(([c.to(DEV), f.clone().to(DEV)], target.to(DEV)), 0)
    lc.backward()
    lg.backward()
    torch.cuda.synchronize()
    assert abs(lg.item() - lc.item()) <= loss_rtol * abs(lc.item()), (lg.item(), lc.item())
    gmax = max(float(p.grad.abs().max()) for p in cpu.model.parameters() if p.grad is not None)
    n = 0
    for (name, a), b in zip(gpu.model.named_parameters(), cpu.model.parameters()):
        if b.grad is None:
            assert a.grad is None or float(a.grad.abs().max()) == 0.0, name
            continue
        if float(b.grad.abs().max()) < 1e-6 * gmax:
            # a conv bias in front of BatchNorm: identically zero up to rounding noise on both sides
            assert float(a.grad.abs().max()) < 1e-5 * gmax, name
            continue
        _close(a.grad.cpu().numpy(), b.grad.numpy(), grad_rtol, name)
        n += 1
    assert n > 0


@pytest.mark.parametrize("se_only", [False, True], ids=["all_segments", "single_ended_only"])
def test_ioni_classifier_preserve_net_matches_the_cpu_path(se_only):
    from waveformml_amd.psd.litseg import LitSegClassifier
    cfg = copy.deepcopy(IONI)
    if se_only:
        cfg["net_config"]["SELoss"] = True
    torch.manual_seed(21)
    gpu, cpu = _pair(LitSegClassifier, cfg)
    rng = np.random.default_rng(8)
    rows, c, f = segment_rows(rng, 48, 6, 130)
    y = torch.from_numpy(rng.integers(0, 5, len(rows)))
    logits_g = gpu.model([c.to(DEV), f.to(DEV)])
    logits_c = cpu.model([c, f])
    assert logits_g.shape == (len(rows), 5)
    _close(logits_g.detach().cpu().numpy(), logits_c.detach().numpy(), 2e-5, "logits")
    _compare_step(gpu, cpu, c, f, y)


@pytest.mark.parametrize("version,params", [
    (0, dict(pointwise_factor=0.3, n_expansion=2, expansion_factor=1.2, pad_factor=1.0, size_factor=3, stride_factor=1.2,
             trainable_weights=True, dropout=0)),
    (1, dict(n_contraction=3, n_expansion=2, expansion_factor=1.2, size_factor=7)),
    (1, dict(n_contraction=2, n_expansion=2, expansion_factor=1.4, size_factor=5, pointwise_factor=0.5, trainable_weights=True)),
    (2, dict(n_contraction=3, n_expansion=1, expansion_factor=2.0, size_factor=3, filter_multiplier=1.4)),
], ids=["v0_pointwise_bias", "v1", "v1_pointwise_bias", "v2_growing_kernel"])
def test_preserve_block_versions_match_the_cpu_path(version, params):
    """SparseConv2DPreserve as a bare block (64 -> 6 channels): same rows out as in, output and gradients against the CPU
    restatement; versions 1 / 2 share one rulebook between all layers of a kernel size (indice_key)."""
    import oracle.spconv as osp
    import waveformml_amd.spconv as sp
    from waveformml_amd.psd.blocks import SparseConv2DPreserve
    torch.manual_seed(5)
    gpu = SparseConv2DPreserve(sp, 64, 6, 5, version=version, **params)
    cpu = SparseConv2DPreserve(osp, 64, 6, 5, version=version, **params)
    cpu.load_state_dict(gpu.state_dict())
    gpu = gpu.to(DEV)
    rng = np.random.default_rng(9)
    rows, c, f = segment_rows(rng, 24, 7, 64)
    idx = c[:, [2, 0, 1]].contiguous()
    fg = f.clone().to(DEV).requires_grad_(True)
    fc = f.clone().requires_grad_(True)
    n0 = sp.ops.BUILD_COUNT
    og = gpu(sp.SparseConvTensor(fg, idx.to(DEV), [14, 11], 24))
    if version > 0:
        kernels = {l[2] for l in gpu.plan if l[2] > 1}
        assert sp.ops.BUILD_COUNT - n0 == len(kernels)               # one rulebook per kernel size, shared by key
    oc = cpu(osp.SparseConvTensor(fc, idx, [14, 11], 24))
    assert torch.equal(og.indices.cpu(), idx) and og.features.shape == (len(rows), 6)
    _close(og.features.detach().cpu().numpy(), oc.features.detach().numpy(), 2e-5, "features")
    w = torch.from_numpy(rng.standard_normal((len(rows), 6)).astype(np.float32))
    (og.features * w.to(DEV)).sum().backward()
    (oc.features * w).sum().backward()
    _close(fg.grad.cpu().numpy(), fc.grad.numpy(), 1e-4, "input gradient")
    gmax = max(float(p.grad.abs().max()) for p in cpu.parameters())
    for (name, a), b in zip(gpu.named_parameters(), cpu.parameters()):
        if name.endswith(".bias") and float(b.grad.abs().max()) < 1e-4 * gmax:
            # a bias whose only reader is a BatchNorm (through per-row linear layers): its gradient is identically zero,
            # what is left is rounding noise on both sides
            assert float(a.grad.abs().max()) < 1e-4 * gmax, name
            continue
        _close(a.grad.cpu().numpy(), b.grad.numpy(), 1e-4, name)


@pytest.mark.parametrize("version", [1, 2, 3])
def test_block_versions_inside_the_psd_net_match_the_cpu_path(version):
    """LitPSD with ``conv_params.version`` 1-3 (late-decaying padding; channel expansion / contraction; shrinking kernel)."""
    import json
    import os
    from waveformml_amd.psd.lit import LitPSD
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "golden", "gep_config.json")) as fh:
        cfg = json.load(fh)
    cfg["system_config"]["n_samples"] = 32
    cfg["net_config"]["imports"] = ["torch.nn", "waveformml_amd.psd.SPConvNet", "waveformml_amd.spconv"]
    cp = cfg["net_config"]["hparams"]["conv_params"]
    cp.update(version=version, stride_factor=1.2, pad_factor=1.0, size_factor=3, pointwise_factor=0)
    if version >= 2:
        cp.update(expansion_factor=1.5, n_expansion=1)
    torch.manual_seed(31)
    gpu, cpu = _pair(LitPSD, cfg)
    rng = np.random.default_rng(10)
    rows, c, f = segment_rows(rng, 16, 8, 64)
    y = torch.from_numpy(rng.integers(0, 3, 16))
    _compare_step(gpu, cpu, c, f, y)


@pytest.mark.parametrize("version,se_only", [(0, False), (1, False), (2, True), (3, False)])
def test_litez_matches_the_cpu_path(version, se_only):
    from waveformml_amd.psd.litz import LitEZ
    extra = dict(n_expand=1, pointwise_factor=1.5) if version == 3 else {}
    cfg = ez_config(["waveformml_amd.spconv"], version, **extra)
    if se_only:
        cfg["net_config"]["SELoss"] = True
    torch.manual_seed(41 + version)
    gpu, cpu = _pair(LitEZ, cfg)
    rng = np.random.default_rng(11)
    rows, c, f = segment_rows(rng, 32, 6, 40)
    t = torch.from_numpy(rng.standard_normal((len(rows), 2)).astype(np.float32))
    _compare_step(gpu, cpu, c, f, t)


def test_segment_classifier_trains_through_the_captured_step():
    """Trainer(capture=True) with one label per ROW (LitSegClassifier): the captured step pads rows and labels to its
    capacity (labels beyond the valid rows = the criterion's ignore_index, an upper bound for the event count) and must
    follow the eager trainer on the same batches: same losses, same parameters, no eager fallback, no overflow."""
    from waveformml_amd.psd.config import load_config
    from waveformml_amd.psd.litseg import LitSegClassifier
    from waveformml_amd.psd.trainer import Trainer
    cfg = _swap_imports(IONI, "waveformml_amd.spconv")
    cfg["optimize_config"].update(lr=0.01, optimizer_params={"momentum": 0.9, "nesterov": True})
    rng = np.random.default_rng(12)
    batches = []
    for s in range(6):
        rows, c, f = segment_rows(rng, 40 + 3 * (s % 3), 5, 130)
        batches.append(([c, f], torch.from_numpy(rng.integers(0, 5, len(rows)))))
    first = max(range(len(batches)), key=lambda i: batches[i][0][0].shape[0])
    batches.insert(0, batches.pop(first))                      # the capture sizes itself on its first batch
    runs = []
    for capture in (False, True):
        torch.manual_seed(7)
        mod = LitSegClassifier(load_config(copy.deepcopy(cfg)))
        tr = Trainer(max_epochs=2, device=DEV, capture=capture, check_every=2)
        hist = tr.fit(mod, [([c.clone(), f.clone()], y.clone()) for (c, f), y in batches])
        torch.cuda.synchronize()
        runs.append((hist, torch.cat([p.detach().reshape(-1).cpu() for p in mod.model.parameters()]), tr.eager_fallbacks))
    (h0, p0, _), (h1, p1, fb) = runs
    assert fb == 0
    for a, b in zip(h0, h1):
        assert abs(a["train_loss"] - b["train_loss"]) <= 1e-4 * abs(a["train_loss"]), (a, b)
    _close(p1.numpy(), p0.numpy(), 1e-4, "parameters after two epochs")


def test_segment_classifier_trains_from_files_through_the_captured_step():
    """The same comparison fed from FILES in the reference's format (verdict r2 item 4): compound `WaveformPairs` tables
    with a per-row `PID` column -> libwfh5 (`label_name: "PID"` through `label_map`, config/examples/
    IoniClassifierCNN.json:67-76) -> the reference's collate -> Trainer, eager and captured: same losses, same
    parameters, no eager fallback."""
    from torch.utils.data import DataLoader
    from waveformml_amd.psd import data as psd_data, h5data
    from waveformml_amd.psd.config import load_config
    from waveformml_amd.psd.litseg import LitSegClassifier
    from waveformml_amd.psd.trainer import Trainer
    cfg = _swap_imports(IONI, "waveformml_amd.spconv")
    cfg["optimize_config"].update(lr=0.01, optimizer_params={"momentum": 0.9, "nesterov": True})
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h5", "r3", "ioni130")
    label_map = {"1": 0, "4": 1, "6": 2, "256": 3, "258": 2, "512": 4}
    ds = h5data.HDF5Dataset([root], "*WaveformPairSim.h5", "WaveformPairs", "coord", "waveform", 39, label_name="PID",
                            label_map=label_map, normalize=True)
    assert len(ds) == 3
    (c0, f0), y0 = ds[0]
    assert f0.shape[1] == 130 and y0.dtype == torch.int64 and len(y0) == len(c0) and int(y0.max()) <= 4
    loader = DataLoader(ds, batch_size=1, shuffle=False, collate_fn=psd_data.collate_fn)
    batches = sorted(list(loader), key=lambda b: -b[0][0].shape[0])          # the capture sizes itself on its first batch
    runs = []
    for capture in (False, True):
        torch.manual_seed(7)
        mod = LitSegClassifier(load_config(copy.deepcopy(cfg)))
        tr = Trainer(max_epochs=2, device=DEV, capture=capture, check_every=2)
        hist = tr.fit(mod, [([c.clone(), f.clone()], y.clone()) for (c, f), y in batches])
        torch.cuda.synchronize()
        runs.append((hist, torch.cat([p.detach().reshape(-1).cpu() for p in mod.model.parameters()]), tr.eager_fallbacks))
    (h0, p0, _), (h1, p1, fb) = runs
    assert fb == 0 and np.isfinite(h0[-1]["train_loss"])
    for a, b in zip(h0, h1):
        assert abs(a["train_loss"] - b["train_loss"]) <= 1e-4 * abs(a["train_loss"]), (a, b)
    _close(p1.numpy(), p0.numpy(), 1e-4, "parameters after two epochs")

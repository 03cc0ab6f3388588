"""CPU tests of the host-side mirror (waveformml_amd/psd) against golden data captured from the
REFERENCE's own Python callers (tests/golden/reference_callers.json, generated in the build container by
tests/golden/make_reference_goldens.py: layer schedules, head widths, output sizes, collate_fn, plugin loader)."""
import json
import os
import types

import numpy as np
import pytest
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_callers.json")) as _f:
    GOLD = json.load(_f)


class _Rec(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self.rec = dict(cls=type(self).__name__, args=[a if not isinstance(a, (list, tuple)) else list(a) for a in args],
                        kwargs=kwargs)


def _recording_spconv():
    sp = types.SimpleNamespace()
    for name in ["SparseConv2d", "ToDense"]:
        setattr(sp, name, type(name, (_Rec,), {}))

    class SparseSequential(nn.Module):
        def __init__(self, *layers):
            super().__init__()
            self.layers = list(layers)
    sp.SparseSequential = SparseSequential
    return sp


@pytest.mark.parametrize("case", GOLD["sparse_conv2d_block_v0"], ids=lambda c: "nin%d_n%d" % (c["inputs"]["nin"], c["inputs"]["n"]))
def test_layer_schedule_matches_reference_generator(case):
    from waveformml_amd.psd.blocks import SparseConv2DBlock
    c = case["inputs"]
    blk = SparseConv2DBlock(_recording_spconv(), c["nin"], c["nout"], c["n"], list(c["size"]), True, **c["params"])
    layers = []
    for m in blk.alg:
        if isinstance(m, _Rec):
            layers.append(m.rec)
        elif isinstance(m, nn.BatchNorm1d):
            layers.append(dict(cls="BatchNorm1d", args=[m.num_features]))
        elif isinstance(m, nn.Dropout):
            layers.append(dict(cls="Dropout", args=[m.p]))
        else:
            layers.append(dict(cls=type(m).__name__, args=[]))
    assert layers == case["layers"]
    assert [int(v) for v in blk.out_size] == case["out_size"]


def test_gep_schedule_is_the_one_survey_quotes():
    """GEP.json: 300 -> 252 (1x1) -> 158 -> 64 (3x3), dense [10, 7, 64], head 4480 -> 116 -> 3."""
    g = GOLD["sparse_conv2d_block_v0"][0]
    convs = [l for l in g["layers"] if l["cls"] == "SparseConv2d"]
    assert [(l["args"][0], l["args"][1], l["args"][2]) for l in convs] == [(300, 252, 1), (252, 158, 3), (158, 64, 3)]
    assert g["out_size"] == [10, 7, 64]
    assert GOLD["linear_block"][0]["widths"] == [[4480, 116], [116, 3]]


@pytest.mark.parametrize("case", GOLD["linear_block"], ids=lambda c: "%d_%d_%d" % (c["nin"], c["nout"], c["n"]))
def test_linear_block_widths(case):
    from waveformml_amd.psd.blocks import LinearBlock
    lb = LinearBlock(case["nin"], case["nout"], case["n"])
    assert [[m.in_features, m.out_features] for m in lb.alg] == case["widths"]


def test_output_size_table():
    from waveformml_amd.psd.blocks import conv_output_size
    for c in GOLD["calc_output_size"]:
        assert conv_output_size(c["size"], c["nout"], c["fs"], c["stride"], c["pad"], c["dil"], c["ndim"]) == c["out"], c


def test_collate_fn_matches_reference_and_shifts_in_place():
    from waveformml_amd.psd.data import collate_fn
    g = GOLD["collate_fn"]
    batch = [[[torch.tensor(i["coords"], dtype=torch.int32), torch.tensor(i["feats"])], torch.tensor(i["labels"])]
             for i in g["inputs"]]
    (c, f), l = collate_fn(batch)
    assert c.tolist() == g["coords"] and l.tolist() == g["labels"]
    np.testing.assert_array_equal(f.numpy(), np.asarray(g["feats"], np.float32))
    assert c.dtype == torch.int32 and int(c[-1, 2]) + 1 == len(g["labels"])
    assert batch[1][0][0][:, 2].min() >= len(g["inputs"][0]["labels"])       # the inputs were shifted IN PLACE


def test_collate_3d_keeps_event_ids_contiguous():
    from waveformml_amd.psd.data import SyntheticPulseDataset, make_loader
    ds = SyntheticPulseDataset(3, 4, 32, layout="3d", seed=1)
    (c, f), y = next(iter(make_loader(ds, 3, pin_memory=False)))
    assert c.shape[1] == 4 and f.shape[1] == 2 and y.shape[0] == 12
    ev = c[:, 3]
    assert int(ev[0]) == 0 and int(ev[-1]) == 11 and bool((ev[1:] >= ev[:-1]).all())


def test_plugin_loader_matches_reference():
    from waveformml_amd.psd.config import DictionaryUtility, ModuleUtility
    g = GOLD["module_utility"]
    mu = ModuleUtility(["torch.nn", "collections"])
    inst = mu.create_class_instances(g["spec"])
    got = [("class:" + x.__name__) if isinstance(x, type) else ("instance:" + type(x).__name__) for x in inst]
    assert got == g["result"] and sorted(mu.modules) == g["keys"]
    with pytest.raises(IOError):
        mu.retrieve_class("nosuch.Thing")
    d = GOLD["dictionary_utility"]
    obj = DictionaryUtility.to_object(d["input"])
    assert DictionaryUtility.to_dict(obj) == d["roundtrip"] and [obj.a, obj.b.c[1].d] == d["attr"]
    # the plugin key of a dotted module name is its LAST component: this is how the MI355X operators bind
    assert "spconv" in ModuleUtility(["waveformml_amd.spconv"]).modules


def test_litpsd_builds_from_reference_style_config_and_exposes_the_lightning_surface():
    from waveformml_amd.psd.config import load_config
    from waveformml_amd.psd.lit import LitPSD
    root = os.path.dirname(HERE)
    m = LitPSD(load_config(os.path.join(root, "config", "psd_c2_3d.json")))
    for attr in ["model", "criterion", "lr", "training_step", "validation_step", "test_step", "configure_optimizers"]:
        assert hasattr(m, attr)
    assert m.model.n_linear == 35840 and m.model.ndim == 3 and m.model.spatial_size == [14, 11, 256]
    assert m.model.permute_tensor.tolist() == [3, 0, 1, 2]
    opt, sched = m.configure_optimizers()
    assert type(opt[0]).__name__ == "SGD" and opt[0].defaults["nesterov"] and type(sched[0]).__name__ == "ExponentialLR"
    names = [n for n, _ in m.model.sparseModel.named_parameters()]
    assert names[0] == "0.weight" and tuple(m.model.sparseModel[0].weight.shape) == (3, 3, 3, 2, 32)


def test_gep_hparams_config_builds_the_2d_net():
    """reference config/examples/GEP.json with only the `imports` changed."""
    from waveformml_amd.psd.config import load_config
    from waveformml_amd.psd.lit import LitPSD
    cfg = json.load(open(os.path.join(HERE, "golden", "gep_config.json")))
    m = LitPSD(load_config(cfg))
    assert [s["nout"] for s in m.model.schedule] == [252, 158, 64] and m.model.n_linear == 4480
    assert [tuple(l.weight.shape) for l in m.model.linear] == [(116, 4480), (3, 116)]


def test_synthetic_generator_is_seeded_and_well_formed():
    from waveformml_amd.psd import synthetic
    c1, f1, y1 = synthetic.generate(16, 64, 3, seed=5)
    c2, f2, y2 = synthetic.generate(16, 64, 3, seed=5)
    assert np.array_equal(c1, c2) and np.array_equal(f1, f2) and np.array_equal(y1, y2)
    assert c1.dtype == np.int32 and f1.dtype == np.float32 and c1.shape[1] == 4 and f1.shape[1] == 2
    assert c1[:, 0].max() < 14 and c1[:, 1].max() < 11 and c1[:, 2].max() < 64 and set(c1[:, 3]) == set(range(16))
    assert len({tuple(r) for r in c1.tolist()}) == len(c1)            # distinct sites
    assert 0 < f1.max() <= 1.0 and (f1.max(1) >= 8 / (2 ** 14 - 1) - 1e-9).all()   # zero-suppression threshold


def test_per_segment_block_schedules_match_the_reference_builders():
    """zblocks.{z,point,ez}_schedule against the layer lists the REFERENCE's SparseConv2DForZ / Pointwise2DForZ /
    SparseConv2DForEZ (v0) constructors produced (tests/golden/z_schedules.json, made by make_z_goldens.py)."""
    from waveformml_amd.psd import zblocks
    gold = json.load(open(os.path.join(HERE, "golden", "z_schedules.json")))

    def as_layers(plan, todense=True):
        out = []
        for cin, cout, k, pad, bn in plan:
            out.append(["SparseConv2d", cin, cout, k, 1, pad])
            if bn:
                out.append(["BatchNorm1d", cout])
            out.append(["ReLU"])
        return out + ([["ToDense"]] if todense else [])

    for case in gold["z"]:
        assert as_layers(zblocks.z_schedule(**case["args"])) == case["layers"], case["args"]
    for case in gold["point"]:
        assert as_layers(zblocks.point_schedule(**case["args"])) == case["layers"], case["args"]
    for case in gold["ez"]:
        assert as_layers(zblocks.ez_schedule(**case["args"])) == case["layers"], case["args"]
    assert len(gold["z"]) == 90 and len(gold["ez"]) == 15


def _z_config(imports):
    return {
        "system_config": {"model_name": "SingleEndedZConv", "n_samples": 20, "gpu_enabled": False, "half_precision": 0},
        "net_config": {"criterion_class": "L1Loss", "criterion_params": [], "imports": ["torch.nn"] + imports,
                       "net_type": "2DConvolution", "algorithm": "conv",
                       "hparams": {"conv": {"kernel_size": 3, "n_layers": 3}, "point": {"pointwise_layers": 2}}},
        "optimize_config": {"imports": ["torch.optim"], "lr": 0.01, "optimizer_class": "optim.SGD",
                            "optimizer_params": {"momentum": 0.9}},
        "dataset_config": {"imports": []},
    }


def test_litz_segment_loss_on_the_cpu_restatement():
    """LitZ (reference src/engineering/LitZ.py, LitBase._calc_segment_loss) over the CPU restatement of spconv: the
    loss equals sum |prediction - target| over the ACTIVE segments / rows, computed independently from the dense map."""
    from waveformml_amd.psd.config import load_config
    from waveformml_amd.psd.litz import LitZ
    torch.manual_seed(0)
    m = LitZ(load_config(_z_config(["oracle.spconv"])))
    assert [p[:3] for p in m.model.model.plan] == [(40, 27, 3), (27, 14, 1), (14, 1, 1)]
    rng = np.random.default_rng(2)
    B = 5
    rows = sorted({(int(rng.integers(0, 14)), int(rng.integers(0, 11)), e) for e in range(B) for _ in range(3)},
                  key=lambda r: r[2])
    c = torch.tensor(rows, dtype=torch.int32)
    f = torch.from_numpy(rng.random((len(rows), 40)).astype(np.float32))
    z = torch.from_numpy(rng.standard_normal(len(rows)).astype(np.float32))
    loss = m.training_step(([c, f], z), 0)
    dense = m.model([c, f]).detach()
    want = sum(abs(float(dense[e, 0, x, y]) - float(z[i])) for i, (x, y, e) in enumerate(rows)) / len(rows)
    assert abs(loss.item() - want) <= 1e-5 * abs(want)
    loss.backward()
    assert all(p.grad is not None for p in m.model.parameters())


def test_checkpoint_round_trip_in_lightning_layout(tmp_path):
    """Trainer.save_checkpoint writes the dictionary Lightning's ModelCheckpoint writes for the reference
    (``state_dict`` / ``epoch`` / ``global_step`` / ``optimizer_states`` / ``lr_schedulers``, main.py:190-204) under the
    ``epoch=..-val_loss=...ckpt`` name; ``load_from_checkpoint(path, config)`` (reference Evaluate.py:72) and
    ``read_checkpoint`` load it -- and a bare ``state_dict`` file -- with ``weights_only=True``."""
    import copy
    from waveformml_amd.psd.config import load_config
    from waveformml_amd.psd.lit import LitPSD
    from waveformml_amd.psd.trainer import Trainer, load_from_checkpoint, read_checkpoint
    with open(os.path.join(HERE, "golden", "gep_config.json")) as f:
        cfg = json.load(f)
    cfg["net_config"]["imports"] = ["oracle.spconv" if m == "waveformml_amd.spconv" else m for m in cfg["net_config"]["imports"]]
    torch.manual_seed(3)
    mod = LitPSD(load_config(copy.deepcopy(cfg)))
    opt = mod.configure_optimizers()
    optimizer, scheduler = (opt[0][0], opt[1][0]) if isinstance(opt, tuple) else (opt, None)
    # one step so that the optimizer has momentum to save
    for p in mod.model.parameters():
        p.grad = torch.full_like(p, 0.01)
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    tr = Trainer(device="cpu", default_root_dir=str(tmp_path))
    tr.global_step = 17
    path = tr.save_checkpoint(mod, optimizer, scheduler, 4, str(tmp_path / "epoch=4-val_loss=0.12.ckpt"))
    ck = read_checkpoint(path)
    assert ck["epoch"] == 4 and ck["global_step"] == 17
    assert set(ck) >= {"state_dict", "epoch", "global_step", "optimizer_states", "lr_schedulers"}
    assert set(ck["state_dict"]) == set(mod.state_dict())
    bufs = [st["momentum_buffer"] for st in ck["optimizer_states"][0]["state"].values()]
    assert len(bufs) == len(list(mod.model.parameters())) and all(torch.is_tensor(b) for b in bufs)
    if scheduler is not None:
        assert ck["lr_schedulers"][0]["last_epoch"] == 1
    twin = load_from_checkpoint(path, load_config(copy.deepcopy(cfg)))
    for k, v in mod.state_dict().items():
        assert torch.equal(twin.state_dict()[k], v), k
    # optimizer / scheduler state loads back into fresh objects
    opt2 = twin.configure_optimizers()
    optimizer2, scheduler2 = (opt2[0][0], opt2[1][0]) if isinstance(opt2, tuple) else (opt2, None)
    optimizer2.load_state_dict(ck["optimizer_states"][0])
    if scheduler2 is not None:
        scheduler2.load_state_dict(ck["lr_schedulers"][0])
        assert scheduler2.get_last_lr() == scheduler.get_last_lr()
    # a bare state_dict (round-1 files) is accepted too
    bare = str(tmp_path / "bare.ckpt")
    torch.save(mod.state_dict(), bare)
    assert set(read_checkpoint(bare)["state_dict"]) == set(mod.state_dict())


def test_checkpoint_optimizer_state_is_per_parameter_both_ways(tmp_path):
    """The trainer's optimizer owns ONE flat parameter; its checkpoints nevertheless carry the optimizer state the way a
    torch optimizer over ``model.parameters()`` writes it (what Lightning stores for the reference), and a state written
    in that layout -- by this trainer or by the reference -- is gathered back into the flat state on resume."""
    from waveformml_amd.psd.ddp import FlatGradAllReducer
    from waveformml_amd.psd.trainer import load_optimizer_state, per_parameter_optimizer_state
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.ReLU(), torch.nn.Linear(4, 3))
    twin = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.ReLU(), torch.nn.Linear(4, 3))
    twin.load_state_dict(net.state_dict())
    red = FlatGradAllReducer(net.parameters(), world_size=1)
    flat_opt = torch.optim.SGD(red.optimizer_parameters(), lr=0.1, momentum=0.9, nesterov=True)
    ref_opt = torch.optim.SGD(twin.parameters(), lr=0.1, momentum=0.9, nesterov=True)
    x = torch.randn(7, 5)
    for _ in range(2):
        red.reset()                                   # the Trainer's sequence (psd/trainer.Trainer.training_step)
        net(x).square().sum().backward()
        red.finish()
        flat_opt.step()
        ref_opt.zero_grad()
        twin(x).square().sum().backward()
        ref_opt.step()
    params = list(net.parameters())
    sd = per_parameter_optimizer_state(flat_opt, params)
    assert sorted(sd["state"]) == list(range(len(params))) and sd["param_groups"][0]["params"] == list(range(len(params)))
    want = ref_opt.state_dict()
    for i, p in enumerate(params):
        assert tuple(sd["state"][i]["momentum_buffer"].shape) == tuple(p.shape)
        assert torch.allclose(sd["state"][i]["momentum_buffer"], want["state"][i]["momentum_buffer"], atol=1e-7)
    # the reference's optimizer takes it as it is
    other = torch.optim.SGD(twin.parameters(), lr=0.1, momentum=0.9, nesterov=True)
    other.load_state_dict(sd)
    # and the flat optimizer takes the reference's layout
    net2 = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.ReLU(), torch.nn.Linear(4, 3))
    red2 = FlatGradAllReducer(net2.parameters(), world_size=1)
    flat2 = torch.optim.SGD(red2.optimizer_parameters(), lr=0.1, momentum=0.9, nesterov=True)
    load_optimizer_state(flat2, want, list(net2.parameters()))
    back = per_parameter_optimizer_state(flat2, list(net2.parameters()))          # whatever order the flat layout has
    for i in range(len(params)):
        assert torch.equal(back["state"][i]["momentum_buffer"], want["state"][i]["momentum_buffer"])
    # a state for another model: refused with both layouts named
    with pytest.raises(RuntimeError, match="covers 2 parameters"):
        bad = {"state": {}, "param_groups": [dict(want["param_groups"][0], params=[0, 1])]}
        load_optimizer_state(flat2, bad, list(net2.parameters()))


def test_agreed_capacity_covers_every_ranks_own_capacity():
    """psd/graph.GraphedTrainStep.capacity_for (the capacity ranks capture with when batch shapes are agreed ahead of
    time): for any rank's own (rows, labels) below the agreed (largest rows, smallest label count) the rank's own
    capacity never exceeds the agreed one -- across the granule steps (64 / 256 / 512) and the label-dependent headroom --
    so max(own, agreed) is the same number on every rank; and fits_counts decides from the agreed counts alone."""
    import types
    from waveformml_amd.psd.graph import GraphedTrainStep as G
    rng = np.random.default_rng(3)
    for _ in range(2000):
        rows_max = int(rng.integers(1, 200000))
        labels_min = int(rng.integers(1, 4096))
        own_rows = int(rng.integers(1, rows_max + 1))
        own_labels = int(rng.integers(labels_min, 2 * labels_min + 1))
        assert G.capacity_for(own_rows, own_labels) <= G.capacity_for(rows_max, labels_min)
    step = types.SimpleNamespace(per_row=False, n_cap=1000, labels=torch.zeros(16))
    assert G.fits_counts(step, 1000, 16, 16, True) and not G.fits_counts(step, 1001, 16, 16, True)
    assert not G.fits_counts(step, 900, 15, 16, True) and not G.fits_counts(step, 900, 16, 17, True)
    seg = types.SimpleNamespace(per_row=True, n_cap=1000, labels=torch.zeros(1000))
    assert G.fits_counts(seg, 1000, 1000, 1000, False) and not G.fits_counts(seg, 900, 900, 900, True)

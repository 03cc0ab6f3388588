"""The block builders of waveformml_amd.psd (SparseConv2DBlock versions 1-3, SparseConv2DPreserve versions 0-2,
SparseConv2DForEZ versions 1-3) against the layer lists the REFERENCE's own constructors produced for the same arguments
(tests/golden/block_schedules.json, generated in the build container by tests/golden/make_block_goldens.py with a
recording stand-in for spconv): same classes, positional and keyword arguments, in the same order, and the same exception
class where the reference's argument checks refuse a combination."""
import json
import os
import types

import pytest
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "block_schedules.json")) as _f:
    GOLD = json.load(_f)


class _Rec(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self.rec = dict(cls=type(self).__name__, args=[a if not isinstance(a, (list, tuple)) else list(a) for a in args],
                        kwargs=kwargs)


def _recording_spconv():
    sp = types.SimpleNamespace()
    for name in ["SparseConv2d", "SubMConv2d", "SparseInverseConv2d", "ToDense"]:
        setattr(sp, name, type(name, (_Rec,), {}))

    class SparseSequential(nn.Module):
        def __init__(self, *layers):
            super().__init__()
            self.layers = list(layers)
    sp.SparseSequential = SparseSequential
    return sp


def _layers_of(mods):
    out = []
    for m in mods:
        if isinstance(m, _Rec):
            out.append(dict(cls=m.rec["cls"], args=m.rec["args"], kwargs=m.rec["kwargs"]))
        elif isinstance(m, nn.BatchNorm1d):
            out.append(dict(cls="BatchNorm1d", args=[m.num_features], kwargs={}))
        elif isinstance(m, nn.Dropout):
            out.append(dict(cls="Dropout", args=[m.p], kwargs={}))
        else:
            out.append(dict(cls=type(m).__name__, args=[], kwargs={}))
    return out


def _build(make):
    try:
        return make(), None
    except Exception as e:
        return None, type(e).__name__


def _id(case):
    a = case["args"]
    return "v%d_%s" % (a["version"], "_".join("%s" % a[k] for k in sorted(a) if k not in ("version", "size", "to_dense")))[:70]


@pytest.mark.parametrize("case", GOLD["block"], ids=_id)
def test_block_versions_1_to_3(case):
    from waveformml_amd.psd.blocks import SparseConv2DBlock
    a = dict(case["args"])
    a["size"] = list(a["size"])
    blk, err = _build(lambda: SparseConv2DBlock(_recording_spconv(), **a))
    assert err == case["error"]
    if err is None:
        assert _layers_of(blk.alg) == case["layers"]
        assert [int(v) for v in blk.out_size] == case["out_size"]


@pytest.mark.parametrize("case", GOLD["preserve"], ids=_id)
def test_preserve_versions_0_to_2(case):
    from waveformml_amd.psd.blocks import SparseConv2DPreserve
    blk, err = _build(lambda: SparseConv2DPreserve(_recording_spconv(), **case["args"]))
    assert err == case["error"]
    if err is None:
        assert _layers_of(blk.alg) == case["layers"]


def test_ioni_classifier_schedule_is_the_one_survey_quotes():
    """config/examples/IoniClassifierCNN.json: 130 -> 138 -> 146 -> 154 -> 104 -> 54 -> 5, each layer a SparseConv2d
    followed by the SparseInverseConv2d of the same indice key."""
    g = GOLD["preserve"][0]
    convs = [l for l in g["layers"] if l["cls"] == "SparseConv2d"]
    assert [l["args"][0] for l in convs] + [convs[-1]["args"][1]] == [130, 138, 146, 154, 104, 54, 5]
    assert [l["args"][2] for l in convs] == [3, 3, 2, 2, 2, 2]            # kernels (SURVEY.md 8c i)
    assert [l["args"][4] for l in convs] == [1, 1, 0, 0, 0, 0]            # paddings
    assert [l["args"][3] for l in convs] == [1] * 6                        # strides
    inv = [l for l in g["layers"] if l["cls"] == "SparseInverseConv2d"]
    assert [l["args"][3] for l in inv] == [c["kwargs"]["indice_key"] for c in convs] == ["ind_%d" % i for i in range(6)]


@pytest.mark.parametrize("case", GOLD["ez"], ids=_id)
def test_ez_versions_1_to_3(case):
    from waveformml_amd.psd.zblocks import SparseConv2DForEZ
    blk, err = _build(lambda: SparseConv2DForEZ(_recording_spconv(), **case["args"]))
    assert err == case["error"]
    if err is None:
        assert _layers_of(blk.network.layers) == case["layers"]

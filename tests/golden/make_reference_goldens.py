"""Generates tests/golden/reference_callers.json by importing the REFERENCE's own Python callers
(/root/reference, this container only) -- the code on either side of the spconv boundary:

  * src/models/SPConvBlocks.py  SparseConv2DBlock (version 0)  -> layer schedules (GEP.json hparams and variants)
  * src/models/ConvBlocks.py    LinearBlock                    -> head widths
  * src/utils/ModelValidation.py calc_output_size              -> output-size table
  * src/engineering/PSDDataModule.py collate_fn                -> 3-item batch in/out
  * src/utils/util.py           ModuleUtility / DictionaryUtility -> plugin-loader behaviour

Third-party modules that are not installed here are replaced by inert stand-ins for the IMPORT only
(git, pytorch_lightning); `spconv` is a recorder whose constructors just remember their arguments, so
nothing of this repository's implementation takes part in producing the goldens.
The committed JSON is data (inputs and expected outputs); no reference source is copied.

Run:  python tests/golden/make_reference_goldens.py
"""
import json
import os
import sys
import types

import torch
from torch import nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


class _Rec(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self.rec = dict(cls=type(self).__name__, args=[a if not isinstance(a, (list, tuple)) else list(a) for a in args],
                        kwargs=kwargs)


def _stub_modules():
    sp = types.ModuleType("spconv")
    for name in ["SparseConv2d", "SparseConv3d", "SubMConv2d", "SubMConv3d", "SparseInverseConv2d", "ToDense",
                 "SparseConvTensor"]:
        setattr(sp, name, type(name, (_Rec,), {}))

    class SparseSequential(nn.Module):
        def __init__(self, *layers):
            super().__init__()
            self.layers = list(layers)
    sp.SparseSequential = SparseSequential
    sys.modules["spconv"] = sp
    sys.modules["git"] = types.ModuleType("git")
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningDataModule = object
    pl.LightningModule = nn.Module
    plugins = types.ModuleType("pytorch_lightning.plugins")
    plugins.DDPPlugin = object
    pl.plugins = plugins
    sys.modules["pytorch_lightning"] = pl
    sys.modules["pytorch_lightning.plugins"] = plugins


def main():
    _stub_modules()
    sys.path.insert(0, REF)
    from src.models.SPConvBlocks import SparseConv2DBlock
    from src.models.ConvBlocks import LinearBlock
    from src.utils.ModelValidation import ModelValidation, DIM, NIN, NOUT, FS, STR, PAD, DIL
    from src.engineering.PSDDataModule import collate_fn
    from src.utils.util import DictionaryUtility, ModuleUtility
    out = {}

    # ---- layer schedules
    schedules = []
    cases = [
        dict(nin=300, nout=20, n=3, size=[14, 11, 300], params=dict(pointwise_factor=0.1735, size_factor=4, pad_factor=0.667,
                                                                stride_factor=1, dil_factor=1, trainable_weights=False)),
        dict(nin=300, nout=20, n=3, size=[14, 11, 300], params=dict(size_factor=3, pad_factor=1.0, stride_factor=2, dil_factor=1)),
        dict(nin=128, nout=16, n=4, size=[14, 11, 128], params=dict(depth_factor=0.5, size_factor=5, pad_factor=0.5, dil_factor=2)),
        dict(nin=64, nout=64, n=2, size=[14, 11, 64], params=dict(dropout=0.1, pad_factor=1.0)),
    ]
    for c in cases:
        blk = SparseConv2DBlock(c["nin"], c["nout"], c["n"], list(c["size"]), True, **c["params"])
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
        schedules.append(dict(inputs=c, layers=layers, out_size=[int(v) for v in blk.out_size]))
    out["sparse_conv2d_block_v0"] = schedules

    # ---- LinearBlock
    out["linear_block"] = []
    for nin, nout, n in [(4480, 3, 2), (35840, 3, 1), (1000, 10, 3), (64, 64, 2)]:
        lb = LinearBlock(nin, nout, n)
        out["linear_block"].append(dict(nin=nin, nout=nout, n=n, widths=[[m.in_features, m.out_features] for m in lb.alg]))

    # ---- calc_output_size
    table = []
    for size, fs, st, pd, dil, nout, ndim in [([14, 11, 300], 3, 1, 0, 1, 252, 2), ([14, 11, 252], 3, 1, 1, 1, 158, 2),
                                              ([14, 11, 64], 4, 2, 1, 1, 32, 2), ([14, 11, 256, 2], 3, 1, 0, 1, 32, 3),
                                              ([12, 9, 64, 32], 3, 2, 1, 1, 16, 3), ([14, 11, 33], 5, 1, 2, 2, 8, 2)]:
        arg = {DIM: ndim, NIN: size[-1], NOUT: nout, FS: [fs] * 4, STR: [st] * 4, PAD: [pd] * 4, DIL: [dil] * 4}
        res = ModelValidation.calc_output_size(arg, list(size), "cur", "prev", ndim)
        table.append(dict(size=size, fs=fs, stride=st, pad=pd, dil=dil, nout=nout, ndim=ndim, out=[int(v) for v in res]))
    out["calc_output_size"] = table

    # ---- collate_fn (column 2 is the event id in the 2-D layout)
    g = torch.Generator().manual_seed(5)
    items = []
    for n_ev, n_rows in [(3, 7), (2, 4), (4, 9)]:
        ev = torch.sort(torch.randint(0, n_ev, (n_rows,), generator=g)).values
        ev[-1] = n_ev - 1
        coords = torch.stack([torch.randint(0, 14, (n_rows,), generator=g), torch.randint(0, 11, (n_rows,), generator=g), ev], 1).int()
        feats = torch.rand(n_rows, 6, generator=g)
        labels = torch.randint(0, 3, (n_ev,), generator=g)
        items.append(([coords, feats], labels))
    inputs = [dict(coords=c.tolist(), feats=f.tolist(), labels=l.tolist()) for (c, f), l in items]
    (c, f), l = collate_fn([[[c.clone(), f.clone()], l.clone()] for (c, f), l in items])
    out["collate_fn"] = dict(inputs=inputs, coords=c.tolist(), feats=f.tolist(), labels=l.tolist())

    # ---- plugin loader
    mu = ModuleUtility(["torch.nn", "collections"])
    inst = mu.create_class_instances(["nn.Linear", [4, 2], "nn.ReLU", "nn.Dropout", [0.5], "nn.Identity"])
    out["module_utility"] = dict(spec=["nn.Linear", [4, 2], "nn.ReLU", "nn.Dropout", [0.5], "nn.Identity"],
                                 result=[("class:" + x.__name__) if isinstance(x, type) else ("instance:" + type(x).__name__)
                                         for x in inst],
                                 keys=sorted(mu.modules))
    obj = DictionaryUtility.to_object({"a": 1, "b": {"c": [1, {"d": 2}], "_hidden": 3}})
    out["dictionary_utility"] = dict(input={"a": 1, "b": {"c": [1, {"d": 2}], "_hidden": 3}},
                                     roundtrip=DictionaryUtility.to_dict(obj), attr=[obj.a, obj.b.c[1].d])
    with open(os.path.join(HERE, "reference_callers.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote reference_callers.json:", {k: (len(v) if isinstance(v, list) else "dict") for k, v in out.items()})


if __name__ == "__main__":
    main()

"""Generates tests/golden/z_schedules.json by importing the REFERENCE's per-segment block builders
(/root/reference/src/models/SPConvBlocks.py: SparseConv2DForZ, Pointwise2DForZ, SparseConv2DForEZ version 0; this
container only) with `spconv` replaced by a recorder whose constructors just remember their arguments -- same method as
make_reference_goldens.py.  The JSON is data: constructor arguments in, the layer list (class, arguments) out.

Run:  python tests/golden/make_z_goldens.py
"""
import json
import os
import sys

from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_reference_goldens as base   # noqa: E402


def layers_of(block):
    out = []
    for m in block.network.layers:
        if hasattr(m, "rec"):
            out.append([m.rec["cls"]] + [int(a) for a in m.rec["args"]])
        elif isinstance(m, nn.BatchNorm1d):
            out.append(["BatchNorm1d", m.num_features])
        else:
            out.append([type(m).__name__])
    return out


def main():
    base._stub_modules()
    sys.path.insert(0, base.REF)
    from src.models.SPConvBlocks import Pointwise2DForZ, SparseConv2DForEZ, SparseConv2DForZ
    out = {"z": [], "point": [], "ez": []}
    for in_planes in (300, 150, 64):
        for k in (3, 5, 7):
            for n_layers in (1, 2, 3, 5):
                for pw in (0, 1, 2):
                    if pw > 0 and n_layers == 1 or pw > n_layers:
                        continue
                    args = dict(in_planes=in_planes, kernel_size=k, n_layers=n_layers, pointwise_layers=pw)
                    out["z"].append({"args": args, "layers": layers_of(SparseConv2DForZ(**args))})
        for n in (2, 3, 4):
            out["point"].append({"args": dict(in_planes=in_planes, pointwise_layers=n),
                                 "layers": layers_of(Pointwise2DForZ(in_planes, n))})
        for kw in (dict(), dict(out_planes=1), dict(n_conv=2, n_point=2, conv_position=2, kernel_size=5),
                   dict(n_conv=0, n_point=4), dict(n_conv=3, n_point=0, conv_position=1, kernel_size=7, batchnorm=False)):
            args = dict(in_planes=in_planes, **kw)
            out["ez"].append({"args": args, "layers": layers_of(SparseConv2DForEZ(**args))})
    with open(os.path.join(HERE, "z_schedules.json"), "w") as f:
        json.dump(out, f)
    print({k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()

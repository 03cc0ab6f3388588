"""Generates tests/golden/block_schedules.json by importing the REFERENCE's block builders (/root/reference/src/models/
SPConvBlocks.py, this container only): SparseConv2DBlock versions 1-3, SparseConv2DPreserve versions 0-2 and
SparseConv2DForEZ versions 1-3, with `spconv` replaced by the recorder of make_reference_goldens.py (its constructors only
remember their arguments).  The JSON is data: constructor arguments in, the layer list (class, positional arguments,
keyword arguments) out; no reference source is copied.

Run:  python tests/golden/make_block_goldens.py
"""
import json
import os
import sys

from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_reference_goldens as base   # noqa: E402


def layers_of(mods):
    out = []
    for m in mods:
        if hasattr(m, "rec"):
            out.append(dict(cls=m.rec["cls"], args=m.rec["args"], kwargs=m.rec["kwargs"]))
        elif isinstance(m, nn.BatchNorm1d):
            out.append(dict(cls="BatchNorm1d", args=[m.num_features], kwargs={}))
        elif isinstance(m, nn.Dropout):
            out.append(dict(cls="Dropout", args=[m.p], kwargs={}))
        else:
            out.append(dict(cls=type(m).__name__, args=[], kwargs={}))
    return out


def attempt(make):
    try:
        return make(), None
    except Exception as e:                        # the builders' own argument checks are part of the behaviour
        return None, type(e).__name__


def main():
    base._stub_modules()
    sys.path.insert(0, base.REF)
    from src.models.SPConvBlocks import SparseConv2DBlock, SparseConv2DForEZ, SparseConv2DPreserve
    out = {"block": [], "preserve": [], "ez": []}

    block_params = [
        dict(size_factor=3, pad_factor=1.0, stride_factor=1, dil_factor=1),
        dict(size_factor=5, pad_factor=0.667, stride_factor=2, dil_factor=1, pointwise_factor=0.2, trainable_weights=True),
        dict(size_factor=4, pad_factor=0.5, stride_factor=1.2, dil_factor=1, expansion_factor=1.3, n_expansion=1),
        dict(size_factor=3, pad_factor=1.0, stride_factor=1.2, dil_factor=1, expansion_factor=1.2, n_expansion=2, dropout=0.1,
             pointwise_factor=0.1),
        dict(size_factor=7, pad_factor=1.0, stride_factor=1, dil_factor=2, depth_factor=0.5),
    ]
    for version in (1, 2, 3):
        for (nin, nout, n, size) in [(300, 20, 3, [14, 11, 300]), (130, 5, 6, [14, 11, 130]), (64, 16, 4, [14, 11, 64]),
                                     (32, 32, 2, [14, 11, 32])]:
            for p in block_params:
                if version >= 2 and "depth_factor" in p:
                    continue
                args = dict(nin=nin, nout=nout, n=n, size=size, to_dense=True, version=version, **p)
                blk, err = attempt(lambda: SparseConv2DBlock(**dict(args, size=list(size))))
                rec = dict(args=args, error=err)
                if blk is not None:
                    rec.update(layers=layers_of(blk.alg), out_size=[int(v) for v in blk.out_size])
                out["block"].append(rec)

    ioni = dict(pointwise_factor=0, pad_factor=1.0, size_factor=3, stride_factor=1.2, n_expansion=3, expansion_factor=1.2,
                trainable_weights=False, dil_factor=1)           # config/examples/IoniClassifierCNN.json conv_params
    preserve_cases = [
        dict(nin=130, nout=5, n=6, **ioni),
        dict(nin=130, nout=5, n=6, **dict(ioni, pointwise_factor=0.3, n_expansion=2, dropout=0.2, trainable_weights=True)),
        dict(nin=300, nout=3, n=4, size_factor=5, pad_factor=0.5, stride_factor=2, dil_factor=1, n_expansion=1,
             expansion_factor=1.5),
        dict(nin=64, nout=8, n=3, size_factor=4, pad_factor=1.0, stride_factor=1, n_expansion=0),
        dict(nin=64, nout=8, n=2, size_factor=3, n_expansion=2, expansion_factor=1.1),          # too many expansions
    ]
    for c in preserve_cases:
        args = dict(version=0, **c)
        blk, err = attempt(lambda: SparseConv2DPreserve(**args))
        out["preserve"].append(dict(args=args, error=err, layers=layers_of(blk.alg) if blk is not None else None))
    for version in (1, 2):
        for c in [dict(nin=130, nout=5, n=0, n_contraction=3, n_expansion=2, expansion_factor=1.2, size_factor=7),
                  dict(nin=130, nout=5, n=0, n_contraction=2, n_expansion=3, expansion_factor=1.4, size_factor=5,
                       pointwise_factor=0.5, trainable_weights=True, dropout=0.1),
                  dict(nin=64, nout=2, n=0, n_contraction=1, n_expansion=0, size_factor=3),
                  dict(nin=64, nout=2, n=0, n_contraction=4, n_expansion=0, size_factor=3, filter_multiplier=1.4),
                  dict(nin=64, nout=2, n=0, n_contraction=3, n_expansion=1, expansion_factor=2.0, size_factor=9,
                       filter_multiplier=0.7),
                  dict(nin=64, nout=2, n=0, n_contraction=2, n_expansion=0, size_factor=4),
                  dict(nin=64, nout=2, n=0, n_contraction=0, n_expansion=0, size_factor=3)]:
            args = dict(version=version, **c)
            blk, err = attempt(lambda: SparseConv2DPreserve(**args))
            out["preserve"].append(dict(args=args, error=err, layers=layers_of(blk.alg) if blk is not None else None))

    for version in (1, 2, 3):
        for in_planes in (300, 130, 64):
            for kw in (dict(), dict(out_planes=1), dict(n_conv=2, n_point=2, conv_position=2, kernel_size=5),
                       dict(n_conv=3, n_point=1, conv_position=1, kernel_size=7, batchnorm=False),
                       dict(n_conv=2, n_point=3, conv_position=2, kernel_size=9, n_expand=2, pointwise_factor=1.5),
                       dict(n_conv=1, n_point=2, conv_position=3, kernel_size=3, n_expand=1, pointwise_factor=2.0),
                       dict(n_conv=0, n_point=4), dict(n_conv=1, n_point=1, conv_position=1, kernel_size=4)):
                args = dict(in_planes=in_planes, version=version, **kw)
                blk, err = attempt(lambda: SparseConv2DForEZ(**args))
                out["ez"].append(dict(args=args, error=err,
                                      layers=layers_of(blk.network.layers) if blk is not None else None))
    with open(os.path.join(HERE, "block_schedules.json"), "w") as f:
        json.dump(out, f)
    print({k: (len(v), sum(1 for r in v if r["error"])) for k, v in out.items()})


if __name__ == "__main__":
    main()

"""Layer-schedule generators for the PSD net: host-side mirror of the reference's builders
(src/models/SPConvBlocks.py:411-517 ``SparseConv2DBlock`` version 0, src/models/ConvBlocks.py:82-102
``LinearBlock``, src/utils/ModelValidation.py:119-177 output-size arithmetic).  Pure Python
arithmetic + constructor calls on whatever module is handed in as ``spconv``; the golden schedules
in tests/golden/schedules.json were captured from the reference's own generators.
"""
from math import floor

from torch import nn


def conv_output_length(length, ksize, stride, padding, dilation):
    """Reference ModelValidation.calc_output_size_1d: TRUE division, the caller truncates with int()."""
    return (length + 2 * padding - ksize - (ksize - 1) * (dilation - 1)) / stride + 1


def conv_output_size(size, n_out, ksize, stride, padding, dilation, ndim):
    """[spatial..., C] -> [int(spatial')..., n_out] for one conv layer (same k/s/p/d on every axis)."""
    return [int(conv_output_length(size[d], ksize, stride, padding, dilation)) for d in range(ndim)] + [int(n_out)]


def channel_schedule(nin, nout, n, pointwise_factor=0, depth_factor=0):
    """Channel counts [nin, c1, ..., cn] of a version-0 block (reference SPConvBlocks.py:460-483)."""
    if nin == nout:
        return [nin] * (n + 1)
    if pointwise_factor > 0:
        frames = [nin, nin - int(floor((nin - nout) * pointwise_factor))]
        step = float(nin - nout) / n
    elif depth_factor > 0:
        frames = [nin, int(nin * depth_factor)]
        step = float(frames[-1] - nout) / (n - 1) if n > 1 else 0.0
    else:
        step = float(nin - nout) / n
        return [int(floor(nin - step * i)) for i in range(n + 1)]
    for _ in range(n - 1):
        val = int(floor(frames[-1] - step))
        frames.append(val if val > nout else nout)
    return frames


def layer_hyperparameters(i, n, size_factor, pad_factor, stride_factor, dil_factor, pointwise_first):
    """(kernel, stride, padding, dilation) of layer i (reference SPConvBlocks.py:484-496)."""
    decay = i / (n + 1)
    fs = max(int(floor(size_factor / (i + 1.))), 3)
    st = max(stride_factor - int(floor((stride_factor - 1) / (i + 1.))), 1)
    dil = int(round(dil_factor ** i))
    pd = int(round(pad_factor * (fs - 1) * dil_factor) * decay)
    if i == 0 and pointwise_first:
        return 1, 1, 0, 1
    return fs, st, pd, dil


class SparseConv2DBlock(object):
    """Version-0 block: n x (SparseConv2d -> BatchNorm1d -> ReLU [-> Dropout]) [-> ToDense]."""

    def __init__(self, spconv, nin, nout, n, size, to_dense, size_factor=3, pad_factor=0.0, stride_factor=1,
                 dil_factor=1, pointwise_factor=0, depth_factor=0, dropout=0, trainable_weights=False,
                 version=0, **unused):
        assert n > 0
        if version != 0:
            raise NotImplementedError("only the version-0 schedule is mirrored here")
        self.ndim = len(size) - 1
        self.out_size = list(size)
        self.alg = []
        self.schedule = []
        frames = channel_schedule(nin, nout, n, pointwise_factor, depth_factor)
        for i in range(n):
            fs, st, pd, dil = layer_hyperparameters(i, n, size_factor, pad_factor, stride_factor, dil_factor,
                                                    pointwise_factor > 0)
            # the reference passes `trainable_weights` positionally into spconv's `bias` slot (:498)
            self.alg.append(spconv.SparseConv2d(frames[i], frames[i + 1], fs, st, pd, dil, 1, trainable_weights))
            self.alg.append(nn.BatchNorm1d(frames[i + 1]))
            self.alg.append(nn.ReLU())
            if dropout:
                self.alg.append(nn.Dropout(dropout))
            if self.out_size[-1] != frames[i]:
                raise IOError("Input feature dimension {0} does not match previous output feature dimension {1}."
                              .format(frames[i], self.out_size[-1]))
            self.out_size = conv_output_size(self.out_size, frames[i + 1], fs, st, pd, dil, self.ndim)
            self.schedule.append(dict(nin=frames[i], nout=frames[i + 1], kernel=fs, stride=st, padding=pd,
                                      dilation=dil, out_size=list(self.out_size)))
        if to_dense:
            self.alg.append(spconv.ToDense())
        self.func = spconv.SparseSequential(*self.alg)


class LinearBlock(object):
    """n geometrically interpolated nn.Linear layers, no activations (reference ConvBlocks.py:82-102)."""

    def __init__(self, nin, nout, n):
        assert n > 0 and nin > 0
        factor = pow(float(nout) / nin, 1. / n)
        self.widths = [int(round(nin * pow(factor, i))) for i in range(n + 1)]
        self.alg = [nn.Linear(self.widths[i], self.widths[i + 1]) for i in range(n)]
        self.func = nn.Sequential(*self.alg)

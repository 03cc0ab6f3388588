"""Layer-schedule generators for the PSD net: host-side mirror of the reference's builders
(src/models/SPConvBlocks.py:411-727 ``SparseConv2DBlock`` versions 0-3, :730-948 ``SparseConv2DPreserve`` versions 0-2,
src/models/ConvBlocks.py:82-102 ``LinearBlock``, src/utils/ModelValidation.py:119-177 output-size arithmetic).  Pure
Python arithmetic + constructor calls on whatever module is handed in as ``spconv``; the golden schedules in
tests/golden/reference_callers.json and tests/golden/block_schedules.json were captured from the reference's own
generators (tests/golden/make_reference_goldens.py, make_block_goldens.py).
"""
from math import ceil, floor

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


def frame_expansion(first, factor, n, use_round=False):
    """n channel counts growing linearly from ``first`` towards round(factor * first) (reference :389-397)."""
    step = float(int(round(factor * first)) - first) / n
    frames, cur = [], first
    for _ in range(n):
        cur = int(round(cur + step)) if use_round else int(floor(cur + step))
        frames.append(cur)
    return frames


def frame_contraction(first, nout, n, use_round=False):
    """n channel counts shrinking linearly from ``first`` towards ``nout`` (reference :400-408)."""
    step = float(first - nout) / n
    frames, cur = [], first
    for _ in range(n):
        cur = int(round(cur - step)) if use_round else int(floor(cur - step))
        frames.append(cur)
    return frames


def expansion_contraction_frames(nin, nout, n, pointwise_factor, expansion_factor, n_expansion):
    """Channel counts of the expansion / contraction schedules (block versions 2 and 3, preserve version 0):
    [nin] (+ one pointwise step) + n_expansion growing steps + the remaining shrinking steps."""
    n_contraction = n - n_expansion - (1 if pointwise_factor > 0 else 0)
    if n_contraction < 1:
        raise ValueError("n_contraction too large, must be < n - 1" if pointwise_factor > 0
                         else "n_contraction too large, must be < n")
    frames = [nin]
    if pointwise_factor > 0:
        frames.append(nin - int(floor((nin - nout) * pointwise_factor)))
    if n_expansion > 0:
        frames += frame_expansion(frames[-1], expansion_factor, n_expansion)
    frames += frame_contraction(frames[-1], nout, n_contraction)
    return frames


def late_decay(i, n, pointwise_first):
    """The decay factor versions 1-3 scale padding (and, in version 3, the kernel) with: 1 at the first spatial
    layer, 0 at the last (reference :553-563)."""
    if n <= 1:
        return 1.
    return 1. - (i - 1) / (n - 1) if pointwise_first else 1. - i / (n - 1)


def block_layers(version, nin, nout, n, size_factor=3, pad_factor=0.0, stride_factor=1, dil_factor=1, pointwise_factor=0,
                 depth_factor=0, expansion_factor=0, n_expansion=0):
    """[(cin, cout, kernel, stride, padding, dilation)] of a SparseConv2DBlock, versions 0-3 (reference :450-727).
    Versions 1-3 divide by (n - 1) for the stride, as the reference does: n = 1 raises ZeroDivisionError there too."""
    pw = pointwise_factor > 0
    if version in (0, 1):
        frames = channel_schedule(nin, nout, n, pointwise_factor, depth_factor)
    else:
        frames = expansion_contraction_frames(nin, nout, n, pointwise_factor, expansion_factor, n_expansion)
    layers = []
    for i in range(n):
        if version == 0:
            fs, st, pd, dil = layer_hyperparameters(i, n, size_factor, pad_factor, stride_factor, dil_factor, pw)
        else:
            decay = late_decay(i, n, pw)
            fs = int(ceil(size_factor * decay)) if version == 3 else int(floor(size_factor / (i + 1.)))
            fs = max(fs, 2)
            st = max(int(round(stride_factor * i / (n - 1))), 1)
            dil = int(round(dil_factor ** i))
            if version == 1:
                pd = int(round(pad_factor * (fs - 1) * dil_factor * decay))
            else:
                pd = int(round(pad_factor * ((fs - 1) / 2.) * dil_factor * decay))
            if i == 0 and pw:
                fs, st, pd, dil = 1, 1, 0, 1
        layers.append((frames[i], frames[i + 1], fs, st, pd, dil))
    return layers


class SparseConv2DBlock(object):
    """n x (SparseConv2d -> BatchNorm1d -> ReLU [-> Dropout]) [-> ToDense]; ``version`` picks the schedule arithmetic
    (0: GEP.json's; 1: late-decaying padding; 2 / 3: channel expansion then contraction, 3 with a shrinking kernel)."""

    def __init__(self, spconv, nin, nout, n, size, to_dense, size_factor=3, pad_factor=0.0, stride_factor=1,
                 dil_factor=1, pointwise_factor=0, depth_factor=0, dropout=0, trainable_weights=False,
                 version=0, expansion_factor=0, n_expansion=0, **unused):
        if version not in (0, 1, 2, 3):
            raise ValueError("no version {} available".format(version))
        if version in (0, 1):
            assert n > 0
        self.ndim = len(size) - 1
        self.out_size = list(size)
        self.alg = []
        self.schedule = []
        plan = block_layers(version, nin, nout, n, size_factor, pad_factor, stride_factor, dil_factor, pointwise_factor,
                            depth_factor, expansion_factor, n_expansion)
        frames = [p_[0] for p_ in plan] + [plan[-1][1]]
        for i in range(n):
            fs, st, pd, dil = plan[i][2:]
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


def preserve_layers(version, nin, nout, n=None, size_factor=3, pad_factor=0.0, stride_factor=1, dil_factor=1,
                    pointwise_factor=0, expansion_factor=0, n_expansion=0, n_contraction=1, filter_multiplier=1.0):
    """[(cin, cout, kernel, stride, padding, dilation, indice_key)] of a SparseConv2DPreserve (reference :756-945).
    Version 0: every layer is a SparseConv2d FOLLOWED by the SparseInverseConv2d of the same key (back on the input's
    sites); versions 1 / 2: SubMConv2d layers with "same" padding that share rulebooks per kernel size."""
    pw = pointwise_factor > 0
    layers = []
    if version == 0:
        frames = expansion_contraction_frames(nin, nout, n, pointwise_factor, expansion_factor, n_expansion)
        frames[-1] = nout
        for i in range(n):
            decay = late_decay(i, n, pw)
            fs = max(int(ceil(size_factor * decay)), 2)
            st = max(int(round(stride_factor * i / (n - 1))), 1)
            dil = int(round(dil_factor ** i))
            pd = int(round(pad_factor * ((fs - 1) / 2.) * dil_factor * decay))
            if i == 0 and pw:
                fs, st, pd, dil = 1, 1, 0, 1
            layers.append((frames[i], frames[i + 1], fs, st, pd, dil, "ind_{}".format(i)))
        return layers
    if version not in (1, 2):
        raise ValueError("no version {} available".format(version))
    n = n_contraction + n_expansion
    if pw:
        n_expansion -= 1
    if n < 1:
        raise ValueError("n_contraction + n_expansion must be >=1")
    if size_factor % 2 != 1:
        raise ValueError("size factor must be odd if version == {}".format(version))
    frames = [nin]
    if pw:
        frames.append(int(nin * pointwise_factor))
    if n_expansion > 0:
        frames += frame_expansion(frames[-1], expansion_factor, n_expansion)
    if n_contraction > 0:
        frames += frame_contraction(frames[-1], nout, n_contraction)
    frames[-1] = nout
    for i in range(n):
        if version == 1:
            fs = int(ceil(size_factor * late_decay(i, n, pw)))
        else:
            want = size_factor * (filter_multiplier ** i)
            near = int(round(want))
            up = (near % 2 == 0) == (near - want > 0)          # an even nearest integer moves away from `want`'s side
            fs = int(ceil(want)) if up else int(floor(want))
        if fs % 2 != 1:
            fs -= 1
        fs = max(fs, 3)
        pd = int((fs - 1) / 2)
        base = "ind_0" if version == 1 else "subm0"
        if i == 0 and pw:
            layers.append((frames[i], frames[i + 1], 1, 1, 0, 1, base))
        else:
            key = base if fs < 4 else ("ind_{}" if version == 1 else "subm{}").format(fs)
            layers.append((frames[i], frames[i + 1], fs, 1, pd, 1, key))
    return layers


class SparseConv2DPreserve(nn.Module):
    """Stacks that keep the input's active sites (reference src/models/SPConvBlocks.py:730-948; used by
    SPConvPreserveNet, config/examples/IoniClassifierCNN.json): per layer conv -> [inverse conv] -> BatchNorm1d -> ReLU
    [-> Dropout], the result a SparseConvTensor on the input's row set."""

    def __init__(self, spconv, nin, nout, n, size_factor=3, pad_factor=0.0, stride_factor=1, dil_factor=1,
                 pointwise_factor=0, dropout=0, trainable_weights=False, expansion_factor=0, n_expansion=0, version=0,
                 n_contraction=1, filter_multiplier=1.0):
        super().__init__()
        self.plan = preserve_layers(version, nin, nout, n, size_factor, pad_factor, stride_factor, dil_factor,
                                    pointwise_factor, expansion_factor, n_expansion, n_contraction, filter_multiplier)
        self.alg = []
        for i, (cin, cout, fs, st, pd, dil, key) in enumerate(self.plan):
            conv = spconv.SparseConv2d if version == 0 else spconv.SubMConv2d
            if i == 0 and pointwise_factor > 0:
                self.alg.append(conv(cin, cout, fs, st, pd, dil, 1, bias=trainable_weights, indice_key=key))
            else:   # `trainable_weights` lands in spconv's positional `bias` slot, as in the reference (:804, :876)
                self.alg.append(conv(cin, cout, fs, st, pd, dil, 1, trainable_weights, indice_key=key))
            if version == 0:
                self.alg.append(spconv.SparseInverseConv2d(cout, cout, fs, key, bias=trainable_weights))
            self.alg.append(nn.BatchNorm1d(cout))
            self.alg.append(nn.ReLU())
            if dropout:
                self.alg.append(nn.Dropout(dropout))
        self.func = spconv.SparseSequential(*self.alg)

    def forward(self, x):
        return self.func(x)


class LinearBlock(object):
    """n geometrically interpolated nn.Linear layers, no activations (reference ConvBlocks.py:82-102)."""

    def __init__(self, nin, nout, n):
        assert n > 0 and nin > 0
        factor = pow(float(nout) / nin, 1. / n)
        self.widths = [int(round(nin * pow(factor, i))) for i in range(n + 1)]
        self.alg = [nn.Linear(self.widths[i], self.widths[i + 1]) for i in range(n)]
        self.func = nn.Sequential(*self.alg)

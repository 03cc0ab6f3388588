"""Per-segment regression stacks on the spconv surface: host-side mirrors of the reference's ``SparseConv2DForZ``,
``Pointwise2DForZ`` and ``SparseConv2DForEZ`` (versions 0-3) block builders (src/models/SPConvBlocks.py:261-343, 9-257).

Each is a ``SparseSequential`` of *regular* ``SparseConv2d`` layers with "same" padding ((k-1)/2), ``BatchNorm1d``
between layers, ``ReLU`` after each and ``ToDense`` at the end: [N, 2T] waveform rows on the 14 x 11 PMT grid ->
a dense [B, out, 14, 11] map of per-segment predictions.  The layer SCHEDULES (channel counts, kernel sizes) are the
reference's arithmetic, restated as plain functions so that they can be tested without building modules.
"""
from math import ceil

from torch import nn

from .blocks import frame_contraction, frame_expansion


def z_schedule(in_planes, kernel_size=3, n_layers=2, pointwise_layers=0, pointwise_factor=0.8):
    """[(cin, cout, k, pad, batchnorm)] of SparseConv2DForZ (reference :261-310)."""
    if pointwise_layers > 0:
        if n_layers == 1:
            raise ValueError("n_layers must be > 1 if using pointwise convolution")
        step = int(round(int(round(in_planes * pointwise_factor)) / float(n_layers - 1)))
    else:
        step = int(round(float(in_planes) / float(n_layers)))
    if kernel_size % 2 != 1:
        raise ValueError("Kernel size must be an odd integer")
    if not isinstance(n_layers, int) or n_layers < 1:
        raise ValueError("n_layers must be  integer >= 1")
    # state: k = the kernel the next spatial layer would use.  A leading 1x1 layer pins k to 1; after the LAST leading
    # 1x1 layer k is restored to kernel_size; after every layer a k > 1 shrinks by 2 -- so with kernel_size = 3 the
    # layer that follows the 1x1 block is again 1x1 (the reference's behaviour, kept)
    plan, cin, cout, k, left = [], in_planes, in_planes, kernel_size, pointwise_layers
    for i in range(n_layers):
        last = i == n_layers - 1
        if last:
            cout = 1
        else:
            cout -= step
            if i == 0 and pointwise_layers > 0 and pointwise_factor > 0:
                cout = int(round(pointwise_factor * in_planes))
        pad, restore = int((k - 1) / 2), False
        if left > 0:
            k, pad = 1, 0
            left -= 1
            restore = left == 0
        plan.append((cin, cout, k, pad, not last))
        cin = cout
        if restore:
            k = kernel_size
        if k > 1:
            k -= 2
    return plan


def point_schedule(in_planes, pointwise_layers=2):
    """[(cin, cout, 1, 0, True)] of Pointwise2DForZ (reference :316-340): BatchNorm after EVERY layer, the last too."""
    n = pointwise_layers
    if not isinstance(n, int) or n < 2:
        raise ValueError("n_layers must be  integer >= 2")
    step = int(round(float(in_planes) / float(n - 1)))
    plan, cin, cout = [], in_planes, in_planes
    for i in range(n):
        if i == n - 1:
            cout = 1
        elif i == 0:
            cout = in_planes
        else:
            cout -= step
        plan.append((cin, cout, 1, 0, True))
        cin = cout
    return plan


def _subm_key(k):
    """Versions 1-3 share ONE rulebook between all layers up to 3 x 3 and one per larger kernel (reference :131-134)."""
    return "subm0" if k < 4 else "subm{}".format(k)


def ez_schedule(in_planes, out_planes=2, kernel_size=3, n_conv=1, n_point=3, conv_position=3, pointwise_factor=0.8,
                batchnorm=True, version=0, n_expand=0):
    """SparseConv2DForEZ layer plans (reference :31-257).  Version 0: [(cin, cout, k, pad, batchnorm)] of regular
    SparseConv2d layers; versions 1-3: [(cin, cout, k, pad, batchnorm, indice_key)] of SubMConv2d layers --
    1: version 0's arithmetic (channel counts floored at 1); 2: every spatial layer keeps ``kernel_size``;
    3: rounded channel expansion (by ``pointwise_factor``, which the reference passes as the expansion factor) over
    ``n_expand`` layers then contraction, the spatial kernels shrinking linearly to 3."""
    if version not in (0, 1, 2, 3):
        raise IOError("no version {} available, choose a version <= 1".format(version))
    n_layers = n_conv + n_point
    if version == 3:
        if n_conv > 0 and conv_position < 1:
            raise ValueError("conv position must be >= 1 if n_conv > 0")
        if n_point > 0 and n_layers == 1:
            raise ValueError("n_layers must be > 1 if using pointwise convolution")
        n_contraction = n_layers - n_expand
        if n_contraction < 1:
            raise ValueError("n expand must be <= (n_point + n_conv - 1)")
        if kernel_size % 2 != 1:
            raise ValueError("Kernel size must be an odd integer")
        if not isinstance(n_conv, int) or n_conv < 1:
            raise ValueError("n_conv must be an integer >= 1 ")
        frames = [in_planes]
        if n_expand > 0:
            frames += frame_expansion(frames[-1], pointwise_factor, n_expand, True)
        frames += frame_contraction(frames[-1], out_planes, n_contraction, True)
        frames[-1] = out_planes
        first = conv_position - 1
        plan = []
        for i in range(n_layers):
            if first <= i < first + n_conv:
                decay = 1. - (i - first) / (n_conv - 1) if n_conv > 1 else 1.
                k = int(ceil(kernel_size * decay))
                if k % 2 == 0:
                    k -= 1
                k = max(k, 3)
                pad = int((k - 1) / 2)
            else:
                k, pad = 1, 1                       # the reference's padding for its 1 x 1 layers; a 1 x 1 conv ignores it
            plan.append((frames[i], frames[i + 1], k, pad, batchnorm and i != n_layers - 1, _subm_key(k)))
        return plan
    if n_conv > 0 and conv_position < 1:
        raise ValueError("conv position must be >= 1 if n_conv > 0")
    if n_point > 0:
        if n_layers == 1:
            raise ValueError("n_layers must be > 1 if using pointwise convolution")
        step = int(round(int(round(in_planes * pointwise_factor - out_planes)) / float(n_layers - 1)))
    else:
        step = int(round(float(in_planes - out_planes) / float(n_layers)))
    if kernel_size % 2 != 1:
        raise ValueError("Kernel size must be an odd integer")
    if not isinstance(n_layers, int) or n_layers < 1:
        raise ValueError("n_layers must be  integer >= 1")
    spatial = set(range(conv_position - 1, conv_position - 1 + n_conv)) if n_conv > 0 else set()
    plan, cin, cout = [], in_planes, in_planes
    for i in range(n_layers):
        last = i == n_layers - 1
        if last:
            cout = out_planes
        else:
            cout -= step
            if i == 0 and n_point > 0 and pointwise_factor > 0:
                cout = int(round(pointwise_factor * in_planes))
        if i not in spatial:
            k = 1
        elif version == 2:
            k = max(kernel_size, 3)
        else:
            k = max(kernel_size - int((i + 1 - conv_position) * 2), 3)
        if k % 2 == 0:
            raise ValueError("error: kernel size is even")
        if version == 1 and cout <= 0:
            cout = 1
        if version == 0:
            plan.append((cin, cout, k, int((k - 1) / 2), batchnorm and not last))
        else:
            plan.append((cin, cout, k, int((k - 1) / 2), batchnorm and not last, _subm_key(k)))
        cin = cout
    return plan


def _stack(spconv, plan, todense=True):
    layers = []
    for cin, cout, k, pad, bn, *key in plan:
        if key:
            layers.append(spconv.SubMConv2d(cin, cout, k, 1, pad, indice_key=key[0]))
        else:
            layers.append(spconv.SparseConv2d(cin, cout, k, 1, pad))
        if bn:
            layers.append(nn.BatchNorm1d(cout))
        layers.append(nn.ReLU())
    if todense:
        layers.append(spconv.ToDense())
    return spconv.SparseSequential(*layers)


class _Block(nn.Module):
    def forward(self, x):
        return self.network(x)


class SparseConv2DForZ(_Block):
    def __init__(self, spconv, in_planes, kernel_size=3, n_layers=2, pointwise_layers=0, pointwise_factor=0.8,
                 todense=True):
        super().__init__()
        self.plan = z_schedule(in_planes, kernel_size, n_layers, pointwise_layers, pointwise_factor)
        self.network = _stack(spconv, self.plan, todense)


class Pointwise2DForZ(_Block):
    def __init__(self, spconv, in_planes, pointwise_layers=2):
        super().__init__()
        self.plan = point_schedule(in_planes, pointwise_layers)
        self.network = _stack(spconv, self.plan)


class SparseConv2DForEZ(_Block):
    def __init__(self, spconv, in_planes, out_planes=2, kernel_size=3, n_conv=1, n_point=3, conv_position=3,
                 pointwise_factor=0.8, batchnorm=True, version=0, n_expand=0):
        super().__init__()
        self.plan = ez_schedule(in_planes, out_planes, kernel_size, n_conv, n_point, conv_position, pointwise_factor,
                                batchnorm, version, n_expand)
        self.network = _stack(spconv, self.plan)

"""Shared test helpers: seeded sparse inputs and the dense-conv value oracle."""
import numpy as np
import torch
import torch.nn.functional as F


def rand_coords(rng, batch, shape, n, batch_sorted=True):
    """n distinct active sites, int32 [n, D+1] batch-first; events contiguous (as collate_fn
    produces them, reference src/engineering/PSDDataModule.py:10-20), random order inside an event."""
    vol = int(np.prod(shape))
    sel = rng.choice(batch * vol, size=n, replace=False)
    if batch_sorted:
        sel = sel[np.argsort(sel // vol, kind="stable")]
    pos = sel % vol
    cols = [sel // vol]
    for d in range(len(shape)):
        stride = int(np.prod(shape[d + 1:]))
        cols.append((pos // stride) % shape[d])
    return np.stack(cols, 1).astype(np.int32)


def densify(idx, feat, batch, shape):
    C = feat.shape[1]
    d = np.zeros([batch, C] + list(shape), np.float32)
    for r, i in enumerate(idx):
        d[(i[0], slice(None)) + tuple(i[1:])] = feat[r]
    return d


def dense_conv_at(idx, feat, W, batch, shape, out_idx, stride, padding, dilation):
    """F.conv{1,2,3}d on the densified input, sampled at out_idx.  W is [*k, Cin, Cout]."""
    ndim = len(shape)
    d = torch.from_numpy(densify(idx, feat, batch, shape)).double()
    Wt = torch.from_numpy(np.asarray(W)).double().permute(ndim + 1, ndim, *range(ndim)).contiguous()
    conv = {1: F.conv1d, 2: F.conv2d, 3: F.conv3d}[ndim]
    y = conv(d, Wt, None, stride, padding, dilation).numpy()
    got = np.stack([y[(i[0], slice(None)) + tuple(i[1:])] for i in out_idx]) if len(out_idx) else np.zeros((0, W.shape[-1]))
    mask = np.zeros(y.shape, bool)
    for i in out_idx:
        mask[(i[0], slice(None)) + tuple(i[1:])] = True
    outside = float(np.abs(y[~mask]).max()) if (~mask).any() else 0.0
    return got, outside


CASES = [
    # ndim, shape, ksize, stride, padding, dilation, subm
    (2, (7, 6), 3, 1, 0, 1, True),
    (3, (5, 6, 7), 3, 1, 0, 1, True),
    (3, (14, 11, 24), 3, 1, 0, 1, True),
    (2, (14, 11), (3, 1), 1, 0, 1, True),
    (2, (9, 8), 3, 2, 1, 1, False),
    (2, (14, 11), 3, 1, 0, 1, False),
    (3, (6, 7, 9), (3, 3, 3), (1, 1, 2), (0, 1, 1), 1, False),
    (2, (9, 9), 3, 1, 2, 2, False),
    (3, (4, 5, 16), (2, 3, 3), (1, 2, 4), (0, 1, 0), 1, False),
    (3, (14, 11, 32), 3, (1, 1, 4), 0, 1, False),
    (1, (33,), 5, 3, 2, 1, False),
]


def norm(v, ndim):
    return [int(v)] * ndim if np.isscalar(v) else [int(x) for x in v]

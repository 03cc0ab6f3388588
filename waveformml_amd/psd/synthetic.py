"""Seeded synthetic PROSPECT-like waveform events (SURVEY.md 8d / BASELINE.md 3).

No physics files exist offline, so benchmarks and tests draw events from this generator:
multiplicity 1+Poisson(2) clipped to [1,10] on a random-walk cluster of the 14x11 segment grid; two
PMT channels per hit; 3-sample rise + two-exponential decay (tau 5 / 30 samples, slow fraction
0.1+0.15*class); amplitude LogUniform[200,12000] ADC split Beta(4,4) between the PMTs; Gaussian
noise sigma 3 ADC; integers clipped to 14 bit, zero-suppressed below 8 ADC, scaled by 1/(2^14-1)
(reference src/datasets/HDF5Dataset.py:14-17,345-346).

Layouts (reference src/datasets/PulseDataset.py:543-625):
  2-D: coord int32 [n,3] = (x, y, evt),    feat [n, 2T]   (left samples then right samples)
  3-D: coord int32 [n,4] = (x, y, t, evt), feat [n, 2]    one voxel per sample with either PMT >= 8 ADC
"""
import numpy as np

NX, NY = 14, 11
ADC_MAX = 2 ** 14 - 1
THRESHOLD = 8


def _cluster(rng, m):
    cells = [(int(rng.integers(NX)), int(rng.integers(NY)))]
    tries = 0
    while len(cells) < m and tries < 200:
        tries += 1
        x, y = cells[int(rng.integers(len(cells)))]
        dx, dy = [(1, 0), (-1, 0), (0, 1), (0, -1)][int(rng.integers(4))]
        c = (x + dx, y + dy)
        if 0 <= c[0] < NX and 0 <= c[1] < NY and c not in cells:
            cells.append(c)
    return cells


def generate(n_events, n_samples, n_type=3, seed=1234, rank=0, layout="3d", label=None):
    """Returns (coords int32, feats float32, labels int64) in the reference's dataset layout.  ``label``: every event of
    this class (the files of one class directory, tools/soak_from_files.py) instead of a random class per event."""
    rng = np.random.default_rng(seed + rank)
    T = int(n_samples)
    t = np.arange(T, dtype=np.float64)
    labels = (rng.integers(0, n_type, size=n_events).astype(np.int64) if label is None
              else np.full((n_events,), int(label), np.int64))
    coords, feats = [], []
    for e in range(n_events):
        m = int(np.clip(1 + rng.poisson(2), 1, 10))
        slow = 0.1 + 0.15 * int(labels[e])
        for (x, y) in _cluster(rng, m):
            t0 = rng.uniform(0.08 * T, 0.16 * T)
            amp = np.exp(rng.uniform(np.log(200.0), np.log(12000.0)))
            split = rng.beta(4, 4)
            dt = t - t0
            rise = np.clip(dt / 3.0, 0.0, 1.0)
            decay = np.where(dt > 3.0, (1 - slow) * np.exp(-(dt - 3.0) / 5.0) + slow * np.exp(-(dt - 3.0) / 30.0), 1.0)
            shape = rise * decay
            wf = np.stack([amp * split * shape, amp * (1 - split) * shape])
            wf = np.rint(wf + rng.normal(0.0, 3.0, size=wf.shape))
            wf = np.clip(wf, 0, ADC_MAX)
            wf[wf < THRESHOLD] = 0
            if layout == "2d":
                if wf.any():
                    coords.append(np.array([[x, y, e]], np.int32))
                    feats.append(wf.reshape(1, 2 * T))
            else:
                on = np.nonzero((wf[0] > 0) | (wf[1] > 0))[0]
                if len(on):
                    c = np.empty((len(on), 4), np.int32)
                    c[:, 0], c[:, 1], c[:, 2], c[:, 3] = x, y, on, e
                    coords.append(c)
                    feats.append(wf[:, on].T)
    width = 2 * T if layout == "2d" else 2
    ncol = 3 if layout == "2d" else 4
    if coords:
        coords = np.concatenate(coords).astype(np.int32)
        feats = (np.concatenate(feats) * (1.0 / ADC_MAX)).astype(np.float32)
    else:
        coords, feats = np.zeros((0, ncol), np.int32), np.zeros((0, width), np.float32)
    # every event id 0..n_events-1 must own at least one row for `coords[-1,-1]+1` to give the batch
    # size (reference src/models/SPConvNet.py:63); amplitudes >= 200 ADC guarantee that here.
    return coords, feats, labels

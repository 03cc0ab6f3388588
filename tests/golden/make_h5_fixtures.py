"""Writes the small HDF5 fixtures under tests/golden/h5/ and their expected arrays (expected.npz).

Run with the image's h5py interpreter (python3.10 has no h5py):   /opt/conda/bin/python3.9 tests/golden/make_h5_fixtures.py

The reference ships no data files, so these are synthetic events written in the reference's two on-disk layouts:
  * compound tables "WaveformPairs" (2-D: coord int32[3] = (x, y, evt), waveform int16[2T]) and "Waveform3DPairs"
    (3-D: coord int32[4] = (x, y, t, evt), waveform float32[2]) with the surrounding members of
    reference src/datasets/H5CompoundTypes.py:105-120 (evt, t, dt, z, E, PSD, PE, ..., EZ, PID) and attribute nevents
    -- one file per class directory, as src/datasets/PulseDataset.py:543-625 expects ("*WaveformPairSim.h5");
  * the group layout of "combined" files (reference src/datasets/PulseDataset.py:312-333): <table>/coord,
    <table>/waveform, <table>/labels (int8), gzip-6 chunks, attribute nevents on the group.
"""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "h5")
T = 10          # samples per PMT channel in the 2-D fixtures
rng = np.random.default_rng(77)
expected = {}


def events_2d(n_events, label):
    coords, wfs = [], []
    for e in range(n_events):
        for _ in range(int(rng.integers(1, 4))):
            coords.append([int(rng.integers(0, 14)), int(rng.integers(0, 11)), e])
            wfs.append(rng.integers(0, 2 ** 14, size=2 * T))
    return np.asarray(coords, np.int32), np.asarray(wfs, np.int16)


def write_compound_2d(path, n_events):
    coords, wfs = events_2d(n_events, 0)
    n = len(coords)
    dt = np.dtype([("evt", "<i8"), ("t", "<f8"), ("dt", "<f4"), ("z", "<f4"), ("E", "<f4"), ("PSD", "<f4"),
                   ("PE", "<f4", (2,)), ("coord", "<i4", (3,)), ("waveform", "<i2", (2 * T,)), ("EZ", "<f4", (2,)),
                   ("PID", "<i4")])
    rec = np.zeros(n, dt)
    rec["evt"] = coords[:, 2]
    rec["t"] = rng.random(n)
    rec["E"] = rng.random(n).astype(np.float32)
    rec["coord"] = coords
    rec["waveform"] = wfs
    rec["PID"] = 3
    with h5py.File(path, "w") as f:
        d = f.create_dataset("WaveformPairs", data=rec)
        d.attrs.create("nevents", np.array([n_events]))
    return coords, wfs


def write_compound_3d(path, n_events):
    coords, feats = [], []
    for e in range(n_events):
        x, y = int(rng.integers(0, 14)), int(rng.integers(0, 11))
        t0 = int(rng.integers(0, 20))
        for t in range(t0, t0 + int(rng.integers(3, 9))):
            coords.append([x, y, t, e])
            feats.append(rng.random(2))
    coords, feats = np.asarray(coords, np.int32), np.asarray(feats, np.float32)
    dt = np.dtype([("evt", "<i8"), ("coord", "<i4", (4,)), ("waveform", "<f4", (2,)), ("PID", "<i4")])
    rec = np.zeros(len(coords), dt)
    rec["evt"] = coords[:, 3]
    rec["coord"] = coords
    rec["waveform"] = feats
    with h5py.File(path, "w") as f:
        d = f.create_dataset("Waveform3DPairs", data=rec)
        d.attrs.create("nevents", np.array([n_events]))
    return coords, feats


def write_group_2d(path, n_events):
    coords, wfs = events_2d(n_events, 0)
    labels = rng.integers(0, 3, size=n_events).astype(np.int8)
    with h5py.File(path, "w") as f:
        f.create_dataset("WaveformPairs/coord", compression="gzip", compression_opts=6, data=coords,
                         chunks=(min(16, len(coords)), 3))
        f.create_dataset("WaveformPairs/waveform", compression="gzip", compression_opts=6, data=wfs,
                         chunks=(min(16, len(wfs)), 2 * T))
        f.create_dataset("WaveformPairs/labels", compression="gzip", compression_opts=6, data=labels,
                         chunks=(n_events,), dtype=np.int8)
        f["WaveformPairs"].attrs.create("nevents", np.array([n_events]))
    return coords, wfs, labels


for cls, name, ne in (("Gamma", "a_WaveformPairSim.h5", 9), ("Gamma", "b_WaveformPairSim.h5", 5),
                      ("Electron", "a_WaveformPairSim.h5", 12)):
    c, w = write_compound_2d(os.path.join(OUT, cls, name), ne)
    expected["%s/%s/coord" % (cls, name)] = c
    expected["%s/%s/waveform" % (cls, name)] = w
c, w = write_compound_3d(os.path.join(OUT, "Gamma", "a_Waveform3DPairSim.h5"), 7)
expected["Gamma/a_Waveform3DPairSim.h5/coord"] = c
expected["Gamma/a_Waveform3DPairSim.h5/waveform"] = w
c, w, l = write_group_2d(os.path.join(OUT, "combined", "Combined_0_WaveformPairSim.h5"), 11)
expected["combined/Combined_0_WaveformPairSim.h5/coord"] = c
expected["combined/Combined_0_WaveformPairSim.h5/waveform"] = w
expected["combined/Combined_0_WaveformPairSim.h5/labels"] = l
for cls, name, ne in (("Electron", "a_Waveform3DPairSim.h5", 6), ("Gamma", "b_Waveform3DPairSim.h5", 4)):
    c, w = write_compound_3d(os.path.join(OUT, cls, name), ne)
    expected["%s/%s/coord" % (cls, name)] = c
    expected["%s/%s/waveform" % (cls, name)] = w


def write_compound_3d_gzip(path, n_events, shuffle_events=False):
    """the same compound table, chunked (64 records) + gzip-6: what libwfh5's chunk-parallel path reads"""
    coords, feats = [], []
    order = list(range(n_events))
    if shuffle_events:
        order = order[::2] + order[1::2]                 # event ids NOT ascending: the search must fall back to a scan
    for e in order:
        for _ in range(int(rng.integers(1, 3))):
            x, y, t0 = int(rng.integers(0, 14)), int(rng.integers(0, 11)), int(rng.integers(0, 20))
            for t in range(t0, t0 + int(rng.integers(5, 30))):
                coords.append([x, y, t, e])
                feats.append(rng.random(2))
    coords, feats = np.asarray(coords, np.int32), np.asarray(feats, np.float32)
    dt = np.dtype([("evt", "<i8"), ("coord", "<i4", (4,)), ("waveform", "<f4", (2,)), ("PID", "<i4")])
    rec = np.zeros(len(coords), dt)
    rec["evt"] = coords[:, 3]
    rec["coord"] = coords
    rec["waveform"] = feats
    with h5py.File(path, "w") as f:
        d = f.create_dataset("Waveform3DPairs", data=rec, chunks=(64,), compression="gzip", compression_opts=6)
        d.attrs.create("nevents", np.array([n_events]))
    return coords, feats


os.makedirs(os.path.join(OUT, "gz"), exist_ok=True)
for name, ne, shuf in (("sorted_Waveform3DPairSim.h5", 23, False), ("unsorted_Waveform3DPairSim.h5", 10, True)):
    c, w = write_compound_3d_gzip(os.path.join(OUT, "gz", name), ne, shuf)
    expected["gz/%s/coord" % name] = c
    expected["gz/%s/waveform" % name] = w
np.savez_compressed(os.path.join(OUT, "expected.npz"), **expected)
print("wrote", sorted(expected))

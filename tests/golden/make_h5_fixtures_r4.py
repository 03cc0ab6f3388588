"""Round-4 HDF5 fixtures under tests/golden/h5/r4/ (+ expected_r4.npz): label columns of other stored types than int32
and a label file with FLOAT labels -- the cases in which the reference's label handling
(src/datasets/HDF5Dataset.py:319-341, convert_label :582-585) differs by path.

Run with the image's h5py interpreter:   /opt/conda/bin/python3.9 tests/golden/make_h5_fixtures_r4.py
"""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "h5", "r4", "labels")
os.makedirs(OUT, exist_ok=True)
rng = np.random.default_rng(404)
T, n_events = 4, 6
coords, wf = [], []
for e in range(n_events):
    for _ in range(int(rng.integers(1, 4))):
        coords.append([int(rng.integers(0, 14)), int(rng.integers(0, 11)), e])
        wf.append(rng.integers(0, 2 ** 14, size=2 * T))
n = len(coords)
pid = rng.choice(np.array([1, 4, 6, 7]), n)
dt = np.dtype([("evt", "<i8"), ("coord", "<i4", (3,)), ("waveform", "<i2", (2 * T,)), ("PID", "<i4"), ("PID16", "<i2"),
               ("PID8", "<i1"), ("PID64", "<i8")])
rec = np.zeros(n, dt)
rec["evt"] = np.asarray(coords)[:, 2]
rec["coord"] = np.asarray(coords, np.int32)
rec["waveform"] = np.asarray(wf, np.int16)
for m in ("PID", "PID16", "PID8", "PID64"):
    rec[m] = pid
with h5py.File(os.path.join(OUT, "lab_1_WaveformPairSim.h5"), "w") as f:
    d = f.create_dataset("WaveformPairs", data=rec, chunks=(8,), compression="gzip", compression_opts=6)
    d.attrs.create("nevents", np.array([n_events]))
lab = np.zeros(n_events, np.dtype([("label", "<f4"), ("weight", "<f4")]))
lab["label"] = rng.choice(np.array([0.0, 1.0, 2.5, 3.0], np.float32), n_events)
with h5py.File(os.path.join(OUT, "lab_1_FLabel.h5"), "w") as f:
    d = f.create_dataset("EventLabels", data=lab)
    d.attrs.create("nevents", np.array([n_events]))
np.savez(os.path.join(HERE, "expected_r4.npz"), pid=pid, flabel=lab["label"], coord=rec["coord"])
print("wrote", n, "rows,", n_events, "events")

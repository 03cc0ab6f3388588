"""Round-3 HDF5 fixtures under tests/golden/h5/r3/ (+ expected_r3.npz): the reference's dataset features beyond the PSD
tables -- per-row label columns, additional fields, label files, other member names, length-based ranges.

Run with the image's h5py interpreter:   /opt/conda/bin/python3.9 tests/golden/make_h5_fixtures_r3.py

Written the way the reference's files are laid out (it ships none): compound tables after
src/datasets/H5CompoundTypes.py -- a `WaveformPairCal`-like record (evt, t, ..., coord, waveform, EZ, PID plus a `phys`
float32[8] vector) in "*WaveformPairSim.h5", a detector-pulse record with members `det` / `pulse` in "*PulseNorm.h5",
and label files "*Label.h5" (reference `label_file_pattern`, src/datasets/HDF5Dataset.py:404-427) whose `label_name`
table carries the label as its FIRST member, one record per event.
"""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "h5", "r3")
T = 8
rng = np.random.default_rng(303)
exp = {}
PIDS = np.array([1, 4, 6, 256, 258, 512], np.int32)


def seg_file(path, n_events, key, T=T):
    coords, wf, pid, phys, ez = [], [], [], [], []
    for e in range(n_events):
        for _ in range(int(rng.integers(1, 5))):
            coords.append([int(rng.integers(0, 14)), int(rng.integers(0, 11)), e])
            wf.append(rng.integers(0, 2 ** 14, size=2 * T))
            pid.append(int(rng.choice(PIDS)))
            phys.append(rng.random(8))
            ez.append(rng.random(2))
    n = len(coords)
    dt = np.dtype([("evt", "<i8"), ("t", "<f8"), ("E", "<f4"), ("PE", "<f4", (2,)), ("coord", "<i4", (3,)),
                   ("waveform", "<i2", (2 * T,)), ("phys", "<f4", (8,)), ("EZ", "<f4", (2,)), ("PID", "<i4")])
    rec = np.zeros(n, dt)
    rec["evt"] = np.asarray(coords)[:, 2]
    rec["t"] = rng.random(n)
    rec["coord"] = np.asarray(coords, np.int32)
    rec["waveform"] = np.asarray(wf, np.int16)
    rec["phys"] = np.asarray(phys, np.float32)
    rec["EZ"] = np.asarray(ez, np.float32)
    rec["PID"] = np.asarray(pid, np.int32)
    with h5py.File(path, "w") as f:
        d = f.create_dataset("WaveformPairs", data=rec, chunks=(16,), compression="gzip", compression_opts=6)
        d.attrs.create("nevents", np.array([n_events]))
    for m in ("coord", "waveform", "phys", "EZ", "PID"):
        exp[key + "/" + m] = rec[m]


def label_file(path, n_events, key):
    dt = np.dtype([("label", "<i4"), ("weight", "<f4")])
    rec = np.zeros(n_events, dt)
    rec["label"] = rng.integers(0, 4, n_events)
    rec["weight"] = rng.random(n_events)
    with h5py.File(path, "w") as f:
        d = f.create_dataset("EventLabels", data=rec)
        d.attrs.create("nevents", np.array([n_events]))
    exp[key + "/label"] = rec["label"]


def pulse_file(path, n_rows, key):
    dt = np.dtype([("t", "<f8"), ("evt", "<i8"), ("det", "<i4"), ("pulse", "<f4", (12,)), ("phys", "<f4", (3,)), ("PID", "<i4")])
    rec = np.zeros(n_rows, dt)
    rec["evt"] = np.arange(n_rows)
    rec["det"] = rng.integers(0, 308, n_rows)
    rec["pulse"] = rng.random((n_rows, 12))
    rec["phys"] = rng.random((n_rows, 3))
    rec["PID"] = rng.choice(PIDS, n_rows)
    with h5py.File(path, "w") as f:
        f.create_dataset("WaveformNorm", data=rec)          # no nevents attribute: ranges count rows (event_based=False)
    for m in ("det", "pulse", "phys", "PID"):
        exp[key + "/" + m] = rec[m]


os.makedirs(os.path.join(OUT, "ioni"), exist_ok=True)
os.makedirs(os.path.join(OUT, "pulses"), exist_ok=True)
seg_file(os.path.join(OUT, "ioni", "run_1_WaveformPairSim.h5"), 9, "ioni/run_1")
seg_file(os.path.join(OUT, "ioni", "run_2_WaveformPairSim.h5"), 7, "ioni/run_2")
label_file(os.path.join(OUT, "ioni", "run_1_Label.h5"), 9, "ioni/run_1")
label_file(os.path.join(OUT, "ioni", "run_2_Label.h5"), 7, "ioni/run_2")
pulse_file(os.path.join(OUT, "pulses", "p_1_PulseNorm.h5"), 23, "pulses/p_1")
# 65 samples per PMT (130 input channels: config/examples/IoniClassifierCNN.json) for the GPU test that trains the
# segment classifier from files
os.makedirs(os.path.join(OUT, "ioni130"), exist_ok=True)
for i, ne in enumerate((14, 12, 13)):
    seg_file(os.path.join(OUT, "ioni130", "seg_%d_WaveformPairSim.h5" % (i + 1)), ne, "ioni130/seg_%d" % (i + 1), T=65)
np.savez(os.path.join(HERE, "expected_r3.npz"), **exp)
print("wrote", sorted(exp))

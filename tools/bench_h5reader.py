"""Throughput of the native HDF5 -> COO reader (libwfh5.so) against the reference's way of reading the same file
(h5py: the WHOLE compound table into numpy, then event slicing with numpy.where -- src/datasets/HDF5Dataset.py:225-347,
430-476).  Host-only.  The h5py side runs in the image's conda interpreter (python3.10 has no h5py).

usage: python tools/bench_h5reader.py [events] [T]        (writes and deletes a scratch file under /tmp)
"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CONDA = "/opt/conda/bin/python3.9"
events = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256

WRITER = r'''
import sys, numpy as np, h5py
path, events, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(1)
coords = []
for e in range(events):
    for _ in range(int(rng.integers(1, 5))):
        x, y, t0 = int(rng.integers(0, 14)), int(rng.integers(0, 11)), int(rng.integers(0, T // 4))
        n = int(rng.integers(T // 8, T // 2))
        c = np.empty((n, 4), np.int32); c[:, 0] = x; c[:, 1] = y; c[:, 2] = np.arange(t0, t0 + n); c[:, 3] = e
        coords.append(c)
coords = np.concatenate(coords)
dt = np.dtype([("evt", "<i8"), ("t", "<f8"), ("dt", "<f4"), ("z", "<f4"), ("E", "<f4"), ("PSD", "<f4"), ("PE", "<f4", (2,)),
               ("coord", "<i4", (4,)), ("waveform", "<f4", (2,)), ("EZ", "<f4", (2,)), ("PID", "<i4")])
rec = np.zeros(len(coords), dt)
rec["coord"] = coords; rec["evt"] = coords[:, 3]; rec["waveform"] = rng.random((len(coords), 2)).astype(np.float32)
with h5py.File(path, "w") as f:
    d = f.create_dataset("Waveform3DPairs", data=rec, chunks=(4096,), compression="gzip", compression_opts=6)
    d.attrs.create("nevents", np.array([events]))
print(len(coords))
'''
READER = r'''
import sys, time, numpy as np, h5py
path, e0, e1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
t = time.perf_counter()
with h5py.File(path, "r") as f:
    data = f["Waveform3DPairs"][()]                      # whole table, every member (what _load_data caches)
coords, vals = data["coord"], data["waveform"]
a = np.where(coords[:, 3] == e0)[0][0] if e0 > 0 else 0
b = np.where(coords[:, 3] == e1 + 1)[0][0]
c, v = coords[a:b].astype(np.int32), vals[a:b].astype(np.float32)
print(time.perf_counter() - t, len(c))
'''

with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
    path = os.path.join(tmp, "bench_Waveform3DPairSim.h5")
    rows = int(subprocess.check_output([CONDA, "-c", WRITER, path, str(events), str(T)]).split()[-1])
    size = os.path.getsize(path)
    e0, e1 = events // 4, events // 4 + 255                 # one 256-event item out of the middle of the file
    ref_s, ref_rows = subprocess.check_output([CONDA, "-c", READER, path, str(e0), str(e1)]).split()
    from waveformml_amd.psd import h5data
    best = None
    for _ in range(3):
        t = time.perf_counter()
        with h5data.H5Table(path, "Waveform3DPairs") as tb:
            r0, r1 = tb.event_rows(e0, e1, 3)
            c, f = tb.read_rows(r0, r1)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    assert int(ref_rows) == r1 - r0
    whole = {}
    for threads in (1, 2, 4, 8):
        h5data.set_threads(threads)
        bw = None
        for _ in range(3):
            t = time.perf_counter()
            with h5data.H5Table(path, "Waveform3DPairs") as tb:
                c_all, f_all = tb.read_rows(0, tb.n_rows)
            dt = time.perf_counter() - t
            bw = dt if bw is None else min(bw, dt)
        whole[threads] = bw
    print(json.dumps({"file_rows": rows, "file_events": events, "file_MB": round(size / 1e6, 1), "item_events": 256,
                      "item_rows": r1 - r0, "h5py_whole_table_then_slice_s": round(float(ref_s), 4),
                      "wfh5_item_s": round(best, 4), "speedup_item": round(float(ref_s) / best, 1),
                      "wfh5_whole_file_s_by_threads": {str(k): round(v, 4) for k, v in whole.items()},
                      "wfh5_whole_file_Mrows_per_s_by_threads": {str(k): round(rows / v / 1e6, 2) for k, v in whole.items()},
                      "wfh5_whole_file_kevents_per_s_by_threads": {str(k): round(events / v / 1e3, 1) for k, v in whole.items()}}))

"""Soak run of the captured training loop FROM FILES, end to end through the reference-shaped data path:
HDF5 files on disk (written here with h5py under the image's conda interpreter, gzip-chunked compound tables as the
reference's converters write them) -> libwfh5 reader inside DataLoader workers -> PSDDataModule / PulseDataset3D / collate ->
DevicePrefetcher -> Trainer(capture=True): bf16 rows, HIP-graph step.  Prints one JSON object with
  * loader: events/s of the DataLoader alone for several worker counts (the per-rank loader budget),
  * train: steps, events/s including loading, eager fallbacks, overflow checks, loss per epoch.
The three classes differ in their pulse decay constant, so the loss has something to learn.

usage: python tools/soak_from_files.py [files_per_class] [events_per_file] [epochs] [workers]
"""
import copy
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CONDA = "/opt/conda/bin/python3.9"
files_per_class = int(sys.argv[1]) if len(sys.argv) > 1 else 48
events_per_file = int(sys.argv[2]) if len(sys.argv) > 2 else 85
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
workers = int(sys.argv[4]) if len(sys.argv) > 4 else 8
T = 256
CLASSES = ["Gamma", "Electron", "Positron"]

WRITER = r'''
import sys, os, numpy as np, h5py
root, files, events, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dt = np.dtype([("evt", "<i8"), ("t", "<f8"), ("dt", "<f4"), ("z", "<f4"), ("E", "<f4"), ("PSD", "<f4"), ("PE", "<f4", (2,)),
               ("coord", "<i4", (4,)), ("waveform", "<f4", (2,)), ("EZ", "<f4", (2,)), ("PID", "<i4")])
rows = 0
for ci, name in enumerate(["Gamma", "Electron", "Positron"]):
    os.makedirs(os.path.join(root, name), exist_ok=True)
    tau = (12.0, 20.0, 32.0)[ci]
    for fi in range(files):
        rng = np.random.default_rng(1000 * ci + fi)
        coords, wf = [], []
        for e in range(events):
            for _ in range(int(rng.integers(1, 4))):
                x, y, t0 = int(rng.integers(0, 14)), int(rng.integers(0, 11)), int(rng.integers(0, T // 4))
                n = int(rng.integers(T // 8, T // 2))
                c = np.empty((n, 4), np.int32); c[:, 0] = x; c[:, 1] = y; c[:, 2] = np.arange(t0, t0 + n); c[:, 3] = e
                amp = rng.uniform(2000, 12000)
                s = amp * np.exp(-np.arange(n) / tau)
                w = np.stack([s * rng.uniform(0.7, 1.0), s * rng.uniform(0.7, 1.0)], 1) + rng.normal(0, 30, (n, 2))
                coords.append(c); wf.append(w.astype(np.float32))
        coords = np.concatenate(coords); wf = np.concatenate(wf)
        rec = np.zeros(len(coords), dt)
        rec["coord"] = coords; rec["evt"] = coords[:, 3]; rec["waveform"] = wf
        with h5py.File(os.path.join(root, name, "%03d_Waveform3DPairSim.h5" % fi), "w") as f:
            d = f.create_dataset("Waveform3DPairs", data=rec, chunks=(min(4096, len(rec)),), compression="gzip", compression_opts=4)
            d.attrs.create("nevents", np.array([events]))
        rows += len(coords)
print(rows)
'''


def main():
    import torch
    from waveformml_amd.psd.config import DictionaryUtility
    from waveformml_amd.psd.lit import LitPSD
    from waveformml_amd.psd.PSDDataModule import PSDDataModule
    from waveformml_amd.psd.trainer import Trainer
    from waveformml_amd.spconv import ops
    tmp = tempfile.mkdtemp(prefix="wfs_soak_", dir="/tmp")
    out = {"files_per_class": files_per_class, "events_per_file": events_per_file, "classes": len(CLASSES), "T": T}
    try:
        t0 = time.perf_counter()
        rows = int(subprocess.run([CONDA, "-c", WRITER, tmp, str(files_per_class), str(events_per_file), str(T)],
                                  check=True, capture_output=True, text=True).stdout.strip())
        size = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(tmp) for f in fs)
        out["files"] = {"rows": rows, "bytes_on_disk": size, "write_seconds": round(time.perf_counter() - t0, 1)}
        print("wrote %d files, %d rows, %.1f MB" % (files_per_class * len(CLASSES), rows, size / 1e6), flush=True)
        cfg = json.load(open(os.path.join(ROOT, "config", "psd_c2_3d.json")))
        n_events = files_per_class * events_per_file
        cfg["dataset_config"] = {"imports": ["waveformml_amd.psd.PulseDataset"], "dataset_class": "PulseDataset.PulseDataset3D",
                                 "base_path": tmp, "paths": CLASSES, "n_train": n_events, "n_validate": 0, "n_test": 0,
                                 "dataloader_params": {"batch_size": len(CLASSES), "num_workers": workers,
                                                       "pin_memory": True}}
        cfg["optimize_config"].update(lr=0.004, optimizer_params={"momentum": 0.9, "nesterov": True})

        def module_and_loader(nw, pack=True):
            c = copy.deepcopy(cfg)
            c["dataset_config"]["pack_batches"] = pack
            c["dataset_config"]["dataloader_params"]["num_workers"] = nw
            if nw > 0:
                c["dataset_config"]["dataloader_params"].update(persistent_workers=True, prefetch_factor=4)
            conf = DictionaryUtility.to_object(c)
            dm = PSDDataModule(conf, "cpu")
            dm.setup("fit")
            return conf, dm.train_dataloader()

        # ---- loader alone: the budget one rank's workers deliver
        out["loader"] = []
        for nw in sorted({1, 4, workers, 2 * workers}):
            _, loader = module_and_loader(nw)
            for _ in loader:                       # first pass: worker start-up, page cache
                pass
            t0 = time.perf_counter()
            ev = vox = 0
            for (c, f), y in loader:
                ev += int(y.shape[0])
                vox += int(c.shape[0])
            dt = time.perf_counter() - t0
            out["loader"].append({"workers": nw, "events_per_s": round(ev / dt), "voxels_per_s": round(vox / dt),
                                  "batches": len(loader), "seconds": round(dt, 2), "handover": "one buffer per batch"})
            print("loader", out["loader"][-1], flush=True)
            del loader
        _, loader = module_and_loader(workers, pack=False)
        for _ in loader:
            pass
        t0 = time.perf_counter()
        ev = sum(int(y.shape[0]) for (c, f), y in loader)
        out["loader"].append({"workers": workers, "events_per_s": round(ev / (time.perf_counter() - t0)),
                              "handover": "one segment per tensor (plain DataLoader)"})
        print("loader", out["loader"][-1], flush=True)
        del loader
        if not torch.cuda.is_available():
            print(json.dumps(out))
            return
        # ---- training from the files
        ops.ASSUME_VALID_UNIQUE_INDICES = True
        ops.PREFETCH_RULEBOOKS = True
        torch.manual_seed(0)
        conf, loader = module_and_loader(workers)
        mod = LitPSD(conf)
        tr = Trainer(max_epochs=epochs, device="cuda:0", feature_dtype=torch.bfloat16, capture=True, check_every=25)
        t0 = time.perf_counter()
        hist = tr.fit(mod, loader)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = epochs * len(loader)
        out["train"] = {"epochs": epochs, "steps": steps, "seconds": round(dt, 2),
                        "events_per_s_incl_loading_and_capture": round(steps * len(CLASSES) * events_per_file / dt),
                        "events_per_s_per_epoch": [round(h["steps"] * len(CLASSES) * events_per_file / h["train_seconds"])
                                                   for h in hist],
                        "n_cap": tr._graph.n_cap if tr._graph is not None else None, "eager_fallbacks": tr.eager_fallbacks,
                        "history": hist}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

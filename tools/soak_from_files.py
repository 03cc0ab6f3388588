"""Soak run of the captured training loop FROM FILES, end to end through the reference-shaped data path:
HDF5 files on disk (written here with h5py under the image's conda interpreter, gzip-chunked compound tables as the
reference's converters write them) -> libwfh5 reader inside DataLoader workers -> PSDDataModule / PulseDataset3D / collate ->
DevicePrefetcher -> Trainer(capture=True): bf16 rows, HIP-graph step.  Prints one JSON object with
  * loader: events/s of the DataLoader alone for several worker counts (the per-rank loader budget),
  * train: steps, events/s including loading, eager fallbacks, overflow checks, loss per epoch.
The files hold the BENCH'S events (psd/synthetic.generate, ~330 voxels per event; one class per directory), so loader
and training rates are comparable with bench.py's in events/s AND voxels/s.

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
# one class directory per event class, every event drawn by psd/synthetic.generate -- THE BENCH'S EVENTS (1 + Poisson(2)
# hit segments, ~110 active samples each: ~330 voxels per event), not lighter stand-ins (round 2's files had 159)
import importlib.util, sys, os, numpy as np, h5py
root, files, events, T, synth = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
spec = importlib.util.spec_from_file_location("wfs_synthetic", synth)
synthetic = importlib.util.module_from_spec(spec)
spec.loader.exec_module(synthetic)
dt = np.dtype([("evt", "<i8"), ("t", "<f8"), ("dt", "<f4"), ("z", "<f4"), ("E", "<f4"), ("PSD", "<f4"), ("PE", "<f4", (2,)),
               ("coord", "<i4", (4,)), ("waveform", "<f4", (2,)), ("EZ", "<f4", (2,)), ("PID", "<i4")])
rows = 0
for ci, name in enumerate(["Gamma", "Electron", "Positron"]):
    os.makedirs(os.path.join(root, name), exist_ok=True)
    for fi in range(files):
        coords, wf, _y = synthetic.generate(events, T, 3, seed=100000 * ci + fi, label=ci)
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
    if os.environ.get("WFS_SWITCH_INTERVAL"):           # experiment: GIL hand-over between the trainer and stager threads
        sys.setswitchinterval(float(os.environ["WFS_SWITCH_INTERVAL"]))
    from waveformml_amd.psd.config import DictionaryUtility
    from waveformml_amd.psd.lit import LitPSD
    from waveformml_amd.psd.PSDDataModule import PSDDataModule
    from waveformml_amd.psd.trainer import Trainer
    from waveformml_amd.spconv import ops
    tmp = tempfile.mkdtemp(prefix="wfs_soak_", dir="/tmp")
    out = {"files_per_class": files_per_class, "events_per_file": events_per_file, "classes": len(CLASSES), "T": T,
           "inflate_threads_per_read": int(os.environ.get("WFH5_THREADS", "4")),
           "handover": "shared page-locked ring" if os.environ.get("WFS_LOADER_RING", "1") != "0" else "one shared-memory segment per message + pin_memory thread"}
    try:
        t0 = time.perf_counter()
        rows = int(subprocess.run([CONDA, "-c", WRITER, tmp, str(files_per_class), str(events_per_file), str(T),
                                   os.path.join(ROOT, "waveformml_amd", "psd", "synthetic.py")],
                                  check=True, capture_output=True, text=True).stdout.strip())
        size = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(tmp) for f in fs)
        out["files"] = {"rows": rows, "bytes_on_disk": size, "write_seconds": round(time.perf_counter() - t0, 1),
                        "voxels_per_event": round(rows / (len(CLASSES) * files_per_class * events_per_file), 1)}
        print("wrote %d files, %d rows, %.1f MB" % (files_per_class * len(CLASSES), rows, size / 1e6), flush=True)
        cfg = json.load(open(os.path.join(ROOT, "config", "psd_c2_3d.json")))
        n_events = files_per_class * events_per_file
        cfg["dataset_config"] = {"imports": ["waveformml_amd.psd.PulseDataset"], "dataset_class": "PulseDataset.PulseDataset3D",
                                 "base_path": tmp, "paths": CLASSES, "n_train": n_events, "n_validate": 0, "n_test": 0,
                                 "dataloader_params": {"batch_size": len(CLASSES), "num_workers": workers,
                                                       "pin_memory": os.environ.get("WFS_SOAK_PIN", "1") != "0"}}
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

        # ---- loader alone: the budget one rank's workers deliver, per hand-over group (batches per worker message)
        out["loader"] = []
        for group in (1, 4):
            os.environ["WFS_LOADER_GROUP"] = str(group)
            for nw in ([workers] if os.environ.get("WFS_SOAK_QUICK") else sorted({1, 4, workers, 2 * workers})):
                _, loader = module_and_loader(nw)
                for _ in loader:                       # first pass: worker start-up, page cache
                    pass
                t0 = time.perf_counter()
                ev = vox = passes = 0
                while passes < 3 or time.perf_counter() - t0 < 2.0:      # at least three passes and two seconds
                    for (c, f), y in loader:
                        ev += int(y.shape[0])
                        vox += int(c.shape[0])
                    passes += 1
                dt = time.perf_counter() - t0
                out["loader"].append({"workers": nw, "batches_per_message": group, "events_per_s": round(ev / dt),
                                      "voxels_per_s": round(vox / dt), "batches": passes * len(loader),
                                      "seconds": round(dt, 2)})
                print("loader", out["loader"][-1], flush=True)
                del loader
        os.environ["WFS_LOADER_GROUP"] = os.environ.get("WFS_SOAK_TRAIN_GROUP", "4")
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
        tr = Trainer(max_epochs=epochs, device="cuda:0", feature_dtype=torch.bfloat16, capture=True,
                     check_every=int(os.environ.get("WFS_SOAK_CHECK_EVERY", "100")))
        t0 = time.perf_counter()
        hist = tr.fit(mod, loader)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = epochs * len(loader)
        out["train"] = {"epochs": epochs, "steps": steps, "seconds": round(dt, 2),
                        "events_per_s_incl_loading_and_capture": round(steps * len(CLASSES) * events_per_file / dt),
                        "events_per_s_per_epoch": [round(h["steps"] * len(CLASSES) * events_per_file / h["train_seconds"])
                                                   for h in hist],
                        "voxels_per_s_per_epoch": [round(rows / h["train_seconds"]) for h in hist],
                        "n_cap": tr.last_capacity or None, "eager_fallbacks": tr.eager_fallbacks,
                        "recaptures": tr.recaptures, "loader_group": int(os.environ.get("WFS_LOADER_GROUP", "1")),
                        "history": hist}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

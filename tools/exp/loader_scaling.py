"""Loader-only scaling on the GPU box's host share: events/s of the from-files DataLoader for (workers, libwfh5 inflate
threads, batches per message, pin_memory).  usage: python tools/exp/loader_scaling.py [files_per_class]"""
import copy, json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
src = open(os.path.join(ROOT, "tools", "soak_from_files.py")).read()
WRITER = src[src.index("WRITER = r'''") + len("WRITER = r'''"):src.index("'''\n\n\ndef main")]
files = int(sys.argv[1]) if len(sys.argv) > 1 else 300
tmp = tempfile.mkdtemp(prefix="wfs_lscale_", dir="/tmp")
try:
    subprocess.run(["/opt/conda/bin/python3.9", "-c", WRITER, tmp, str(files), "85", "256",
                    os.path.join(ROOT, "waveformml_amd", "psd", "synthetic.py")], check=True, capture_output=True)
    import torch
    from waveformml_amd.psd.config import DictionaryUtility
    from waveformml_amd.psd.PSDDataModule import PSDDataModule
    cfg = json.load(open(os.path.join(ROOT, "config", "psd_c2_3d.json")))
    CLASSES = ["Gamma", "Electron", "Positron"]
    for nw, nt, group, pin in [(4, 4, 4, True), (8, 2, 4, True), (16, 1, 4, True), (12, 1, 4, True), (8, 1, 4, True),
                               (16, 1, 4, False), (16, 1, 8, True), (16, 1, 1, True), (14, 1, 4, True)]:
        os.environ["WFS_LOADER_GROUP"] = str(group)
        os.environ["WFH5_THREADS"] = str(nt)
        c = copy.deepcopy(cfg)
        c["dataset_config"] = {"imports": ["waveformml_amd.psd.PulseDataset"], "dataset_class": "PulseDataset.PulseDataset3D",
                               "base_path": tmp, "paths": CLASSES, "n_train": files * 85, "n_validate": 0, "n_test": 0,
                               "pack_batches": True,
                               "dataloader_params": {"batch_size": 3, "num_workers": nw, "pin_memory": pin,
                                                     "persistent_workers": True, "prefetch_factor": 4}}
        dm = PSDDataModule(DictionaryUtility.to_object(c), "cpu")
        dm.setup("fit")
        dl = dm.train_dataloader()
        for _ in dl:
            pass
        t0 = time.perf_counter()
        ev = passes = 0
        while passes < 3 or time.perf_counter() - t0 < 2.0:
            for (co, f), y in dl:
                ev += int(y.shape[0])
            passes += 1
        dt = time.perf_counter() - t0
        print("workers %2d  inflate threads %d  group %d  pin %d : %7.0f events/s" % (nw, nt, group, pin, ev / dt), flush=True)
        del dl, dm
finally:
    shutil.rmtree(tmp, ignore_errors=True)

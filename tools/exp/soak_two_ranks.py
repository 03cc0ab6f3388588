"""Rehearsal of the TWO-RANK training loop from files on ONE card (gloo: RCCL refuses two ranks on one device): each rank
reads its share of the files (DistributedSampler), DataLoader workers -> shared page-locked ring -> DevicePrefetcher ->
Trainer(capture=True) with the batch shapes agreed ahead of time (graph.ShapeAgreement) -> captured step, gradients
exchanged after the replay (gloo).  Checks that both ranks finish every epoch with bit-identical parameters and reports
fallbacks / re-captures.  Not a throughput number: two ranks share the card and gloo moves the gradients through the host.

usage: python tools/exp/soak_two_ranks.py [files_per_class] [events_per_file] [epochs] [workers] [agree_block]"""
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(tmp, files_per_class, events_per_file, epochs, workers, agree_block, out):
    import copy
    import faulthandler
    import torch
    import torch.distributed as dist
    if os.environ.get("WFS_WATCHDOG"):          # every thread's stack on stderr, then exit, after N seconds
        faulthandler.dump_traceback_later(int(os.environ["WFS_WATCHDOG"]), exit=True)
    torch.cuda.set_device(0)
    torch.cuda.set_stream(torch.cuda.Stream())
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    from waveformml_amd.psd.config import DictionaryUtility
    from waveformml_amd.psd.lit import LitPSD
    from waveformml_amd.psd.PSDDataModule import PSDDataModule
    from waveformml_amd.psd.trainer import Trainer
    from waveformml_amd.spconv import ops
    classes = ["Gamma", "Electron", "Positron"]
    cfg = json.load(open(os.path.join(ROOT, "config", "psd_c2_3d.json")))
    cfg["dataset_config"] = {"imports": ["waveformml_amd.psd.PulseDataset"], "dataset_class": "PulseDataset.PulseDataset3D",
                             "base_path": tmp, "paths": classes,
                             "n_train": (files_per_class - 2 * int(os.environ.get("WFS_SOAK_VALIDATE", "0"))) * events_per_file,
                             "n_validate": int(os.environ.get("WFS_SOAK_VALIDATE", "0")) * events_per_file,
                             "n_test": int(os.environ.get("WFS_SOAK_VALIDATE", "0")) * events_per_file,
                             "pack_batches": True,
                             "dataloader_params": {"batch_size": len(classes), "num_workers": workers, "pin_memory": True,
                                                   "persistent_workers": True, "prefetch_factor": 4}}
    cfg["optimize_config"].update(lr=0.004, optimizer_params={"momentum": 0.9, "nesterov": True})
    os.environ.setdefault("WFS_LOADER_GROUP", "4")
    ops.ASSUME_VALID_UNIQUE_INDICES = True
    ops.PREFETCH_RULEBOOKS = True
    torch.manual_seed(0)
    conf = DictionaryUtility.to_object(copy.deepcopy(cfg))
    dm = PSDDataModule(conf, "cpu")
    dm.setup("fit")
    loader = dm.train_dataloader()
    mod = LitPSD(conf)
    root = os.path.join(tmp, "ckpt") if os.environ.get("WFS_SOAK_RESUME") == "1" else None
    tr = Trainer(max_epochs=epochs, device="cuda:0", feature_dtype=torch.bfloat16, capture=True, check_every=25,
                 agree_block=agree_block, default_root_dir=root)
    val = dm.val_dataloader() if os.environ.get("WFS_SOAK_VALIDATE", "0") != "0" else None
    hist = tr.fit(mod, loader, val)
    torch.cuda.synchronize()
    if root is not None:
        # rank 0 wrote the checkpoints (best validation loss so far); BOTH ranks resume a fresh module from the last one
        # and train one more epoch: the replicas must still be bit-identical afterwards
        import glob
        dist.barrier()
        last = sorted(glob.glob(os.path.join(root, "*.ckpt")), key=os.path.getmtime)[-1]
        torch.manual_seed(1)
        mod = LitPSD(conf)
        tr = Trainer(max_epochs=epochs + 1, device="cuda:0", feature_dtype=torch.bfloat16, capture=True, check_every=25,
                     agree_block=agree_block, resume_from_checkpoint=last)
        hist = hist + tr.fit(mod, loader, val)
        torch.cuda.synchronize()
    params = torch.cat([p.detach().float().reshape(-1).cpu() for p in mod.model.parameters()])
    torch.save({"params": params, "steps": [h["steps"] for h in hist], "loss": [h["train_loss"] for h in hist],
                "seconds": [h["train_seconds"] for h in hist], "eager_fallbacks": tr.eager_fallbacks,
                "val_loss": [float(h.get("val_loss", float("nan"))) for h in hist],
                "recaptures": tr.recaptures, "n_cap": tr.last_capacity}, out + ".rank%s" % os.environ["RANK"])
    dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    if argv and argv[0] == "--child":
        child(argv[1], int(argv[2]), int(argv[3]), int(argv[4]), int(argv[5]), int(argv[6]), argv[7])
        return
    fpc = int(argv[0]) if len(argv) > 0 else 60
    epf = int(argv[1]) if len(argv) > 1 else 85
    epochs = int(argv[2]) if len(argv) > 2 else 3
    workers = int(argv[3]) if len(argv) > 3 else 4
    block = int(argv[4]) if len(argv) > 4 else 4
    saved, sys.argv = sys.argv, [sys.argv[0]]
    import tools.soak_from_files as sff             # the file writer (h5py under the image's conda interpreter)
    sys.argv = saved
    import torch
    tmp = tempfile.mkdtemp(prefix="wfs_soak2_", dir="/tmp")
    try:
        rows = int(subprocess.run([sff.CONDA, "-c", sff.WRITER, tmp, str(fpc), str(epf), str(sff.T),
                                   os.path.join(ROOT, "waveformml_amd", "psd", "synthetic.py")],
                                  check=True, capture_output=True, text=True).stdout.strip())
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        out = os.path.join(tmp, "res")
        world = int(os.environ.get("WFS_SOAK_WORLD", "2"))          # at most 6 processes may use the card together
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", tmp, str(fpc), str(epf),
                                           str(epochs), str(workers), str(block), out], env=env, cwd=ROOT))
        codes = [p.wait(timeout=int(os.environ.get("WFS_WATCHDOG", "880")) + 20) for p in procs]
        assert codes == [0] * world, codes
        res = [torch.load(out + ".rank%d" % r, weights_only=True) for r in range(world)]
        r0, r1 = res[0], res[-1]
        same = all(bool(torch.equal(r0["params"], r["params"])) for r in res[1:])
        print(json.dumps({"files_per_class": fpc, "events_per_file": epf, "rows": rows, "epochs": epochs, "workers_per_rank": workers,
                          "agree_block": block, "ranks": world, "replicas_bit_identical": same, "steps_per_epoch": r0["steps"],
                          "loss_per_epoch": [r0["loss"], r1["loss"]], "val_loss_per_epoch": [r0["val_loss"], r1["val_loss"]], "seconds_per_epoch": [r0["seconds"], r1["seconds"]],
                          "eager_fallbacks": [r0["eager_fallbacks"], r1["eager_fallbacks"]],
                          "recaptures": [r0["recaptures"], r1["recaptures"]], "n_cap": [r0["n_cap"], r1["n_cap"]]}))
        assert same and r0["steps"] == r1["steps"]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()

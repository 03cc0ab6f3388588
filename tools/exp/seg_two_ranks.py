"""Two-rank rehearsal (one card, gloo) of the PER-ROW captured step: LitSegClassifier (one label per active segment, the
step pads rows and labels to its capacity), every rank with its own batches of other row counts, batch shapes agreed ahead
of time (graph.ShapeAgreement).  Replicas must end bit-identical.  usage: python tools/exp/seg_two_ranks.py"""
import copy, os, socket, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(out):
    import numpy as np, torch, torch.distributed as dist
    rank = int(os.environ["RANK"])
    torch.cuda.set_device(0)
    torch.cuda.set_stream(torch.cuda.Stream())
    dist.init_process_group("gloo", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
    from test_gpu_segment_callers import IONI, _swap_imports, segment_rows
    from waveformml_amd.psd.config import load_config
    from waveformml_amd.psd.litseg import LitSegClassifier
    from waveformml_amd.psd.trainer import Trainer
    cfg = _swap_imports(IONI, "waveformml_amd.spconv")
    cfg["optimize_config"].update(lr=0.01, optimizer_params={"momentum": 0.9, "nesterov": True})
    rng = np.random.default_rng(100 + rank)
    batches = []
    for s in range(8):
        n_events = (30, 44, 52, 36)[(s + rank) % 4] + (40 if (s == 5 and rank == 1) else 0)     # one batch far larger on rank 1
        rows, c, f = segment_rows(rng, n_events, 5, 130)
        batches.append(([c, f], torch.from_numpy(rng.integers(0, 5, len(rows)))))
    torch.manual_seed(7)
    mod = LitSegClassifier(load_config(copy.deepcopy(cfg)))
    tr = Trainer(max_epochs=2, device="cuda:0", capture=True, check_every=3, agree_block=int(os.environ.get("AGREE", "4")))
    hist = tr.fit(mod, batches)
    torch.cuda.synchronize()
    torch.save({"params": torch.cat([p.detach().reshape(-1).cpu() for p in mod.model.parameters()]),
                "rows": [int(b[0][0].shape[0]) for b in batches], "fallbacks": tr.eager_fallbacks,
                "recaptures": tr.recaptures, "n_cap": int(tr.last_capacity), "loss": [h["train_loss"] for h in hist]},
               out + ".rank%d" % rank)
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    import tempfile, torch
    tmp = tempfile.mkdtemp(prefix="wfs_seg2_", dir="/tmp")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", os.path.join(tmp, "res")],
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                                       HSA_ENABLE_IPC_MODE_LEGACY="0"), cwd=ROOT) for r in range(2)]
    codes = [p.wait(timeout=400) for p in procs]
    assert codes == [0, 0], codes
    r0, r1 = (torch.load(os.path.join(tmp, "res.rank%d" % r), weights_only=True) for r in range(2))
    print({"rows": [r0["rows"], r1["rows"]], "n_cap": [r0["n_cap"], r1["n_cap"]], "fallbacks": [r0["fallbacks"], r1["fallbacks"]],
           "recaptures": [r0["recaptures"], r1["recaptures"]], "loss": [r0["loss"], r1["loss"]],
           "replicas_bit_identical": bool(torch.equal(r0["params"], r1["params"]))})
    assert torch.equal(r0["params"], r1["params"])

"""Multi-process CPU tests (gloo, world_size 2) of the data-parallel gradient exchange
(waveformml_amd/psd/ddp.py): flat parameter / flat gradient buffers, bucketed asynchronous all-reduce
launched from gradient hooks, averaging, identical replicas after the optimizer step.
The GPU path differs only in the backend name ("nccl" == RCCL) -- see bench.py."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _net():
    torch.manual_seed(7)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.BatchNorm1d(16), torch.nn.ReLU(),
                               torch.nn.Linear(16, 16, bias=False), torch.nn.ReLU(), torch.nn.Linear(16, 3))


def _data(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(12, 6, generator=g), torch.randint(0, 3, (12,), generator=g)


def _worker(rank, world, port, n_buckets, out):
    from waveformml_amd.psd.ddp import FlatGradAllReducer, broadcast_parameters
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        net = _net()
        if rank != 0:                       # replicas start different; the broadcast must fix that
            with torch.no_grad():
                for p in net.parameters():
                    p.add_(1.0)
        broadcast_parameters(net)
        red = FlatGradAllReducer(net.parameters(), n_buckets=n_buckets)
        opt = torch.optim.SGD(red.optimizer_parameters(), lr=0.1, momentum=0.9, nesterov=True)
        crit = torch.nn.CrossEntropyLoss()
        x, y = _data(rank)
        for _ in range(3):
            red.reset()
            crit(net(x), y).backward()
            red.finish()
            opt.step()
        out[rank] = (red.flat_grad.clone(), [p.detach().clone() for p in net.parameters()],
                     [p.data_ptr() for p in net.parameters()], red.flat_param.data_ptr(),
                     [(s, e, list(i)) for s, e, i in red.buckets])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_buckets", [1, 3])
def test_two_rank_gloo_matches_single_process_average(n_buckets):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_buckets, out), nprocs=world, join=True)
    g0, p0, ptr0, fp0, buckets = out[0]
    g1, p1, _, _, _ = out[1]
    assert torch.equal(g0, g1)
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)                     # replicas stay bit-identical
    # parameters are views of the flat parameter, laid out in reverse order
    assert min(ptr0) == fp0 and len(buckets) == n_buckets
    assert buckets[0][0] == 0 and buckets[-1][1] == g0.numel()
    assert sorted(i for _, _, idxs in buckets for i in idxs) == list(range(len(p0)))
    # single-process reference: per-rank BatchNorm statistics (no SyncBN), gradients averaged over ranks
    net = _net()
    nets = [net, _net()]
    nets[1].load_state_dict(net.state_dict())
    opts = [torch.optim.SGD(n.parameters(), lr=0.1, momentum=0.9, nesterov=True) for n in nets]
    crit = torch.nn.CrossEntropyLoss()
    for _ in range(3):
        for r in range(2):
            opts[r].zero_grad()
            x, y = _data(r)
            crit(nets[r](x), y).backward()
        for pa, pb in zip(nets[0].parameters(), nets[1].parameters()):
            avg = (pa.grad + pb.grad) / 2
            pa.grad, pb.grad = avg.clone(), avg.clone()
        for o in opts:
            o.step()
    for a, b in zip(p0, nets[0].parameters()):
        torch.testing.assert_close(a, b.detach(), rtol=1e-5, atol=1e-6)


def test_single_process_reducer_packs_gradients_and_flat_optimizer_step():
    from waveformml_amd.psd.ddp import FlatGradAllReducer
    net, ref = _net(), _net()
    red = FlatGradAllReducer(net.parameters(), n_buckets=2, world_size=1)
    assert red.world == 1 and len(red.optimizer_parameters()) == 1
    opt = torch.optim.SGD(red.optimizer_parameters(), lr=0.05, momentum=0.9)
    opt_ref = torch.optim.SGD(ref.parameters(), lr=0.05, momentum=0.9)
    x, y = _data(0)
    crit = torch.nn.CrossEntropyLoss()
    for _ in range(2):
        red.reset()
        crit(net(x), y).backward()
        red.finish()
        opt.step()
        opt_ref.zero_grad()
        crit(ref(x), y).backward()
        opt_ref.step()
    for i, (a, b) in enumerate(zip(net.parameters(), ref.parameters())):
        o, n = red.slices[i]
        torch.testing.assert_close(red.flat_grad[o:o + n].view_as(b), b.grad)
        torch.testing.assert_close(a.detach(), b.detach())
    assert set(net.state_dict()) == set(ref.state_dict())        # checkpoints keep the module's own names


def _sampler_worker(rank, world, port, out):
    from waveformml_amd.psd import data
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ds = data.SyntheticPulseDataset(8, 3, 16, n_type=3, layout="3d", seed=5)
        seen = {}
        for shuffle in (False, True):
            loader = data.make_loader(ds, 2, shuffle=shuffle, pin_memory=False)
            assert isinstance(loader.sampler, torch.utils.data.distributed.DistributedSampler)
            per_epoch = []
            for epoch in range(2):
                loader.sampler.set_epoch(epoch)
                per_epoch.append(list(iter(loader.sampler)))
                n_events = sum(int(y.shape[0]) for (_cf, y) in loader)
                assert n_events == 4 * 3                     # 8 items / 2 ranks, 3 events each
            seen[shuffle] = per_epoch
        out[rank] = seen
    finally:
        dist.destroy_process_group()


def test_two_ranks_read_disjoint_items():
    """Each rank gets its own 1/N share of the items (an item = one file's event range), as under the reference's
    Lightning DDP, which swaps DistributedSamplers into the loaders (src/utils/util.py:228-239); a shuffled loader is
    re-shuffled per epoch by set_epoch -- identically on all ranks, so the shares stay disjoint."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_sampler_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for shuffle in (False, True):
        for epoch in range(2):
            a, b = out[0][shuffle][epoch], out[1][shuffle][epoch]
            assert set(a).isdisjoint(b) and sorted(a + b) == list(range(8)), (shuffle, epoch, a, b)
    assert out[0][True][0] != out[0][True][1]                    # another permutation per epoch
    assert out[0][False][0] == out[0][False][1] == [0, 2, 4, 6]


def _agree_worker(rank, world, port, out):
    from waveformml_amd.psd.graph import _agree_max
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out[rank] = (_agree_max(1 if rank == 1 else 0, None), _agree_max(0, None))
    finally:
        dist.destroy_process_group()


def test_ranks_agree_on_a_one_sided_capture_failure():
    """psd/graph.py: whether the in-graph exchange could be captured is decided by ALL ranks (MAX over a host-side flag
    on a gloo side group) -- a failure on one rank only must send every rank down the exchange-after-replay path."""
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_agree_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out[0] == (1, 0) and out[1] == (1, 0)


def test_bench_counts_gpus_from_sysfs_without_the_runtime(tmp_path, monkeypatch):
    """bench.py's launcher parent must not touch the GPU runtime before it starts its ranks: devices are counted from
    the KFD topology (a node with SIMDs is a GPU; CPU nodes have none), cut by *_VISIBLE_DEVICES."""
    import importlib
    bench = importlib.import_module("bench")
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):
        d = tmp_path / "nodes" / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n" % (64 if simd == 0 else 0, simd))
    monkeypatch.setenv("WFS_KFD_TOPOLOGY", str(tmp_path))
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    assert bench.visible_gpus() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpus() == 2
    monkeypatch.setenv("WFS_KFD_TOPOLOGY", str(tmp_path / "nowhere"))
    assert bench.visible_gpus() is None


def _agreement_worker(rank, world, port, block, out):
    from waveformml_amd.psd.graph import ShapeAgreement
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # five batches; rank 1's third has more rows, rank 0's fifth one label fewer than rows
        rows = [[100, 90, 110, 95, 80], [101, 88, 230, 95, 80]][rank]
        labels = [[16, 16, 16, 16, 79], [16, 16, 16, 16, 80]][rank]
        ag = ShapeAgreement(None, block)
        got = []
        # the prefetcher's pattern: `depth` batches staged ahead, flush when the loader is exhausted
        depth, staged = block + 2, 0
        for i in range(5):
            while staged < 5 and staged < i + depth:
                ag.stage(rows[staged], labels[staged])
                staged += 1
                if staged == 5:
                    ag.flush()
            got.append(ag.next())
        assert not ag.local and not ag.ready
        out[rank] = got
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("block", [1, 2, 4, 8])
def test_shape_agreement_gives_every_rank_the_same_counts(block):
    """psd/graph.ShapeAgreement (the multi-rank Trainer's replay-or-ordinary-step decision): rows MAX, labels MIN / MAX
    and the rows != labels flag over the ranks, in staging order, whole and partial blocks."""
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_agreement_worker, args=(world, _free_port(), block, out), nprocs=world, join=True)
    want = [(101, 16, 16, True), (90, 16, 16, True), (230, 16, 16, True), (95, 16, 16, True), (80, 79, 80, True)]
    assert out[0] == out[1] == want

"""GPU parity of the event-local SubM rulebook build (csrc/evrulebook.hip, include/wfsparse.h "Event-local rulebook
build") through the C ABI: event offsets against numpy, nbr_out BIT-EXACT against the CPU oracle's spconv order
(oracle/spconv_ref.c) and against the chip-wide build, the failure flags (batch column not sorted,
duplicate coordinates, more active cells than the LDS tables hold), and a captured training step that must be
bit-identical with the build on and off (same tables -> same arithmetic).
"""
import ctypes

import numpy as np
import pytest
import torch

from helpers import rand_coords

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _lib():
    from waveformml_amd import _lib
    return _lib, _lib.load()


def _offsets(idx_t, B, n_dev=None):
    from waveformml_amd.spconv import ops
    return ops.event_offsets(idx_t, B, n_dev)


def _sorted_by_event(idx):
    return np.ascontiguousarray(idx[np.argsort(idx[:, 0], kind="stable")])


def _ev_subm(idx_t, B, shape, ksize, dilation=1, n_dev=None):
    L, lib = _lib()
    ndim = len(shape)
    g = L.make_geometry(ndim, B, list(shape), [ksize] * ndim if np.isscalar(ksize) else list(ksize), [1] * ndim, [0] * ndim,
                        [dilation] * ndim, True)
    assert lib.wfs_event_rulebook_ok(ctypes.byref(g))
    N = idx_t.shape[0]
    ev = _offsets(idx_t, B, n_dev)
    nbr = torch.full((int(g.K), N), -7, dtype=torch.int32, device=DEV)
    flags = torch.zeros((int(lib.wfs_event_rulebook_flag_ints(B)),), dtype=torch.int32, device=DEV)   # sticky: the caller clears
    L.check(lib.wfs_event_rulebook_subm(ctypes.byref(g), L.ptr(idx_t), N, L.ptr(n_dev), L.ptr(ev), L.ptr(nbr), None,
                                        L.ptr(flags), L.stream_ptr()))
    torch.cuda.synchronize()
    nb = flags.numel() // 3
    f = flags.cpu().numpy()
    return nbr, (int(f[:nb].any()), int(f[nb:2 * nb].any()), int(f[2 * nb:].any())), ev


def _oracle_nbr_out(idx, B, shape, ksize, dilation=1):
    """nbr_out [K, N] from the oracle's indice_pairs (spconv's encoding): column k lists (input row, output row)."""
    from oracle import ref
    _out, pairs, num = ref.get_indice_pairs(idx, B, shape, ksize, 1, 0, dilation, 0, True)
    K, N = pairs.shape[1], idx.shape[0]
    nbr = np.full((K, N), -1, np.int32)
    for k in range(K):
        n = int(num[k])
        nbr[k, pairs[0, k, :n]] = pairs[1, k, :n]
    return nbr


def test_event_offsets_against_numpy_and_flags():
    rng = np.random.default_rng(5)
    B = 37
    b = np.sort(rng.integers(0, B, size=5000)).astype(np.int32)
    b = b[(b != 3) & (b != 36)]                      # empty events in the middle and at the end
    idx = np.stack([b, rng.integers(0, 9, len(b)), rng.integers(0, 64, len(b))], 1).astype(np.int32)
    t = torch.from_numpy(idx).to(DEV)
    ev = _offsets(t, B).cpu().numpy()
    assert np.array_equal(ev[:B + 1], np.searchsorted(b, np.arange(B + 1)))
    assert not ev[B + 1:].any()
    # device-side row count: rows beyond it do not count
    nv = torch.tensor([3000], dtype=torch.int64, device=DEV)
    ev = _offsets(t, B, nv).cpu().numpy()
    assert np.array_equal(ev[:B + 1], np.minimum(np.searchsorted(b, np.arange(B + 1)), 3000)) and not ev[B + 1:].any()
    # not grouped by event / out of range -> a flag word is set
    bad = idx.copy()
    bad[[100, 2000]] = bad[[2000, 100]]
    assert _offsets(torch.from_numpy(bad).to(DEV), B).cpu().numpy()[B + 1:].any()
    bad = idx.copy()
    bad[-1, 0] = B
    assert _offsets(torch.from_numpy(bad).to(DEV), B).cpu().numpy()[B + 1:].any()
    # no rows at all
    ev = _offsets(torch.zeros((0, 3), dtype=torch.int32, device=DEV), B).cpu().numpy()
    assert not ev.any()


@pytest.mark.parametrize("shape,ksize,dilation", [((14, 11, 64), 3, 1), ((9, 40), 3, 1), ((50,), 3, 1), ((6, 5, 33), (3, 1, 3), 1),
                                                  ((12, 30), 2, 1), ((8, 7, 20), 3, 2), ((5, 4, 3, 16), (1, 3, 3, 3), 1)])
def test_event_local_subm_bitexact_vs_oracle(shape, ksize, dilation):
    rng = np.random.default_rng(11)
    B = 9
    idx = _sorted_by_event(rand_coords(rng, B, shape, min(1500, B * int(np.prod(shape)) // 3)))
    t = torch.from_numpy(idx).to(DEV)
    nbr, flags, _ = _ev_subm(t, B, shape, ksize, dilation)
    assert flags == (0, 0, 0)
    want = _oracle_nbr_out(idx, B, shape, ksize, dilation)
    assert np.array_equal(nbr.cpu().numpy(), want)


def test_event_local_subm_at_psd_scale_equals_chipwide_build():
    from waveformml_amd.psd import synthetic
    from waveformml_amd.spconv import ops
    B, T = 256, 256
    c, _f, _y = synthetic.generate(B, T, 3, seed=77)
    idx = torch.from_numpy(np.ascontiguousarray(c[:, [3, 0, 1, 2]])).to(DEV)
    N = idx.shape[0]
    cap = N + 777                                     # device-count mode: capacity rows beyond the valid count
    pad = torch.cat([idx, torch.full((cap - N, 4), 123456, dtype=torch.int32, device=DEV)])
    nv = torch.tensor([N], dtype=torch.int64, device=DEV)
    nbr, flags, _ = _ev_subm(pad, B, (14, 11, T), 3, 1, nv)
    assert flags == (0, 0, 0)
    rb = ops.build_rulebook(idx, B, [14, 11, T], [3] * 3, [1] * 3, [0] * 3, [1] * 3, True)
    assert torch.equal(nbr[:, :N], rb.nbr_out)


def test_event_local_subm_flags():
    rng = np.random.default_rng(3)
    B, shape = 6, (14, 11, 32)
    idx = _sorted_by_event(rand_coords(rng, B, shape, 900))
    # batch column not sorted
    bad = idx.copy()
    bad[[10, 700]] = bad[[700, 10]]
    nbr, flags, _ = _ev_subm(torch.from_numpy(bad).to(DEV), B, shape, 3)
    assert flags[0] == 1
    assert bool((nbr == -1).all())          # a failed build leaves "no neighbour" everywhere, never unwritten entries
    # duplicate coordinates: detected (the caller then takes the chip-wide build, which resolves "the last row wins")
    dup = np.concatenate([idx[:50], idx[:1], idx[50:]])
    dup = _sorted_by_event(dup)
    _, flags, _ = _ev_subm(torch.from_numpy(dup).to(DEV), B, shape, 3)
    assert flags[1] == 1 and flags[0] == 0
    # an index outside the spatial shape
    out = idx.copy()
    out[5, 3] = 32
    _, flags, _ = _ev_subm(torch.from_numpy(out).to(DEV), B, shape, 3)
    assert flags[2] == 1
    # more active cells in one event than the LDS sample arrays hold (64 KiB = 32 cells of 1024 samples): flagged, not wrong
    many = np.array([(0, x, y, 0) for x in range(14) for y in range(11)], np.int32)
    nbr, flags, _ = _ev_subm(torch.from_numpy(many).to(DEV), 1, (14, 11, 1024), 3)
    assert flags[0] == 1 and bool((nbr == -1).all())
    # ... and the other events of such a batch are built as usual
    rest = idx.copy()
    rest[:, 0] += 1
    both = np.concatenate([np.concatenate([many[:, :3], np.zeros((len(many), 1), np.int32)], 1), rest])
    want = _oracle_nbr_out(rest, B + 1, (14, 11, 32), 3)
    big = both.copy()
    nbr, flags, _ = _ev_subm(torch.from_numpy(np.ascontiguousarray(big)).to(DEV), B + 1, (14, 11, 32), 3)
    assert flags == (0, 0, 0)               # 154 cells of 32 samples fit the pool: nothing to flag at this length
    assert np.array_equal(nbr.cpu().numpy()[:, len(many):], np.where(want >= 0, want + len(many), -1))


def test_captured_step_survives_an_ungrouped_batch_until_check_raises():
    """a batch that is not grouped by event, fed to a captured step: the event-local builds flag it and leave
    benign tables (no stale indices are gathered through), the step's loss is finite, and check() raises."""
    import copy
    import json
    import os
    from waveformml_amd.psd import synthetic
    from waveformml_amd.psd.config import DictionaryUtility
    from waveformml_amd.psd.ddp import FlatGradAllReducer
    from waveformml_amd.psd.graph import GraphedTrainStep
    from waveformml_amd.psd.lit import LitPSD
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "config", "psd_c2_3d.json")) as f:
        cfg = json.load(f)
    T, B = 64, 32
    cfg["system_config"]["n_samples"] = T
    cfg["net_config"]["algorithm"][-1] = [32 * 10 * 7 * 4, 3]
    c, f, y = synthetic.generate(B, T, 3, seed=5)
    good = ([torch.from_numpy(c).to(DEV), torch.from_numpy(f).to(DEV)], torch.from_numpy(y).to(DEV))
    c2, f2 = c.copy(), f.copy()
    i, j = 3, len(c) - 5                      # rows of the first and the last event change places
    c2[[i, j]], f2[[i, j]] = c2[[j, i]], f2[[j, i]]
    bad = ([torch.from_numpy(c2).to(DEV), torch.from_numpy(f2).to(DEV)], torch.from_numpy(y).to(DEV))
    torch.manual_seed(0)
    mod = LitPSD(DictionaryUtility.to_object(copy.deepcopy(cfg))).to(DEV)
    red = FlatGradAllReducer(mod.model.parameters(), world_size=1)
    mod.optimizer_parameters = red.optimizer_parameters()
    opt = mod.configure_optimizers()[0][0]
    step = GraphedTrainStep(mod, opt, red, good)
    assert np.isfinite(float(step(good)))
    step.check()
    assert np.isfinite(float(step(bad)))
    assert np.isfinite(float(step(good)))       # sticky: the good replay in between does not hide the failure
    with pytest.raises(RuntimeError, match="grouped by event"):
        step.check()
    assert np.isfinite(float(step(good)))
    step.check()


def test_captured_step_identical_with_and_without_event_local_build():
    """Same tables -> the captured step is bit-identical; check() stays silent on well-formed batches."""
    import copy
    import json
    import os
    from waveformml_amd.psd import synthetic
    from waveformml_amd.psd.config import DictionaryUtility
    from waveformml_amd.psd.ddp import FlatGradAllReducer
    from waveformml_amd.psd.graph import GraphedTrainStep
    from waveformml_amd.psd.lit import LitPSD
    from waveformml_amd.spconv import ops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "config", "psd_c2_3d.json")) as f:
        cfg = json.load(f)
    T, B = 64, 32
    cfg["system_config"]["n_samples"] = T
    cfg["net_config"]["algorithm"][-1] = [32 * 10 * 7 * 4, 3]
    batches = []
    for seed in (5, 6, 7):
        c, f, y = synthetic.generate(B, T, 3, seed=seed)
        batches.append(([torch.from_numpy(c).to(DEV), torch.from_numpy(f).to(DEV)], torch.from_numpy(y).to(DEV)))
    losses, params = [], []
    old = ops.EVENT_LOCAL, ops.EVENT_LOCAL_CONV
    try:
        for on in (False, True):
            ops.EVENT_LOCAL = ops.EVENT_LOCAL_CONV = on          # SubM (evrulebook.hip) and strided (evconv.hip) builds
            torch.manual_seed(0)
            mod = LitPSD(DictionaryUtility.to_object(copy.deepcopy(cfg))).to(DEV)
            red = FlatGradAllReducer(mod.model.parameters(), world_size=1)
            mod.optimizer_parameters = red.optimizer_parameters()
            opt = mod.configure_optimizers()[0][0]
            step = GraphedTrainStep(mod, opt, red, batches[0])
            ls = [float(step(bt)) for bt in batches]
            step.check()
            assert bool(step._event_flags) == on
            losses.append(ls)
            params.append(red.flat_param.detach().clone())
    finally:
        ops.EVENT_LOCAL, ops.EVENT_LOCAL_CONV = old
    assert losses[0] == losses[1], losses
    assert torch.equal(params[0], params[1])


def test_resume_from_checkpoint_keeps_the_momentum_under_capture(tmp_path):
    """Trainer(resume_from_checkpoint=, capture=True): the capture's calibration / warm-up steps must not be folded
    into the checkpointed momentum.  One epoch eager -> checkpoint -> one more epoch, eager and captured:
    same parameters and momentum (1e-5 of scale: a replayed step sums some reductions in another order)."""
    import copy
    import json
    import os
    from waveformml_amd.psd import synthetic
    from waveformml_amd.psd.config import DictionaryUtility
    from waveformml_amd.psd.lit import LitPSD
    from waveformml_amd.psd.trainer import Trainer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "config", "psd_c2_3d.json")) as f:
        cfg = json.load(f)
    T, B = 64, 16
    cfg["system_config"]["n_samples"] = T
    cfg["net_config"]["algorithm"][-1] = [32 * 10 * 7 * 4, 3]
    cfg["optimize_config"].pop("scheduler_class", None)
    batches = []
    for seed in (21, 22, 23):
        c, f, y = synthetic.generate(B, T, 3, seed=seed)
        batches.append(([torch.from_numpy(c), torch.from_numpy(f)], torch.from_numpy(y)))

    def module():
        torch.manual_seed(3)
        return LitPSD(DictionaryUtility.to_object(copy.deepcopy(cfg)))

    # one epoch, then a checkpoint with the optimizer's momentum in it
    t_save = Trainer(max_epochs=1, device=DEV, default_root_dir=str(tmp_path))
    m1 = module()
    t_save.fit(m1, batches, val_loader=batches)
    ck = t_save.last_checkpoint
    assert ck is not None
    res = {}
    for capture in (False, True):
        m = module()
        tr = Trainer(max_epochs=2, device=DEV, capture=capture, resume_from_checkpoint=ck)
        hist = tr.fit(m, batches)
        assert [h["epoch"] for h in hist] == [1]
        res[capture] = torch.cat([p.detach().flatten().float().cpu() for p in m.model.parameters()])
    scale = float(res[False].abs().max())
    assert float((res[False] - res[True]).abs().max()) <= 2e-5 * scale


@pytest.fixture(params=[True, False], ids=["event_local_conv", "chip_wide_conv"])
def conv_build(request):
    """Both device-count builds of a regular conv: csrc/evconv.hip (one launch, one workgroup per event; the default)
    and rulebook.hip's chip-wide build."""
    from waveformml_amd.spconv import ops
    old = ops.EVENT_LOCAL_CONV
    ops.EVENT_LOCAL_CONV = request.param
    yield request.param
    ops.EVENT_LOCAL_CONV = old


@pytest.mark.parametrize("shape,ksize,stride,padding", [((14, 11, 64), (3, 3, 3), (1, 1, 4), (0, 0, 0)), ((14, 11), (3, 3), (1, 1), (0, 0)),
                                                        ((12, 30), (3, 3), (2, 2), (1, 1)), ((9, 8, 20), (2, 2, 2), (2, 2, 2), (0, 0, 0)),
                                                        ((14, 11, 64), (3, 3, 3), (1, 1, 4), (0, 0, 1)), ((10, 7, 33), (3, 2, 3), (2, 1, 3), (1, 0, 2)),
                                                        ((40,), (3,), (2,), (1,)), ((6, 5, 4, 12), (1, 2, 2, 3), (1, 1, 2, 2), (0, 0, 1, 1))])
def test_device_count_conv_build_equals_exact_build(shape, ksize, stride, padding, conv_build):
    """The regular / strided conv build with device-side counts (what a captured step runs: capacities for N and M, the
    valid counts in device memory) against the exact-size build, which the oracle tests pin: same output coordinates in
    spconv's first-seen order, same tables, bit for bit; rows beyond the device-side count never enter; a capacity that
    is too small raises the overflow flag and keeps the first rows."""
    from waveformml_amd.spconv import ops
    rng = np.random.default_rng(17)
    B = 7
    ndim = len(shape)
    idx = _sorted_by_event(rand_coords(rng, B, shape, min(1200, B * int(np.prod(shape)) // 3)))
    t = torch.from_numpy(idx).to(DEV)
    N = idx.shape[0]
    exact = ops.build_rulebook(t, B, list(shape), list(ksize), list(stride), list(padding), [1] * ndim, False, known_unique=True)
    cap_n = N + 313
    pad = torch.cat([t, torch.full((cap_n - N, ndim + 1), 99999, dtype=torch.int32, device=DEV)])
    nv = torch.tensor([N], dtype=torch.int64, device=DEV)
    cap_m = exact.M + 77
    rb = ops.build_rulebook(pad, B, list(shape), list(ksize), list(stride), list(padding), [1] * ndim, False, n_dev=nv,
                            out_capacity=cap_m)
    torch.cuda.synchronize()
    M = int(rb.m_dev)
    assert (rb.events_out is not None) == conv_build
    assert M == exact.M and int(rb.overflow) == 0
    assert torch.equal(rb.out_indices[:M], exact.out_indices)
    assert torch.equal(rb.nbr_out[:, :N], exact.nbr_out)
    assert torch.equal(rb.nbr_in[:, :M], exact.nbr_in)
    if conv_build:
        assert not rb.event_flags.any()
        # the event offsets of the OUTPUT rows, and the cell -> row map dense() takes
        out_b = exact.out_indices[:, 0].cpu().numpy()
        assert np.array_equal(rb.events_out[:B + 1].cpu().numpy(), np.searchsorted(out_b, np.arange(B + 1)))
        assert not rb.events_out[B + 1:].any()
        V = int(np.prod(rb.out_spatial_shape))
        cells = rb.cell_map[2].cpu().numpy()
        oi = exact.out_indices.cpu().numpy().astype(np.int64)
        lin = oi[:, 0]
        for d in range(ndim):
            lin = lin * rb.out_spatial_shape[d] + oi[:, 1 + d]
        want = np.full(B * V, -1, np.int32)
        want[lin] = np.arange(M, dtype=np.int32)
        assert np.array_equal(cells, want)
    # a capacity that is too small: flagged, the tables hold what fits
    small = ops.build_rulebook(pad, B, list(shape), list(ksize), list(stride), list(padding), [1] * ndim, False, n_dev=nv,
                               out_capacity=max(exact.M // 2, 1))
    torch.cuda.synchronize()
    assert int(small.overflow) == 1 and int(small.m_dev) == max(exact.M // 2, 1)
    keep = int(small.m_dev)
    assert torch.equal(small.out_indices[:keep], exact.out_indices[:keep])


def test_event_local_conv_chain_at_psd_scale_and_packed_tables():
    """The bench's two strided layers (14 x 11 x 256 -> 12 x 9 x 64 -> 10 x 7 x 16, 256 events) built event-locally, the
    second from the first one's output rows and event offsets, against the exact builds; the packed by-input table
    [9, N] expands to the dense [27, N] bit for bit."""
    from waveformml_amd.psd import synthetic
    from waveformml_amd.spconv import ops
    B, T = 256, 256
    c, _f, _y = synthetic.generate(B, T, 3, seed=1234)
    idx = torch.from_numpy(np.ascontiguousarray(c[:, [3, 0, 1, 2]])).to(DEV)
    N = idx.shape[0]
    geo = ([3] * 3, [1, 1, 4], [0] * 3, [1] * 3)
    e1 = ops.build_rulebook(idx, B, [14, 11, T], *geo, False, known_unique=True)
    e2 = ops.build_rulebook(e1.out_indices, B, e1.out_spatial_shape, *geo, False, known_unique=True)
    cap = N + 2048
    pad = torch.cat([idx, torch.full((cap - N, 4), 7, dtype=torch.int32, device=DEV)])
    nv = torch.tensor([N], dtype=torch.int64, device=DEV)
    store1, store2 = {}, {}
    for _rep in range(3):                     # the same state three times: epochs 1, 2, 3
        r1 = ops.build_rulebook(pad, B, [14, 11, T], *geo, False, n_dev=nv, out_capacity=e1.M + 1000, flags=store1,
                                want_cell_map=False)
        r2 = ops.build_rulebook(r1.out_indices, B, r1.out_spatial_shape, *geo, False, n_dev=r1.m_dev,
                                out_capacity=e2.M + 500, events=r1.events_out, flags=store2)
        torch.cuda.synchronize()
        assert r1.packed_kl == 3 and r1.nbr_out_packed.shape == (9, cap) and r1.cell_map is None
        assert int(r1.m_dev) == e1.M and int(r2.m_dev) == e2.M
        assert torch.equal(r1.out_indices[:e1.M], e1.out_indices) and torch.equal(r2.out_indices[:e2.M], e2.out_indices)
        assert torch.equal(r1.nbr_out[:, :N], e1.nbr_out) and torch.equal(r1.nbr_in[:, :e1.M], e1.nbr_in)
        assert torch.equal(r2.nbr_out[:, :e1.M], e2.nbr_out) and torch.equal(r2.nbr_in[:, :e2.M], e2.nbr_in)
        assert not r1.event_flags.any() and not r2.event_flags.any() and int(r1.overflow) == 0 and int(r2.overflow) == 0
    assert int(store1["conv_state"][0]) == 3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_packed_table_products_equal_the_dense_table_products(dtype):
    """dX and dW of a 32 -> 32 strided layer through the packed [9, N] table against the same kernels on the dense
    [27, N] table: the same gathers, the same order -> bit-identical."""
    from waveformml_amd.psd import synthetic
    from waveformml_amd.spconv import functional as Fsp
    from waveformml_amd.spconv import ops
    B, T = 64, 128
    c, _f, _y = synthetic.generate(B, T, 3, seed=5)
    idx = torch.from_numpy(np.ascontiguousarray(c[:, [3, 0, 1, 2]])).to(DEV)
    N = idx.shape[0]
    nv = torch.tensor([N - 17], dtype=torch.int64, device=DEV)
    rb = ops.build_rulebook(idx, B, [14, 11, T], [3] * 3, [1, 1, 4], [0] * 3, [1] * 3, False, n_dev=nv, out_capacity=2 * N)
    assert rb.packed_kl == 3
    M = rb.M
    g = torch.Generator(device=DEV).manual_seed(1)
    X = torch.randn((N, 32), device=DEV, generator=g).to(dtype)
    dY = torch.randn((M, 32), device=DEV, generator=g).to(dtype)
    W = torch.randn((27, 32, 32), device=DEV, generator=g)
    t_p, pk = rb.table_by_in(32, 32, X, 1)
    assert pk == 3 and t_p is rb.nbr_out_packed
    dx_p = Fsp.gather_conv(t_p, None, 27, -1, N, dY, W, True, None, nv, None, pk)
    dx_d = Fsp.gather_conv(rb.nbr_out, None, 27, -1, N, dY, W, True, None, nv)
    t_p, pk = rb.table_by_in(32, 32, X, 3)
    assert pk == 3
    dw_p = Fsp.gather_dw(t_p, 27, -1, N, X, dY, False, None, nv, False, None, pk)
    dw_d = Fsp.gather_dw(rb.nbr_out, 27, -1, N, X, dY, False, None, nv)
    torch.cuda.synchronize()
    nvv = int(nv)
    assert torch.equal(dx_p[:nvv], dx_d[:nvv]) and torch.equal(dw_p, dw_d)
    assert float(dx_d[:nvv].float().abs().max()) > 0 and float(dw_d.abs().max()) > 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("device_counts", [False, True])
def test_one_launch_conv_backward_equals_the_two_products(dtype, device_counts):
    """wfs_conv_backward (dW and dX of a 32 -> 32 layer in ONE launch, conv_mfma.hip k_bwd32_bf16) against wfs_gather_dw +
    wfs_gather_conv: the same bodies on the same tables -> bit-identical, for a SubM table (centre offset = the row
    itself), a strided layer's dense table and its packed [9, N] table."""
    from waveformml_amd.psd import synthetic
    from waveformml_amd.spconv import functional as Fsp
    from waveformml_amd.spconv import ops
    B, T = 64, 128
    c, _f, _y = synthetic.generate(B, T, 3, seed=9)
    idx = torch.from_numpy(np.ascontiguousarray(c[:, [3, 0, 1, 2]])).to(DEV)
    N = idx.shape[0]
    nv = torch.tensor([N - 33], dtype=torch.int64, device=DEV) if device_counts else None
    g = torch.Generator(device=DEV).manual_seed(2)
    W = torch.randn((27, 32, 32), device=DEV, generator=g)
    X = torch.randn((N, 32), device=DEV, generator=g).to(dtype)
    sub = ops.build_rulebook(idx, B, [14, 11, T], [3] * 3, [1] * 3, [0] * 3, [1] * 3, True, n_dev=nv)
    con = ops.build_rulebook(idx, B, [14, 11, T], [3] * 3, [1, 1, 4], [0] * 3, [1] * 3, False, known_unique=True,
                             n_dev=nv, out_capacity=2 * N if device_counts else None)
    cases = [("subm", sub.nbr_out, 0, sub.centre_k, N)]
    cases.append(("conv dense", con.nbr_out, 0, -1, con.M))
    if con.nbr_out_packed is not None:
        cases.append(("conv packed", con.nbr_out_packed, con.packed_kl, -1, con.M))
    for name, table, pk, ident, rows_out in cases:
        dY = torch.randn((rows_out, 32), device=DEV, generator=g).to(dtype)
        dx1 = Fsp.gather_conv(table, None, 27, ident, N, dY, W, True, None, nv, None, pk)
        dw1 = Fsp.gather_dw(table, 27, ident, N, X, dY, False, None, nv, False, None, pk)
        dx2, dw2 = Fsp.conv_backward(table, 27, ident, N, X, dY, W, nv, None, pk)
        torch.cuda.synchronize()
        nvv = int(nv) if nv is not None else N
        assert torch.equal(dx1[:nvv], dx2[:nvv]), name
        assert torch.equal(dw1, dw2), name
        assert float(dw1.abs().max()) > 0 and float(dx1[:nvv].float().abs().max()) > 0


def test_event_local_conv_build_flags_an_ungrouped_batch_and_leaves_benign_tables():
    """A batch that is not grouped by event: flag [0] is set, the outputs are empty (m_dev = 0) and every by-input table
    entry the consumers can reach says "none" -- a captured step then computes garbage-free zeros until check() raises."""
    from waveformml_amd.spconv import ops
    rng = np.random.default_rng(3)
    B, shape = 6, (14, 11, 64)
    idx = _sorted_by_event(rand_coords(rng, B, shape, 900))
    bad = idx.copy()
    bad[[10, 700]] = bad[[700, 10]]
    t = torch.from_numpy(bad).to(DEV)
    nv = torch.tensor([900], dtype=torch.int64, device=DEV)
    rb = ops.build_rulebook(t, B, list(shape), [3] * 3, [1, 1, 4], [0] * 3, [1] * 3, False, n_dev=nv, out_capacity=4000)
    torch.cuda.synchronize()
    assert int(rb.event_flags[0]) == 1 and int(rb.m_dev) == 0
    assert bool((rb.nbr_out_packed == -1).all()) and bool((rb.cell_map[2] == -1).all())
    # an index outside the spatial shape: flag [2], the row has no outputs, everything else is built
    out = idx.copy()
    out[5, 3] = 64
    rb = ops.build_rulebook(torch.from_numpy(out).to(DEV), B, list(shape), [3] * 3, [1, 1, 4], [0] * 3, [1] * 3, False,
                            n_dev=nv, out_capacity=4000)
    torch.cuda.synchronize()
    assert int(rb.event_flags[2]) == 1 and int(rb.event_flags[0]) == 0 and bool((rb.nbr_out[:, 5] == -1).all())
    assert int(rb.m_dev) > 0


def test_overflow_flag_of_a_captured_build_is_sticky(conv_build):
    """A captured step is checked every so many replays: the overflow flag of a strided build must survive the replays
    that follow the one that overflowed (the build only ever SETS it; the reader clears it).  A build captured over a
    capacity that fits index set A is replayed with A (flag clear), with B (more output sites: flag set) and with A
    again: the flag is still set, and clearing it by hand brings it back to 0 for the next A."""
    from waveformml_amd.spconv import ops
    rng = np.random.default_rng(23)
    B, shape, ksize, stride, padding = 6, (14, 11, 64), [3, 3, 3], [1, 1, 4], [0, 0, 0]
    N = 900
    a = _sorted_by_event(rand_coords(rng, B, (4, 3, 64), N))                       # one corner of the grid: few outputs
    b = _sorted_by_event(rand_coords(rng, B, shape, N))                            # spread over the grid: many more
    ta, tb = torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)
    ma = ops.build_rulebook(ta, B, list(shape), ksize, stride, padding, [1] * 3, False, known_unique=True).M
    mb = ops.build_rulebook(tb, B, list(shape), ksize, stride, padding, [1] * 3, False, known_unique=True).M
    assert mb > ma + 64
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        static = ta.clone()
        nv = torch.tensor([N], dtype=torch.int64, device=DEV)
        for _ in range(2):
            ops.build_rulebook(static, B, list(shape), ksize, stride, padding, [1] * 3, False, n_dev=nv, out_capacity=ma + 8)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            rb = ops.build_rulebook(static, B, list(shape), ksize, stride, padding, [1] * 3, False, n_dev=nv,
                                    out_capacity=ma + 8)
        rb.overflow.zero_()                        # allocated inside the capture: the runner clears it once
        g.replay()
        torch.cuda.synchronize()
        assert int(rb.overflow) == 0 and int(rb.m_dev) == ma
        static.copy_(tb)
        g.replay()
        torch.cuda.synchronize()
        assert int(rb.overflow) == 1
        static.copy_(ta)
        g.replay()
        torch.cuda.synchronize()
        assert int(rb.overflow) == 1 and int(rb.m_dev) == ma          # a good replay does not clear it
        rb.overflow.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert int(rb.overflow) == 0

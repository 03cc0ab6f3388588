"""The event-parallel rulebook build of a whole layer stack (csrc/rulebook_chain.hip, spconv/ops.build_rulebook_chain)
against the sequential CPU algorithm (oracle/ref.py: spconv 1.2.1's getValidOutPos + first-seen numbering)
applied layer by layer, bit for bit: out_indices, indice_pairs, indice_pair_num of EVERY layer, plus the gather tables
and the cell -> row map against the per-layer builds of rulebook.hip."""
import os

import numpy as np
import pytest
import torch

from helpers import rand_coords

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sp():
    import waveformml_amd.spconv as sp
    return sp


def _waveform_like(rng, B, T, n_hits=3, shuffle_in_event=False):
    rows = []
    for b in range(B):
        ev = set()
        for _ in range(1 + rng.integers(0, n_hits)):
            x, y = rng.integers(0, 14), rng.integers(0, 11)
            t0 = rng.integers(0, T // 4)
            t1 = min(T, t0 + rng.integers(T // 8, T // 2))
            for t in range(t0, t1):
                ev.add((b, x, y, t))
        ev = sorted(ev)
        if shuffle_in_event:
            ev = [ev[i] for i in rng.permutation(len(ev))]
        rows += ev
    return np.asarray(rows, np.int32)


def _oracle_chain(idx, B, shape, specs):
    """[(out_indices, pairs, num, out_shape)] per layer, the CPU algorithm applied layer by layer."""
    from oracle import ref
    out, cur, shp = [], idx, list(shape)
    for (k, s, p, d, subm) in specs:
        o, pairs, num = ref.get_indice_pairs(cur, B, shp, k, s, p, d, 0, subm)
        oshape = shp if subm else ref.conv_output_shape(shp, k, s, p, d)
        out.append((o, pairs, num, list(oshape)))
        if not subm:
            cur, shp = o, list(oshape)
    return out


def _norm(v, n):
    return [int(v)] * n if np.isscalar(v) else [int(x) for x in v]


def _specs(ndim, raw):
    return [(_norm(k, ndim), _norm(s, ndim), _norm(p, ndim), _norm(d, ndim), subm) for (k, s, p, d, subm) in raw]


PSD_STACK = [(3, 1, 0, 1, True), (3, (1, 1, 4), 0, 1, False), (3, (1, 1, 4), 0, 1, False)]
STACKS = [
    ("psd_c2", 3, (14, 11, 64), PSD_STACK),
    ("conv_first", 3, (14, 11, 48), [(3, (1, 1, 2), 0, 1, False), (3, 1, 0, 1, True), ((1, 1, 3), (1, 1, 2), (0, 0, 1), 1, False)]),
    ("2d_pad_dil", 2, (14, 11), [((3, 3), 1, 0, 1, True), ((3, 3), 1, 1, 1, False), ((3, 3), 1, 2, 2, False), ((3, 3), 2, 1, 1, False)]),
    ("1d", 1, (97,), [(5, 3, 2, 1, False), (3, 1, 0, 1, True)]),
]


@pytest.mark.parametrize("name,ndim,shape,raw", STACKS, ids=[s[0] for s in STACKS])
@pytest.mark.parametrize("shuffled", [False, True], ids=["sorted", "shuffled_in_event"])
def test_chain_bitexact_vs_oracle(name, ndim, shape, raw, shuffled):
    sp = _sp()
    rng = np.random.default_rng(7)
    B = 9
    if ndim == 3:
        idx = _waveform_like(rng, B, shape[2], shuffle_in_event=shuffled)
    else:
        n = min(600, B * int(np.prod(shape)) // 3)
        idx = rand_coords(rng, B, shape, n)                    # events contiguous, random order inside an event
    idx = idx[idx[:, 0] != 4]                                  # an event without rows in the middle
    specs = _specs(ndim, raw)
    want = _oracle_chain(idx, B, shape, specs)
    rbs = sp.ops.build_rulebook_chain(torch.from_numpy(idx).to(DEV), B, list(shape), specs,
                                      cell_maps=[not s[4] for s in specs])
    assert rbs is not None and len(rbs) == len(specs)
    torch.cuda.synchronize()
    cur = torch.from_numpy(idx).to(DEV)
    for l, (rb, (o, pairs, num, oshape), spec) in enumerate(zip(rbs, want, specs)):
        assert rb.out_spatial_shape == oshape, l
        assert np.array_equal(rb.out_indices.cpu().numpy(), o), "out_indices of layer %d" % l
        assert np.array_equal(rb.indice_pair_num.cpu().numpy(), num), "indice_pair_num of layer %d" % l
        assert np.array_equal(rb.indice_pairs.cpu().numpy(), pairs), "indice_pairs of layer %d" % l
        # gather tables and the cell map against the per-layer build of the same geometry
        ref_rb = sp.ops._build_rulebook(cur, B, rb.geometry.spatial[:ndim], spec[0], spec[1], spec[2], spec[3], spec[4],
                                        known_unique=True)
        assert torch.equal(rb.nbr_out, ref_rb.nbr_out), "nbr_out of layer %d" % l
        if not spec[4]:
            assert torch.equal(rb.nbr_in, ref_rb.nbr_in), "nbr_in of layer %d" % l
            V = int(np.prod(oshape))
            ticket, slot = rb.cell_map[2]
            rows = torch.where(ticket == -1, torch.full_like(slot, -1), slot).cpu().numpy().reshape(B, V)
            want_rows = np.full((B, V), -1, np.int64)
            lin = np.zeros(len(o), np.int64)
            for d in range(ndim):
                lin = lin * oshape[d] + o[:, 1 + d]
            want_rows[o[:, 0], lin] = np.arange(len(o))
            assert np.array_equal(rows, want_rows), "cell map of layer %d" % l
            cur = rb.out_indices


def test_chain_at_psd_scale_with_device_counts():
    """256 events x 256 samples (BASELINE configs[1] geometry), capacity-padded index buffer + device-side row count,
    output capacities: the valid parts equal the CPU algorithm's, m_dev holds the true counts, no overflow."""
    sp = _sp()
    rng = np.random.default_rng(202)
    B, T = 256, 256
    idx = _waveform_like(rng, B, T)
    n = len(idx)
    specs = _specs(3, PSD_STACK)
    want = _oracle_chain(idx, B, (14, 11, T), specs)
    cap = n + 1000
    buf = torch.randint(0, 11, (cap, 4), dtype=torch.int32, device=DEV)       # garbage beyond the valid rows
    buf[:n] = torch.from_numpy(idx).to(DEV)
    n_dev = torch.tensor([n], dtype=torch.int64, device=DEV)
    caps = [None, len(want[1][0]) + 500, len(want[2][0]) + 100]
    rbs = sp.ops.build_rulebook_chain(buf, B, [14, 11, T], specs, n_dev=n_dev, capacities=caps)
    torch.cuda.synchronize()
    n_in = n
    for l, (rb, (o, pairs, num, _oshape)) in enumerate(zip(rbs, want)):
        m = len(o)
        if rb.subm:
            assert rb.m_dev is n_dev
        else:
            assert int(rb.m_dev) == m and int(rb.overflow) == 0, l
            assert np.array_equal(rb.out_indices[:m].cpu().numpy(), o), l
        # spconv's encoding from the padded table: the compaction honours n_dev
        assert np.array_equal(rb.indice_pair_num.cpu().numpy(), num), l
        got_pairs = rb.indice_pairs.cpu().numpy()
        for k in range(rb.K):
            assert np.array_equal(got_pairs[:, k, :num[k]], pairs[:, k, :num[k]]), (l, k)
        if not rb.subm:
            n_in = m
    # an output capacity that is too small is flagged, never silently cut
    caps[1] = len(want[1][0]) - 7
    rbs = sp.ops.build_rulebook_chain(buf, B, [14, 11, T], specs, n_dev=n_dev, capacities=caps)
    assert int(rbs[1].overflow) == 1 and int(rbs[1].m_dev) == caps[1] and int(rbs[2].overflow) == 0


def test_chain_refuses_what_it_cannot_do():
    """Rows not grouped by event, or an event beyond the LDS tables: exact-size mode returns None (the layers then
    build their own rulebooks), device-count mode raises every regular layer's overflow flag."""
    sp = _sp()
    rng = np.random.default_rng(5)
    B, T = 6, 64
    idx = _waveform_like(rng, B, T)
    specs = _specs(3, PSD_STACK)
    perm = rng.permutation(len(idx))
    mixed = torch.from_numpy(idx[perm]).to(DEV)
    assert sp.ops.build_rulebook_chain(mixed, B, [14, 11, T], specs) is None
    n_dev = torch.tensor([len(idx)], dtype=torch.int64, device=DEV)
    rbs = sp.ops.build_rulebook_chain(mixed, B, [14, 11, T], specs, n_dev=n_dev)
    assert int(rbs[1].overflow) == 1 and int(rbs[2].overflow) == 1
    # one event with 14 x 11 x 20 = 3080 active voxels (> 2048)
    big = np.asarray([(0, x, y, t) for x in range(14) for y in range(11) for t in range(20)], np.int32)
    assert sp.ops.build_rulebook_chain(torch.from_numpy(big).to(DEV), 1, [14, 11, T], specs) is None
    # ... and the stack still runs through the per-layer builds
    net = sp.SparseSequential(sp.SubMConv3d(2, 8, 3, 1, 0, 1, 1, False, "k"), sp.SparseConv3d(8, 8, 3, (1, 1, 4), 0, 1, 1, False),
                              sp.ToDense()).to(DEV)
    x = sp.SparseConvTensor(torch.ones((len(big), 2), device=DEV), torch.from_numpy(big).to(DEV), [14, 11, T], 1)
    x.unique = True
    assert net(x).shape == (1, 8, 12, 9, 16)


def test_sequential_uses_the_chain_and_matches_the_per_layer_path(monkeypatch):
    """SparseSequential builds all its conv layers' rulebooks through the chain (device-count mode, or known-unique
    indices): same dense output and gradients as with the chain switched off, bit for bit, and ONE chain build."""
    sp = _sp()
    rng = np.random.default_rng(17)
    B, T = 7, 64
    idx = _waveform_like(rng, B, T)
    feat = torch.from_numpy(rng.standard_normal((len(idx), 2)).astype(np.float32)).to(DEV)

    def build():
        torch.manual_seed(3)
        return sp.SparseSequential(
            sp.SubMConv3d(2, 32, 3, 1, 0, 1, 1, False, "k0"), torch.nn.BatchNorm1d(32), torch.nn.ReLU(),
            sp.SubMConv3d(32, 32, 3, 1, 0, 1, 1, False, "k0"), torch.nn.BatchNorm1d(32), torch.nn.ReLU(),
            sp.SparseConv3d(32, 32, 3, (1, 1, 4), 0, 1, 1, False), torch.nn.BatchNorm1d(32), torch.nn.ReLU(),
            sp.SparseConv3d(32, 32, 3, (1, 1, 4), 0, 1, 1, False), torch.nn.BatchNorm1d(32), torch.nn.ReLU(),
            sp.ToDense()).to(DEV)

    outs = []
    for on in (False, True):
        monkeypatch.setattr(sp.ops, "EVENT_LOCAL_RULEBOOKS", on)
        net = build()
        x = sp.SparseConvTensor(feat.clone().requires_grad_(True), torch.from_numpy(idx).to(DEV), [14, 11, T], B)
        x.unique = True
        n0 = sp.ops.CHAIN_BUILD_COUNT
        y = net(x)
        assert sp.ops.CHAIN_BUILD_COUNT - n0 == (1 if on else 0)
        y.square().sum().backward()
        outs.append((y.detach().clone(), [p.grad.clone() for p in net.parameters()]))
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)

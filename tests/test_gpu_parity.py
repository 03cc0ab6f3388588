"""GPU parity tests: libwfsparse (HIP, through the C ABI and the spconv-surface shim) vs the CPU
oracle on the same seeded inputs.  Bars (BASELINE.json north_star): rulebook indices BIT-EXACT;
fp32 features / logits / gradients within 1e-5 relative (stated per assert); bf16 storage is
compared with the fp32 oracle at a bf16 tolerance.
"""
import json
import os

import numpy as np
import pytest
import torch

from helpers import CASES, norm, rand_coords

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
DEV = "cuda:0"


def _sp():
    import waveformml_amd.spconv as sp
    return sp


def _assert_close(got, want, rtol=1e-5, what=""):
    """|got - want| <= rtol * max|want| + rtol * |want|  (relative to the tensor's scale)."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    scale = float(np.abs(want).max()) if want.size else 1.0
    np.testing.assert_allclose(got, want, rtol=rtol, atol=rtol * max(scale, 1e-30), err_msg=what)


def _golden():
    with open(os.path.join(HERE, "golden", "rulebook_small.json")) as f:
        return json.load(f)


def _rulebook_both(idx, B, shape, k, s, p, d, subm):
    from oracle import ref
    sp = _sp()
    want = ref.get_indice_pairs(idx, B, shape, k, s, p, d, 0, subm)
    got = sp.ops.get_indice_pairs(torch.from_numpy(idx).to(DEV), B, list(shape), k, s, p, d, 0, subm)
    torch.cuda.synchronize()
    return [g.cpu().numpy() for g in got], want


@pytest.mark.parametrize("case", _golden(), ids=lambda c: "D%d_%s" % (c["ndim"], "subm" if c["subm"] else "conv"))
def test_rulebook_golden_bitexact(case):
    """Committed fixture (independent Python transcription of A.3), incl. duplicate coordinates."""
    sp = _sp()
    idx = torch.tensor(case["indices"], dtype=torch.int32, device=DEV)
    out_idx, pairs, num = sp.ops.get_indice_pairs(idx, case["batch_size"], case["spatial_shape"], case["ksize"],
                                                  case["stride"], case["padding"], case["dilation"], 0, case["subm"])
    assert num.cpu().tolist() == case["indice_pair_num"]
    assert pairs.cpu().tolist() == case["indice_pairs"]
    assert out_idx.cpu().tolist() == case["out_indices"]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "D%d_k%s_s%s_p%s_d%s_%s" % (c[0], c[2], c[3], c[4], c[5], "subm" if c[6] else "conv"))
def test_rulebook_bitexact_vs_oracle(case):
    ndim, shape, k, s, p, d, subm = case
    rng = np.random.default_rng(101)
    B = 5
    n = min(2000, B * int(np.prod(shape)) // 2)
    idx = rand_coords(rng, B, shape, n)
    got, want = _rulebook_both(idx, B, shape, k, s, p, d, subm)
    for g, w, name in zip(got, want, ("out_indices", "indice_pairs", "indice_pair_num")):
        assert g.dtype == np.int32 and g.shape == w.shape, name
        assert np.array_equal(g, w), name


def _waveform_like(rng, B, T, n_hits=3):
    """PMT-grid x time voxels: a few segments per event, each a contiguous run of active samples."""
    rows = []
    for b in range(B):
        for _ in range(1 + rng.integers(0, n_hits)):
            x, y = rng.integers(0, 14), rng.integers(0, 11)
            t0 = rng.integers(0, T // 4)
            t1 = min(T, t0 + rng.integers(T // 8, T // 2))
            for t in range(t0, t1):
                rows.append((b, x, y, t))
    rows = sorted(set(rows))
    return np.asarray(rows, np.int32)


@pytest.mark.parametrize("subm,stride", [(True, 1), (False, (1, 1, 4)), (False, 2)])
def test_rulebook_bitexact_at_psd_scale(subm, stride):
    """14x11x256 grid, batch 256 (BASELINE config[1] geometry): hash path for SubM, direct grid for the
    strided layers; every entry of the rulebook must equal the sequential CPU algorithm's."""
    rng = np.random.default_rng(202)
    B, T = 256, 256
    idx = _waveform_like(rng, B, T)
    got, want = _rulebook_both(idx, B, (14, 11, T), 3, stride, 0, 1, subm)
    for g, w, name in zip(got, want, ("out_indices", "indice_pairs", "indice_pair_num")):
        assert np.array_equal(g, w), name
    assert want[2].sum() > len(idx)


def test_rulebook_empty_and_errors():
    sp = _sp()
    idx = torch.zeros((0, 4), dtype=torch.int32, device=DEV)
    o, p, n = sp.ops.get_indice_pairs(idx, 1, [14, 11, 16], 3, 1, 0, 1, 0, True)
    assert p.shape == (2, 27, 0) and int(n.sum()) == 0
    o, p, n = sp.ops.get_indice_pairs(idx, 1, [14, 11, 16], 3, 2, 0, 1, 0, False)
    assert o.shape == (0, 4) and int(n.sum()) == 0
    one = torch.zeros((1, 4), dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError):          # batch * volume >= 2^31, as spconv's C++ assert
        sp.ops.get_indice_pairs(one, 4096, [1024, 1024, 1024], 3, 1, 0, 1, 0, True)
    with pytest.raises(AssertionError):        # stride and dilation both > 1
        sp.ops.get_indice_pairs(one[:, :3].contiguous(), 1, [8, 8], 3, 2, 0, 2, 0, False)
    bad = torch.tensor([[0, 20, 0, 0]], dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError):          # coordinate outside spatial_shape
        sp.ops.get_indice_pairs(bad, 1, [14, 11, 16], 3, 1, 0, 1, 0, True)
    with pytest.raises(RuntimeError):          # CPU tensors: no fallback
        sp.ops.get_indice_pairs(one.cpu(), 1, [14, 11, 16], 3, 1, 0, 1, 0, True)


def _conv_pair(case, Cin, Cout, n, B, seed, dtype=torch.float32, bias=True):
    """Builds the same layer in the product (GPU) and in the oracle (CPU) with equal weights."""
    from oracle import spconv as osp
    sp = _sp()
    ndim, shape, k, s, p, d, subm = case
    rng = np.random.default_rng(seed)
    idx = rand_coords(rng, B, shape, n)
    feat = rng.standard_normal((n, Cin)).astype(np.float32)
    name = ("SubMConv%dd" if subm else "SparseConv%dd") % ndim
    torch.manual_seed(seed)
    ref_layer = getattr(osp, name)(Cin, Cout, k, s, p, d, 1, bias)
    layer = getattr(sp, name)(Cin, Cout, k, s, p, d, 1, bias).to(DEV)
    layer.load_state_dict(ref_layer.state_dict())
    xr = osp.SparseConvTensor(torch.from_numpy(feat).requires_grad_(True), torch.from_numpy(idx), list(shape), B)
    xg = sp.SparseConvTensor(torch.from_numpy(feat).to(DEV).to(dtype).requires_grad_(True),
                             torch.from_numpy(idx).to(DEV), list(shape), B)
    return layer, ref_layer, xg, xr, rng


@pytest.mark.parametrize("case", CASES, ids=lambda c: "D%d_k%s_s%s_p%s_d%s_%s" % (c[0], c[2], c[3], c[4], c[5], "subm" if c[6] else "conv"))
@pytest.mark.parametrize("chan", [(2, 32), (32, 32), (5, 7), (24, 16), (64, 48), (252, 158), (130, 146), (300, 33)],
                         ids=lambda c: "c%dx%d" % c)
def test_conv_forward_backward_fp32(case, chan):
    Cin, Cout = chan
    B = 3
    n = min(400, B * int(np.prod(case[1])) // 3)
    layer, ref_layer, xg, xr, rng = _conv_pair(case, Cin, Cout, n, B, 303)
    yg, yr = layer(xg), ref_layer(xr)
    assert np.array_equal(yg.indices.cpu().numpy(), yr.indices.numpy())
    assert yg.spatial_shape == [int(v) for v in yr.spatial_shape]
    _assert_close(yg.features.detach().cpu().numpy(), yr.features.detach().numpy(), 1e-5, "forward")
    g = rng.standard_normal(tuple(yr.features.shape)).astype(np.float32)
    yr.features.backward(torch.from_numpy(g))
    yg.features.backward(torch.from_numpy(g).to(DEV))
    _assert_close(xg.features.grad.cpu().numpy(), xr.features.grad.numpy(), 1e-5, "dX")
    _assert_close(layer.weight.grad.cpu().numpy(), ref_layer.weight.grad.numpy(), 1e-5, "dW")
    _assert_close(layer.bias.grad.cpu().numpy(), ref_layer.bias.grad.numpy(), 1e-5, "dbias")


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 2e-2), (torch.float16, 3e-3)], ids=["bf16", "f16"])
def test_conv_bf16_storage_against_fp32_oracle(dtype, tol):
    """16-bit rows in HBM, fp32 accumulate: bf16 has 8 mantissa bits -> tolerance 2e-2 of the tensor scale; fp16 (the
    reference's half_precision rows) 11 bits -> 3e-3."""
    case = (3, (14, 11, 24), 3, 1, 0, 1, True)
    layer, ref_layer, xg, xr, rng = _conv_pair(case, 32, 32, 600, 3, 404, dtype=dtype)
    xr.features = xr.features.detach().to(dtype).float()     # same rounded inputs
    yg, yr = layer(xg), ref_layer(xr)
    assert yg.features.dtype == dtype
    _assert_close(yg.features.detach().float().cpu().numpy(), yr.features.detach().numpy(), tol, "16-bit forward")


def test_inverse_conv_and_rulebook_reuse():
    from oracle import spconv as osp
    sp = _sp()
    rng = np.random.default_rng(505)
    shape, B, n = (14, 11), 4, 150
    idx = rand_coords(rng, B, shape, n)
    feat = rng.standard_normal((n, 6)).astype(np.float32)
    torch.manual_seed(1)
    ref_net = osp.SparseSequential(osp.SparseConv2d(6, 8, 3, 2, 1, 1, 1, True, indice_key="ind_0"), torch.nn.ReLU(),
                                   osp.SparseInverseConv2d(8, 5, 3, "ind_0", bias=False))
    net = sp.SparseSequential(sp.SparseConv2d(6, 8, 3, 2, 1, 1, 1, True, indice_key="ind_0"), torch.nn.ReLU(),
                              sp.SparseInverseConv2d(8, 5, 3, "ind_0", bias=False)).to(DEV)
    net.load_state_dict(ref_net.state_dict())
    fr = torch.from_numpy(feat).requires_grad_(True)
    fg = torch.from_numpy(feat).to(DEV).requires_grad_(True)
    yr = ref_net(osp.SparseConvTensor(fr, torch.from_numpy(idx), list(shape), B))
    xg = sp.SparseConvTensor(fg, torch.from_numpy(idx).to(DEV), list(shape), B)
    yg = net(xg)
    assert np.array_equal(yg.indices.cpu().numpy(), idx) and yg.spatial_shape == list(shape)
    _assert_close(yg.features.detach().cpu().numpy(), yr.features.detach().numpy(), 1e-5, "inverse fwd")
    yr.features.square().sum().backward()
    yg.features.square().sum().backward()
    _assert_close(fg.grad.cpu().numpy(), fr.grad.numpy(), 1e-5, "dX through inverse+conv")
    for a, b in zip(net.parameters(), ref_net.parameters()):
        _assert_close(a.grad.cpu().numpy(), b.grad.numpy(), 1e-5, "param grad")
    # the cached entry unpacks like spconv's 5-tuple and holds the oracle's rulebook
    outids, indices, pairs, num, spatial = xg.indice_dict["ind_0"]
    from oracle import ref
    o, p, c = ref.get_indice_pairs(idx, B, shape, 3, 2, 1, 1, 0, False)
    assert np.array_equal(pairs.cpu().numpy(), p) and np.array_equal(num.cpu().numpy(), c)
    assert np.array_equal(outids.cpu().numpy(), o) and spatial == list(shape)


@pytest.mark.parametrize("subm", [True, False], ids=["subm", "conv"])
def test_duplicate_coordinates_follow_the_cpu_semantics(subm):
    """Duplicate sites: SubM's hash keeps the last row, regular conv adds both rows (A.3/A.4)."""
    case = (2, (7, 6), 3, 1 if subm else 2, 0 if subm else 1, 1, subm)
    layer, ref_layer, xg, xr, rng = _conv_pair(case, 4, 6, 40, 2, 606)
    idx = xr.indices.numpy().copy()
    idx[31] = idx[30]
    idx[12] = idx[11]
    idx = idx[np.argsort(idx[:, 0], kind="stable")]
    xr.indices = torch.from_numpy(idx)
    xg.indices = torch.from_numpy(idx).to(DEV)
    yg, yr = layer(xg), ref_layer(xr)
    assert np.array_equal(yg.indices.cpu().numpy(), yr.indices.numpy())
    _assert_close(yg.features.detach().cpu().numpy(), yr.features.detach().numpy(), 1e-5, "dup forward")
    g = rng.standard_normal(tuple(yr.features.shape)).astype(np.float32)
    yr.features.backward(torch.from_numpy(g))
    yg.features.backward(torch.from_numpy(g).to(DEV))
    _assert_close(xg.features.grad.cpu().numpy(), xr.features.grad.numpy(), 1e-5, "dup dX")
    _assert_close(layer.weight.grad.cpu().numpy(), ref_layer.weight.grad.numpy(), 1e-5, "dup dW")


def test_to_dense_and_backward():
    from oracle import spconv as osp
    sp = _sp()
    rng = np.random.default_rng(707)
    shape, B, n, C = (10, 7, 16), 5, 500, 32
    idx = rand_coords(rng, B, shape, n)
    idx[77] = idx[76]                                  # duplicate: the later row must win
    feat = rng.standard_normal((n, C)).astype(np.float32)
    fr = torch.from_numpy(feat).requires_grad_(True)
    fg = torch.from_numpy(feat).to(DEV).requires_grad_(True)
    dr = osp.SparseConvTensor(fr, torch.from_numpy(idx), list(shape), B).dense()
    dg = sp.SparseConvTensor(fg, torch.from_numpy(idx).to(DEV), list(shape), B).dense()
    assert dg.shape == (B, C) + shape and dg.is_contiguous()
    assert np.array_equal(dg.detach().cpu().numpy(), dr.detach().numpy())
    w = rng.standard_normal(tuple(dr.shape)).astype(np.float32)
    (dr * torch.from_numpy(w)).sum().backward()
    (dg * torch.from_numpy(w).to(DEV)).sum().backward()
    assert np.array_equal(fg.grad.cpu().numpy(), fr.grad.numpy())


def test_small_psd_stack_logits_loss_and_grads():
    """A C2-shaped stack (SubM 2->32, SubM 32->32 sharing the rulebook, strided conv, ToDense, Linear)
    against the CPU restatement: logits and loss within 1e-5 relative, every gradient within 1e-5."""
    from oracle import spconv as osp
    sp = _sp()
    rng = np.random.default_rng(808)
    B, T = 8, 32
    idx = _waveform_like(rng, B, T)
    feat = rng.random((len(idx), 2)).astype(np.float32)
    labels = torch.from_numpy(rng.integers(0, 3, B))

    def build(m):
        return m.SparseSequential(
            # no conv bias in front of BatchNorm: its gradient is identically zero up to rounding noise
            m.SubMConv3d(2, 32, 3, 1, 0, 1, 1, False, "subm0"), torch.nn.BatchNorm1d(32), torch.nn.ReLU(),
            m.SubMConv3d(32, 32, 3, 1, 0, 1, 1, False, "subm0"), torch.nn.BatchNorm1d(32), torch.nn.ReLU(),
            m.SparseConv3d(32, 16, 3, (1, 1, 4), 0, 1, 1, True), torch.nn.ReLU(), m.ToDense())

    torch.manual_seed(2)
    ref_net, ref_head = build(osp), torch.nn.Linear(16 * 12 * 9 * 8, 3)
    net, head = build(sp).to(DEV), torch.nn.Linear(16 * 12 * 9 * 8, 3).to(DEV)
    net.load_state_dict(ref_net.state_dict())
    head.load_state_dict(ref_head.state_dict())
    crit = torch.nn.CrossEntropyLoss(reduction="mean")
    lr = ref_head(ref_net(osp.SparseConvTensor(torch.from_numpy(feat), torch.from_numpy(idx), [14, 11, T], B)).view(B, -1))
    lg = head(net(sp.SparseConvTensor(torch.from_numpy(feat).to(DEV), torch.from_numpy(idx).to(DEV), [14, 11, T], B)).view(B, -1))
    loss_r, loss_g = crit(lr, labels), crit(lg, labels.to(DEV))
    _assert_close(lg.detach().cpu().numpy(), lr.detach().numpy(), 1e-5, "logits")
    assert abs(loss_g.item() - loss_r.item()) <= 1e-5 * abs(loss_r.item())
    loss_r.backward()
    loss_g.backward()
    for (name, a), b in zip(net.named_parameters(), ref_net.parameters()):
        _assert_close(a.grad.cpu().numpy(), b.grad.numpy(), 1e-5, name)


@pytest.mark.parametrize("layer_kind", ["subm32", "conv32_s4", "subm2"])
def test_layers_at_bench_geometry_fp32(layer_kind):
    """The bench workload's own layers on synthetic events (many row tiles, every block/XCD range of the
    MFMA kernels in use): forward, dX and dW against the CPU oracle at 1e-5."""
    from oracle import spconv as osp
    from waveformml_amd.psd import synthetic
    sp = _sp()
    B, T = 96, 256
    c, f, _ = synthetic.generate(B, T, 3, seed=4321)
    idx = np.ascontiguousarray(c[:, [3, 0, 1, 2]])
    rng = np.random.default_rng(909)
    cin = 2 if layer_kind == "subm2" else 32
    feat = f if cin == 2 else rng.standard_normal((len(idx), 32)).astype(np.float32)
    torch.manual_seed(5)
    if layer_kind == "conv32_s4":
        mk = lambda m: m.SparseConv3d(32, 32, 3, (1, 1, 4), 0, 1, 1, True)
    else:
        mk = lambda m: m.SubMConv3d(cin, 32, 3, 1, 0, 1, 1, True, "k")
    ref_layer = mk(osp)
    layer = mk(sp).to(DEV)
    layer.load_state_dict(ref_layer.state_dict())
    fr = torch.from_numpy(feat).requires_grad_(True)
    fg = torch.from_numpy(feat).to(DEV).requires_grad_(True)
    yr = ref_layer(osp.SparseConvTensor(fr, torch.from_numpy(idx), [14, 11, T], B))
    yg = layer(sp.SparseConvTensor(fg, torch.from_numpy(idx).to(DEV), [14, 11, T], B))
    assert np.array_equal(yg.indices.cpu().numpy(), yr.indices.numpy())
    _assert_close(yg.features.detach().cpu().numpy(), yr.features.detach().numpy(), 1e-5, "forward")
    g = rng.standard_normal(tuple(yr.features.shape)).astype(np.float32)
    yr.features.backward(torch.from_numpy(g))
    yg.features.backward(torch.from_numpy(g).to(DEV))
    _assert_close(fg.grad.cpu().numpy(), fr.grad.numpy(), 1e-5, "dX")
    _assert_close(layer.weight.grad.cpu().numpy(), ref_layer.weight.grad.numpy(), 1e-5, "dW")


@pytest.mark.parametrize("C,relu,dtype", [(32, True, torch.float32), (32, False, torch.float32), (7, True, torch.float32),
                                          (252, True, torch.float32), (32, True, torch.bfloat16),
                                          (32, True, torch.float16), (7, False, torch.float16),
                                          (2048, True, torch.float32), (1300, True, torch.bfloat16),
                                          (1777, True, torch.float32), (355, False, torch.float16)],
                         ids=["c32_relu", "c32", "c7_relu", "c252_relu", "c32_relu_bf16", "c32_relu_f16", "c7_f16",
                              "c2048_relu_sliced", "c1300_relu_bf16_sliced", "c1777_relu_odd_sliced", "c355_f16_odd_sliced"])
def test_fused_batchnorm_relu_matches_torch(C, relu, dtype):
    """The fused BatchNorm1d(+ReLU) kernels against torch's own modules on the CPU in fp32 (what the
    reference's SparseSequential runs), training mode: output, running stats, dX, dgamma, dbeta."""
    from waveformml_amd.spconv import functional as Fsp
    rng = np.random.default_rng(1001)
    N = 5000 if C <= 256 else 777
    x = (rng.standard_normal((N, C)) * rng.uniform(0.5, 3.0, C) + rng.uniform(-20, 20, C)).astype(np.float32)
    g = rng.standard_normal((N, C)).astype(np.float32)
    torch.manual_seed(3)
    ref = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5)
        ref.bias.uniform_(-0.5, 0.5)
    bn = torch.nn.BatchNorm1d(C).to(DEV)
    bn.load_state_dict(ref.state_dict())
    xin = torch.from_numpy(x).to(dtype).float()          # the rounded values both sides see
    xr = xin.clone().requires_grad_(True)
    yr = ref(xr)
    if relu:
        yr = torch.relu(yr)
    yr.backward(torch.from_numpy(g))
    xg = xin.to(DEV).to(dtype).requires_grad_(True)
    assert Fsp.can_fuse_batch_norm(bn, xg)
    yg = Fsp.batch_norm_relu(xg, bn, relu)
    yg.backward(torch.from_numpy(g).to(DEV).to(dtype))
    tol = {torch.float32: 1e-5, torch.bfloat16: 2e-2, torch.float16: 2e-3}[dtype]
    _assert_close(yg.detach().float().cpu().numpy(), yr.detach().numpy(), tol, "y")
    _assert_close(bn.running_mean.cpu().numpy(), ref.running_mean.numpy(), 1e-5, "running_mean")
    _assert_close(bn.running_var.cpu().numpy(), ref.running_var.numpy(), 1e-5, "running_var")
    assert int(bn.num_batches_tracked) == 1
    if dtype == torch.float32:
        # a ReLU mask can flip where |pre-activation| ~ 1e-7; compare with a few-outliers-tolerant metric
        dx_g, dx_r = xg.grad.cpu().numpy(), xr.grad.numpy()
        bad = np.abs(dx_g - dx_r) > 1e-5 * np.abs(dx_r).max()
        assert bad.mean() < 1e-4, bad.mean()
        _assert_close(bn.weight.grad.cpu().numpy(), ref.weight.grad.numpy(), 1e-4, "dgamma")
        _assert_close(bn.bias.grad.cpu().numpy(), ref.bias.grad.numpy(), 1e-4, "dbeta")
    # eval mode uses the running statistics
    bn.eval(), ref.eval()
    ye = Fsp.batch_norm_relu(xin.to(DEV).to(dtype), bn, relu)
    yre = torch.relu(ref(xin)) if relu else ref(xin)
    _assert_close(ye.detach().float().cpu().numpy(), yre.detach().numpy(), tol, "eval y")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("layer_kind", ["subm32", "conv32_s4", "subm2"])
def test_layers_at_bench_geometry_bf16(layer_kind, dtype):
    """16-bit storage / 16-bit MFMA with fp32 accumulate against the fp32 oracle fed the same rounded
    inputs and rounded filters: what is left is accumulation order and the rounding of the
    outputs (bf16: 2^-9 relative -> 1e-2 of the tensor scale; fp16: 2^-12 -> 2e-3)."""
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-3
    from oracle import spconv as osp
    from waveformml_amd.psd import synthetic
    sp = _sp()
    B, T = 96, 256
    c, f, _ = synthetic.generate(B, T, 3, seed=4321)
    idx = np.ascontiguousarray(c[:, [3, 0, 1, 2]])
    rng = np.random.default_rng(910)
    cin = 2 if layer_kind == "subm2" else 32
    feat = f if cin == 2 else rng.standard_normal((len(idx), 32)).astype(np.float32)
    feat = torch.from_numpy(feat).to(dtype)
    torch.manual_seed(6)
    if layer_kind == "conv32_s4":
        mk = lambda m: m.SparseConv3d(32, 32, 3, (1, 1, 4), 0, 1, 1, True)
    else:
        mk = lambda m: m.SubMConv3d(cin, 32, 3, 1, 0, 1, 1, True, "k")
    ref_layer = mk(osp)
    layer = mk(sp).to(DEV)
    layer.load_state_dict(ref_layer.state_dict())
    with torch.no_grad():              # the MFMA kernels round the filters to the storage type
        ref_layer.weight.copy_(ref_layer.weight.to(dtype).float())
    fr = feat.float().requires_grad_(True)
    fg = feat.to(DEV).requires_grad_(True)
    yr = ref_layer(osp.SparseConvTensor(fr, torch.from_numpy(idx), [14, 11, T], B))
    yg = layer(sp.SparseConvTensor(fg, torch.from_numpy(idx).to(DEV), [14, 11, T], B))
    assert yg.features.dtype == dtype
    _assert_close(yg.features.detach().float().cpu().numpy(), yr.features.detach().numpy(), tol, "forward")
    g = torch.from_numpy(rng.standard_normal(tuple(yr.features.shape)).astype(np.float32)).to(dtype)
    yr.features.backward(g.float())
    yg.features.backward(g.to(DEV))
    _assert_close(fg.grad.float().cpu().numpy(), fr.grad.numpy(), tol, "dX")
    _assert_close(layer.weight.grad.cpu().numpy(), ref_layer.weight.grad.numpy(), tol, "dW")


def _c2_module(T, n_lin, dtype_seed=0):
    import copy
    import json
    from waveformml_amd.psd.config import DictionaryUtility
    from waveformml_amd.psd.lit import LitPSD
    root = os.path.dirname(HERE)
    cfg = json.load(open(os.path.join(root, "config", "psd_c2_3d.json")))
    cfg["system_config"]["n_samples"] = T
    cfg["net_config"]["algorithm"][-1] = [n_lin, 3]
    torch.manual_seed(11 + dtype_seed)
    return LitPSD(DictionaryUtility.to_object(copy.deepcopy(cfg)))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 6e-3), (torch.float16, 2e-3)])
@pytest.mark.parametrize("device_counts", [False, True])
def test_sparse_head_equals_the_dense_route(dtype, tol, device_counts):
    """ToDense -> view -> Linear off the sparse rows (csrc/shead.hip) against the dense route (dense() through the cell
    map + the streaming head kernels): same logits, same loss, same gradients of every parameter up to the summation
    order of fp32 sums (16-bit rows: up to one rounding of dX per row); with exact-size tensors and with capacity-padded
    ones (device-side counts).  The C ABI entry points are also called on their own: dX rows beyond the valid count stay
    untouched."""
    from waveformml_amd.psd import synthetic
    from waveformml_amd.spconv import functional as Fsp
    T, B = 64, 24
    c, f, y = synthetic.generate(B, T, 3, seed=31)
    coords, feats, labels = torch.from_numpy(c).to(DEV), torch.from_numpy(f).to(DEV).to(dtype), torch.from_numpy(y).to(DEV)
    n = coords.shape[0]
    if device_counts:
        pad = 300
        coords = torch.cat([coords, torch.zeros((pad, 4), dtype=coords.dtype, device=DEV)])
        feats = torch.cat([feats, torch.zeros((pad, feats.shape[1]), dtype=dtype, device=DEV)])
    res = {}
    old = Fsp.SPARSE_HEAD
    try:
        for on in (False, True):
            Fsp.SPARSE_HEAD = on
            mod = _c2_module(T, 32 * 10 * 7 * 4).to(DEV)
            mod.train()
            x = [coords, feats] + ([torch.tensor([n], dtype=torch.int64, device=DEV)] if device_counts else [])
            if device_counts:
                mod.model.batch_size_hint = B
                for m in mod.model.modules():
                    if hasattr(m, "subm") and not m.subm and not m.conv1x1:
                        m.out_capacity = 4 * n
            logits = mod.model(x)
            loss = mod.criterion(logits.float(), labels)
            loss.backward()
            torch.cuda.synchronize()
            res[on] = (logits.detach().float().cpu(), float(loss),
                       [p.grad.detach().float().cpu() for p in mod.model.parameters()])
    finally:
        Fsp.SPARSE_HEAD = old
    _assert_close(res[True][0].numpy(), res[False][0].numpy(), tol, "logits")
    assert abs(res[True][1] - res[False][1]) <= tol * abs(res[False][1])
    for i, (a, b) in enumerate(zip(res[True][2], res[False][2])):
        _assert_close(a.numpy(), b.numpy(), 20 * tol if dtype != torch.float32 else tol * 5, "gradient of parameter %d" % i)


def test_device_count_mode_equals_exact_size_mode():
    """Capacity-padded tensors + device-side row counts (no host read-back anywhere) must give the same
    loss and gradients as the ordinary exact-size path.  The forward is bit-identical; reductions over rows
    (dW, BatchNorm sums) are split by CAPACITY, so their fp32 summation order differs -> 1e-5 of each tensor's scale."""
    from waveformml_amd.psd import synthetic
    T, B = 64, 24
    mod = _c2_module(T, 32 * 10 * 7 * 4).to(DEV)
    c, f, y = synthetic.generate(B, T, 3, seed=77)
    coords, feats, labels = torch.from_numpy(c).to(DEV), torch.from_numpy(f).to(DEV), torch.from_numpy(y).to(DEV)
    mod.zero_grad()
    loss_a = mod.training_step(([coords, feats], labels), 0)
    loss_a.backward()
    grads_a = [p.grad.clone() for p in mod.model.parameters()]
    stats_a = [b.clone() for b in mod.model.buffers()]
    # padded: 30 % spare rows filled with garbage that must never be touched
    mod2 = _c2_module(T, 32 * 10 * 7 * 4).to(DEV)
    n, cap = coords.shape[0], int(coords.shape[0] * 1.3) + 17
    pc = torch.randint(0, 10, (cap, 4), dtype=torch.int32, device=DEV)
    pf = torch.full((cap, 2), float("nan"), device=DEV)
    pc[:n], pf[:n] = coords, feats
    n_valid = torch.tensor([n], dtype=torch.int64, device=DEV)
    loss_b = mod2.training_step(([pc, pf, n_valid], labels), 0)
    loss_b.backward()
    assert loss_a.item() == loss_b.item()
    for ga, p in zip(grads_a, mod2.model.parameters()):
        _assert_close(p.grad.cpu().numpy(), ga.cpu().numpy(), 1e-5, "gradient")
    for sa, b in zip(stats_a, mod2.model.buffers()):
        _assert_close(b.float().cpu().numpy(), sa.float().cpu().numpy(), 1e-6, "BatchNorm running statistics")
    convs = [m for m in mod2.modules() if getattr(m, "last_rulebook", None) is not None and not m.subm]
    assert convs and all(int(m.last_rulebook.overflow) == 0 for m in convs)
    assert all(int(m.last_rulebook.m_dev) <= m.last_rulebook.M for m in convs)


def test_graph_captured_step_matches_eager_steps():
    """The HIP-graph replay of the whole training step (psd/graph.py) against the same steps run eagerly."""
    from waveformml_amd.psd import synthetic
    from waveformml_amd.psd.ddp import FlatGradAllReducer
    from waveformml_amd.psd.graph import GraphedTrainStep
    T, B = 64, 24
    batches = []
    for s in (5, 6, 7):
        c, f, y = synthetic.generate(B, T, 3, seed=s)
        batches.append(([torch.from_numpy(c).to(DEV), torch.from_numpy(f).to(DEV)], torch.from_numpy(y).to(DEV)))

    def make():
        mod = _c2_module(T, 32 * 10 * 7 * 4).to(DEV)
        red = FlatGradAllReducer(mod.model.parameters(), world_size=1)
        mod.optimizer_parameters = red.optimizer_parameters()
        opt = mod.configure_optimizers()[0][0]
        return mod, red, opt

    mod_e, red_e, opt_e = make()
    mod_g, red_g, opt_g = make()
    start = [p.detach().clone() for p in mod_e.model.parameters()]
    # the graphed runner spends one calibration + two warm-up steps on its example batch before capturing
    step = GraphedTrainStep(mod_g, opt_g, red_g, batches[0], warmup=2)
    for _ in range(3):
        red_e.reset()
        mod_e.training_step(batches[0], 0).backward()
        red_e.finish()
        opt_e.step()
    for b in batches:
        red_e.reset()
        le = mod_e.training_step(b, 0)
        le.backward()
        red_e.finish()
        opt_e.step()
        lg = step(b)
        step.check()
        assert abs(lg.item() - le.item()) <= 1e-5 * max(abs(le.item()), 1e-6), (lg.item(), le.item())
    for p0, a, b in zip(start, mod_e.model.parameters(), mod_g.model.parameters()):
        # The padded (capacity) step cuts its row reductions at different places than the exact-size one.  That fp32
        # summation-order noise is fed back through six nesterov steps (momentum 0.98) and the batch statistics, so
        # the trajectories are compared through what the steps DID to a parameter: the two six-step updates agree to
        # 2e-3 of the update's size (single-step gradients agree to 1e-5: test_device_count_mode_equals_exact_size_mode).
        upd_e, upd_g = (a.detach() - p0).cpu().numpy(), (b.detach() - p0).cpu().numpy()
        _assert_close(upd_g, upd_e, 2e-3, "parameter update over 6 steps")


@pytest.mark.parametrize("dtype,O", [(torch.float32, 3), (torch.bfloat16, 3), (torch.float32, 8), (torch.float32, 1),
                                     (torch.float16, 3)],
                         ids=["f32_o3", "bf16_o3", "f32_o8", "f32_o1", "f16_o3"])
@pytest.mark.parametrize("I", [35840, 269], ids=["long_rows", "short_odd_rows"])
def test_skinny_linear_head_matches_torch(dtype, O, I):
    """I = 35840: the PSD head (streamed, vector loads).  I = 269: the second layer of the hybrid net's head
    (Linear(269, 3), reference src/models/SPConvNet.py:40-52): rows of any length on the scalar kernels."""
    from waveformml_amd.spconv import functional as Fsp
    torch.manual_seed(4)
    B = 37
    lin = torch.nn.Linear(I, O)
    x = torch.randn(B, I).to(dtype)
    g = torch.randn(B, O)
    xr = x.float().clone().requires_grad_(True)
    yr = lin(xr)
    yr.backward(g)
    ling = torch.nn.Linear(I, O).to(DEV)
    ling.load_state_dict(lin.state_dict())
    xg = x.to(DEV).requires_grad_(True)
    assert Fsp.can_use_skinny_linear(ling, xg)
    yg = Fsp.skinny_linear(xg, ling)
    yg.backward(g.to(DEV))
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    _assert_close(yg.detach().cpu().numpy(), yr.detach().numpy(), 1e-5, "y")
    _assert_close(xg.grad.float().cpu().numpy(), xr.grad.numpy(), tol, "dx")
    _assert_close(ling.weight.grad.cpu().numpy(), lin.weight.grad.numpy(), 1e-5, "dW")
    _assert_close(ling.bias.grad.cpu().numpy(), lin.bias.grad.numpy(), 1e-5, "db")


def test_side_stream_overlap_gives_identical_results():
    """Rulebook prefetch + dW on a side stream (ops.PREFETCH_RULEBOOKS / OVERLAP_DW) only reorder launches across
    streams: loss and every gradient must be bit-identical to the single-stream run, eagerly and under graph replay."""
    from waveformml_amd.psd import synthetic
    from waveformml_amd.psd.ddp import FlatGradAllReducer
    from waveformml_amd.psd.graph import GraphedTrainStep
    from waveformml_amd.spconv import functional as Fsp, ops
    T, B = 64, 24
    c, f, y = synthetic.generate(B, T, 3, seed=31)
    coords, feats, labels = torch.from_numpy(c).to(DEV), torch.from_numpy(f).to(DEV), torch.from_numpy(y).to(DEV)
    n = coords.shape[0]
    cap = n + 333
    pc = torch.zeros((cap, 4), dtype=torch.int32, device=DEV)
    pf = torch.zeros((cap, 2), device=DEV)
    pc[:n], pf[:n] = coords, feats
    nv = torch.tensor([n], dtype=torch.int64, device=DEV)
    results = []
    for flags in (False, True):
        ops.PREFETCH_RULEBOOKS = ops.OVERLAP_DW = flags
        try:
            mod = _c2_module(T, 32 * 10 * 7 * 4).to(DEV)
            for p in mod.model.parameters():
                p.grad = None
            loss = mod.training_step(([pc, pf, nv], labels), 0)
            loss.backward()
            Fsp.join_side_streams()
            torch.cuda.synchronize()
            results.append((loss.item(), [p.grad.clone() for p in mod.model.parameters()]))
        finally:
            ops.PREFETCH_RULEBOOKS = ops.OVERLAP_DW = False
    assert results[0][0] == results[1][0]
    for a, b in zip(results[0][1], results[1][1]):
        assert torch.equal(a, b)
    # graph replay with the overlaps on vs off: same losses over three different batches
    losses = []
    for flags in (False, True):
        ops.PREFETCH_RULEBOOKS = ops.OVERLAP_DW = flags
        try:
            mod = _c2_module(T, 32 * 10 * 7 * 4).to(DEV)
            red = FlatGradAllReducer(mod.model.parameters(), world_size=1)
            mod.optimizer_parameters = red.optimizer_parameters()
            opt = mod.configure_optimizers()[0][0]
            batches = []
            for s in (41, 42, 43):
                cc, ff, yy = synthetic.generate(B, T, 3, seed=s)
                batches.append(([torch.from_numpy(cc).to(DEV), torch.from_numpy(ff).to(DEV)], torch.from_numpy(yy).to(DEV)))
            step = GraphedTrainStep(mod, opt, red, batches[0])
            ls = []
            for bt in batches:
                ls.append(float(step(bt)))
                step.check()
            losses.append(ls)
        finally:
            ops.PREFETCH_RULEBOOKS = ops.OVERLAP_DW = False
    assert losses[0] == losses[1], losses


def test_training_from_hdf5_files_matches_the_cpu_path():
    """Whole input side of the path: native HDF5 reader -> items -> reference collate -> pinned double-buffered H2D
    (DevicePrefetcher) -> Trainer.fit on the GPU, against the CPU restatement stepping the same batches with plain
    SGD.  Two optimizer steps; weights after them within 1e-5 relative."""
    import copy
    from waveformml_amd.psd import data, h5data
    from waveformml_amd.psd.config import DictionaryUtility
    from waveformml_amd.psd.lit import LitPSD
    from waveformml_amd.psd.trainer import Trainer
    h5 = os.path.join(HERE, "golden", "h5")
    with open(os.path.join(HERE, "..", "config", "psd_c2_3d.json")) as f:
        cfg = json.load(f)
    T = 32
    cfg["system_config"]["n_samples"] = T
    cfg["net_config"]["algorithm"][-1] = [32 * 10 * 7 * 2, 3]
    cfg["optimize_config"]["optimizer_class"] = "optim.SGD"
    cfg["optimize_config"]["optimizer_params"] = {"momentum": 0.9}
    cfg["optimize_config"]["lr"] = 0.05
    cfg["optimize_config"].pop("scheduler_class", None)
    torch.manual_seed(5)
    gpu = LitPSD(DictionaryUtility.to_object(copy.deepcopy(cfg)))
    cpu_cfg = copy.deepcopy(cfg)
    cpu_cfg["net_config"]["imports"] = ["oracle.spconv" if m == "waveformml_amd.spconv" else m
                                        for m in cpu_cfg["net_config"]["imports"]]
    cpu = LitPSD(DictionaryUtility.to_object(cpu_cfg))
    cpu.load_state_dict(gpu.state_dict())

    ds = h5data.PulseDataset3D([os.path.join(h5, "Gamma"), os.path.join(h5, "Electron")], 7)
    assert len(ds) == 2                                   # Gamma/a (7 events, label 0), Electron/a (6 events, label 1)
    loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, collate_fn=data.collate_fn_3d)
    hist = Trainer(max_epochs=1, device=DEV).fit(gpu, loader)
    assert np.isfinite(hist[0]["train_loss"])

    opt = cpu.configure_optimizers()
    for i, batch in enumerate(loader):
        opt.zero_grad()
        loss = cpu.training_step(batch, i)
        loss.backward()
        opt.step()
    assert abs(hist[0]["train_loss"] - loss.item()) <= 1e-5 * abs(loss.item())
    for (name, a), b in zip(gpu.model.named_parameters(), cpu.model.parameters()):
        _assert_close(a.detach().cpu().numpy(), b.detach().numpy(), 1e-5, name)


@pytest.mark.parametrize("B,C", [(256, 3), (7, 2), (3000, 5)])
def test_fused_cross_entropy_matches_torch(B, C):
    """wfs_xent_mean_fwd_bwd against torch.nn.CrossEntropyLoss(reduction='mean') on the CPU in fp32: loss and the
    gradient w.r.t. the logits within 1e-6 relative, including ignored rows (ignore_index) and a scaled upstream
    gradient."""
    from waveformml_amd.spconv import functional as Fsp
    rng = np.random.default_rng(5)
    z = (rng.standard_normal((B, C)) * 4).astype(np.float32)
    t = rng.integers(0, C, B)
    t[::11] = -100
    crit = torch.nn.CrossEntropyLoss(reduction="mean")
    zr = torch.from_numpy(z).requires_grad_(True)
    lr = crit(zr, torch.from_numpy(t))
    (2.5 * lr).backward()
    zg = torch.from_numpy(z).to(DEV).requires_grad_(True)
    tg = torch.from_numpy(t).to(DEV)
    assert Fsp.can_fuse_cross_entropy(crit, zg, tg)
    lg = Fsp.cross_entropy_mean(zg, tg, crit.ignore_index)
    (2.5 * lg).backward()
    assert abs(lg.item() - lr.item()) <= 1e-6 * abs(lr.item())
    _assert_close(zg.grad.cpu().numpy(), zr.grad.numpy(), 1e-6, "dlogits")


@pytest.mark.parametrize("kw", [dict(momentum=0.98, nesterov=True), dict(momentum=0.9, dampening=0.1, weight_decay=1e-3),
                                dict()], ids=["nesterov", "dampening_wd", "plain"])
def test_flat_sgd_matches_torch_sgd(kw):
    """FlatSGD (one HIP launch on the flat parameter) against torch.optim.SGD on the CPU: four steps, the learning rate
    changed by a scheduler after the second, parameters and momentum buffers within 1e-6."""
    from waveformml_amd.psd.optim import FlatSGD
    rng = np.random.default_rng(9)
    w0 = rng.standard_normal(70001).astype(np.float32)
    pr = torch.nn.Parameter(torch.from_numpy(w0.copy()))
    pg = torch.nn.Parameter(torch.from_numpy(w0.copy()).to(DEV))
    ref, opt = torch.optim.SGD([pr], lr=0.02, **kw), FlatSGD([pg], lr=0.02, **kw)
    sr = torch.optim.lr_scheduler.ExponentialLR(ref, gamma=0.5)
    sg = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.5)
    for step in range(4):
        g = rng.standard_normal(70001).astype(np.float32)
        pr.grad = torch.from_numpy(g.copy())
        pg.grad = torch.from_numpy(g.copy()).to(DEV)
        ref.step()
        opt.step()
        if step == 1:
            sr.step()
            sg.step()
    _assert_close(pg.detach().cpu().numpy(), pr.detach().numpy(), 1e-6, "parameters")
    if kw.get("momentum"):
        _assert_close(opt.state[pg]["momentum_buffer"].cpu().numpy(), ref.state[pr]["momentum_buffer"].numpy(), 1e-6, "buf")
    sd = opt.state_dict()
    assert "_lr_dev" not in sd["param_groups"][0] and abs(sd["param_groups"][0]["lr"] - 0.01) < 1e-12


@pytest.mark.parametrize("levels,k,L,dtype", [(3, 3, 300, torch.float32), (1, 2, 64, torch.float32), (4, 5, 2048, torch.float32),
                                              (3, 3, 512, torch.bfloat16), (3, 3, 2048, torch.float16)],
                         ids=["l3_k3_L300", "l1_k2_L64", "l4_k5_L2048", "l3_k3_L512_bf16", "l3_k3_L2048_f16"])
def test_fused_temporal_conv_net_matches_torch(levels, k, L, dtype):
    """The hybrid net's waveform front end, TemporalConvNet(1, [1] * n_dil, k) (reference src/models/ConvBlocks.py:
    114-173 as built at src/models/SPConvNet.py:83-92): one HIP launch per direction (wfs_tcn_fwd / wfs_tcn_bwd) against
    the torch composition of weight-normed Conv1d / chomp / ReLU / residual on the CPU in fp32 -- output, dX and the
    gradients of every weight_g, weight_v and bias within 1e-5 (fp32 rows)."""
    from waveformml_amd.psd.tcn import TemporalConvNet
    rng = np.random.default_rng(31)
    N = 37
    torch.manual_seed(4)
    ref = TemporalConvNet(1, [1] * levels, kernel_size=k, dropout=0.0)
    with torch.no_grad():
        for p in ref.parameters():                  # the reference's N(0, 0.01) taps make every ReLU trivial: spread them
            p.copy_(torch.randn_like(p) * 0.7)
    net = TemporalConvNet(1, [1] * levels, kernel_size=k, dropout=0.0).to(DEV)
    net.load_state_dict(ref.state_dict())
    x = rng.standard_normal((N, 1, L)).astype(np.float32)
    g = rng.standard_normal((N, 1, L)).astype(np.float32)
    xin = torch.from_numpy(x).to(dtype).float()
    xr = xin.clone().requires_grad_(True)
    yr = ref.network(xr)                                            # the torch composition
    yr.backward(torch.from_numpy(g))
    xg = xin.to(DEV).to(dtype).requires_grad_(True)
    assert net._can_fuse(xg)
    yg = net(xg)
    yg.backward(torch.from_numpy(g).to(DEV).to(dtype))
    tol = {torch.float32: 1e-5, torch.bfloat16: 2e-2, torch.float16: 3e-3}[dtype]
    _assert_close(yg.detach().float().cpu().numpy(), yr.detach().numpy(), tol, "y")
    if dtype == torch.float16:
        _assert_close(xg.grad.float().cpu().numpy(), xr.grad.numpy(), tol, "dX")
    if dtype == torch.float32:
        _assert_close(xg.grad.cpu().numpy(), xr.grad.numpy(), 1e-5, "dX")
        for (name, a), b in zip(net.named_parameters(), ref.parameters()):
            _assert_close(a.grad.cpu().numpy(), b.grad.numpy(), 1e-4, name)
    # active dropout (training) is fused too (test_fused_temporal_conv_net_dropout); in eval mode it is the identity
    drop = TemporalConvNet(1, [1] * levels, kernel_size=k, dropout=0.2).to(DEV)
    with torch.no_grad():                       # (the Dropout modules shift the Sequential's keys: copy by position)
        for a, b in zip(drop.parameters(), net.parameters()):
            a.copy_(b)
    assert drop._can_fuse(xg) and drop.eval()._can_fuse(xg)
    assert torch.equal(drop(xg.detach()), net(xg.detach()))


def test_fused_temporal_conv_net_dropout():
    """Dropout inside the fused TCN kernels (training mode of the reference's hybrid front end, ConvBlocks.py:125-134:
    nn.Dropout(p) after each ReLU).  The masks are a counter-based hash of a device-side seed, so they cannot be compared
    bit for bit with torch's; checked instead: (1) the drop rate and the 1 / (1 - p) scale on a net whose convs are the
    identity, (2) determinism for a fixed seed and new masks for a new seed, (3) backward = adjoint of forward for the
    SAME masks (<g, J v> = <J^T g, v>; with fixed masks the net is piecewise linear in x, so a central difference gives
    J v exactly away from the ReLU kinks), (4) tap / bias gradients against central differences."""
    from waveformml_amd.psd.tcn import FusedTCNFunction, TemporalConvNet
    rng = np.random.default_rng(5)
    p = 0.3
    # (1) one level, k = 1, taps 1, biases 0 on positive rows: h1 = x m1, h2 = h1 m2, y = h2 + x
    N, L = 64, 2048
    x = torch.from_numpy(rng.uniform(0.5, 1.5, (N, L)).astype(np.float32)).to(DEV)
    taps = torch.ones((1, 2, 1), device=DEV)
    bias = torch.zeros((1, 2), device=DEV)
    seed = torch.tensor([1234567], dtype=torch.int64, device=DEV)
    y = FusedTCNFunction.apply(x, taps, bias, p, seed)
    ratio = ((y - x) / x).cpu().numpy()                       # m1 m2 in {0, 1 / (1 - p)^2}
    kept = ratio > 0
    assert abs(kept.mean() - (1 - p) ** 2) < 0.01, kept.mean()
    assert np.allclose(ratio[kept], 1.0 / (1 - p) ** 2, rtol=1e-5)
    assert abs(kept.mean(axis=1) - (1 - p) ** 2).max() < 0.06 and abs(kept.mean(axis=0) - (1 - p) ** 2).max() < 0.25
    # (2)
    assert torch.equal(FusedTCNFunction.apply(x, taps, bias, p, seed), y)
    y2 = FusedTCNFunction.apply(x, taps, bias, p, seed + 1)
    assert 0.3 < float(((y2 - x > 0) == (y - x > 0)).float().mean()) < 0.7      # independent masks agree ~58 % of the time
    # (3) + (4) on a 3-level net with spread-out taps
    levels, k, N, L = 3, 3, 9, 300
    taps = torch.from_numpy(rng.standard_normal((levels, 2, k)).astype(np.float32) * 0.6).to(DEV).requires_grad_(True)
    bias = torch.from_numpy(rng.standard_normal((levels, 2)).astype(np.float32) * 0.3).to(DEV).requires_grad_(True)
    x = torch.from_numpy(rng.standard_normal((N, L)).astype(np.float32)).to(DEV).requires_grad_(True)
    g = torch.from_numpy(rng.standard_normal((N, L)).astype(np.float32)).to(DEV)
    f = lambda xx, tt, bb: FusedTCNFunction.apply(xx, tt, bb, p, seed)          # noqa: E731
    y = f(x, taps, bias)
    y.backward(g)
    eps = 1e-3
    v = torch.from_numpy(rng.standard_normal((N, L)).astype(np.float32)).to(DEV)
    with torch.no_grad():
        jv = (f(x + eps * v, taps, bias).double() - f(x - eps * v, taps, bias).double()) / (2 * eps)
        lhs, rhs = float((g.double() * jv).sum()), float((x.grad.double() * v.double()).sum())
        assert abs(lhs - rhs) <= 2e-2 * max(abs(lhs), abs(rhs), 1.0), (lhs, rhs)
        for name, par in (("taps", taps), ("bias", bias)):
            flat = par.detach().reshape(-1)
            for i in range(flat.numel()):
                dpar = torch.zeros_like(flat)
                dpar[i] = 2e-4                   # (2e-3 is already visibly nonlinear in the taps)
                dpar = dpar.reshape(par.shape)
                args_p = (taps + dpar, bias) if name == "taps" else (taps, bias + dpar)
                args_m = (taps - dpar, bias) if name == "taps" else (taps, bias - dpar)
                fd = float((g.double() * (f(x, *args_p).double() - f(x, *args_m).double())).sum() / 4e-4)
                got = float(par.grad.reshape(-1)[i])
                assert abs(fd - got) <= 2e-2 * max(abs(fd), abs(got), 1.0), (name, i, fd, got)
    # the module draws a fresh seed per call in training mode
    net = TemporalConvNet(1, [1] * 2, kernel_size=3, dropout=p).to(DEV)
    xin = torch.randn(5, 1, 128, device=DEV)
    assert net._can_fuse(xin) and not torch.equal(net(xin), net(xin))
    torch.manual_seed(3)
    a = net(xin)
    torch.manual_seed(3)
    assert torch.equal(net(xin), a)


def test_hybrid_2d_net_with_waveform_front_end_matches_the_cpu_path():
    """The reference's hybrid configuration (GEP.json hparams with n_dil > 0, src/models/SPConvNet.py:71-109): the
    fused TemporalConvNet over the [N, 2T] waveform rows feeding the 2-D SparseConv2d stack (300 -> 252 -> 158 -> 64
    channels: the shape-generic conv kernels) and the LinearBlock, against the CPU restatement with the same weights:
    logits, loss and every gradient within 1e-5 / 1e-4."""
    import copy
    from waveformml_amd.psd import synthetic
    from waveformml_amd.psd.config import load_config
    from waveformml_amd.psd.lit import LitPSD
    cfg = json.load(open(os.path.join(HERE, "golden", "gep_config.json")))
    cfg["net_config"]["hparams"]["n_dil"] = 2
    cfg["net_config"]["hparams"]["wf_params"]["dropout"] = 0.0
    torch.manual_seed(21)
    gpu = LitPSD(load_config(copy.deepcopy(cfg)))
    with torch.no_grad():
        for p in gpu.model.waveformLayer.parameters():        # N(0, 0.01) taps would leave the front end almost linear
            p.copy_(torch.randn_like(p) * 0.5)
    cpu_cfg = copy.deepcopy(cfg)
    cpu_cfg["net_config"]["imports"] = ["oracle.spconv" if m == "waveformml_amd.spconv" else m
                                        for m in cpu_cfg["net_config"]["imports"]]
    cpu = LitPSD(load_config(cpu_cfg))
    cpu.load_state_dict(gpu.state_dict())
    gpu = gpu.to(DEV)
    c, f, y = synthetic.generate(8, 150, 3, seed=3, layout="2d")
    assert f.shape[1] == 300
    lc = cpu.training_step(([torch.from_numpy(c), torch.from_numpy(f)], torch.from_numpy(y)), 0)
    lg = gpu.training_step(([torch.from_numpy(c).to(DEV), torch.from_numpy(f).to(DEV)], torch.from_numpy(y).to(DEV)), 0)
    lc.backward()
    lg.backward()
    assert abs(lg.item() - lc.item()) <= 1e-5 * abs(lc.item()), (lg.item(), lc.item())
    for (name, a), b in zip(gpu.model.named_parameters(), cpu.model.parameters()):
        if b.grad is None:
            assert a.grad is None, name
            continue
        _assert_close(a.grad.cpu().numpy(), b.grad.numpy(), 1e-4, name)


def test_occlusion_sweep_reuses_rulebooks_and_matches_separate_evaluations():
    """Inference path (reference Evaluate.py --occlude / scripts/RunOcclusionStudy.py): one batch evaluated for several
    occluded feature columns.  With rulebook reuse the sweep builds each layer geometry's rulebook ONCE; its results
    are identical (same kernels, same inputs) to evaluating every index from scratch, and equal the CPU path's."""
    import copy
    from waveformml_amd.psd import synthetic
    from waveformml_amd.psd.config import load_config
    from waveformml_amd.psd.evaluate import occlusion_sweep
    from waveformml_amd.psd.lit import LitPSD
    sp = _sp()
    cfg = json.load(open(os.path.join(HERE, "golden", "gep_config.json")))
    torch.manual_seed(8)
    gpu = LitPSD(load_config(copy.deepcopy(cfg)))
    cpu_cfg = copy.deepcopy(cfg)
    cpu_cfg["net_config"]["imports"] = ["oracle.spconv" if m == "waveformml_amd.spconv" else m
                                        for m in cpu_cfg["net_config"]["imports"]]
    cpu = LitPSD(load_config(cpu_cfg))
    cpu.load_state_dict(gpu.state_dict())
    gpu = gpu.to(DEV)
    c, f, y = synthetic.generate(12, 150, 3, seed=4, layout="2d")
    batch = ([torch.from_numpy(c).to(DEV), torch.from_numpy(f).to(DEV)], torch.from_numpy(y).to(DEV))
    indices = [None, 0, 17, 150, 299]
    n0 = sp.ops.BUILD_COUNT
    sweep = occlusion_sweep(gpu, batch, indices)
    built = sp.ops.BUILD_COUNT - n0
    assert built == 2, built           # GEP's two 3x3 layers (the 1x1 layer needs no rulebook), once for all five passes
    cpu.eval()
    for idx in indices:
        gpu.occlude_index = idx
        cpu.occlude_index = idx
        with torch.no_grad():
            alone = gpu.eval().test_step(([batch[0][0], batch[0][1].clone()], batch[1]), 0)
            want = cpu.test_step(([torch.from_numpy(c), torch.from_numpy(f.copy())], torch.from_numpy(y)), 0)
        assert float(alone["test_loss"]) == sweep[idx]["test_loss"] and float(alone["test_acc"]) == sweep[idx]["test_acc"]
        assert abs(sweep[idx]["test_loss"] - float(want["test_loss"])) <= 1e-5 * abs(float(want["test_loss"]))
    assert sweep[None] == sweep[0]                               # the reference's falsy index 0
    assert sweep[17]["test_loss"] != sweep[None]["test_loss"]
    assert sp.ops.BUILD_COUNT - n0 == 2 + 2 * len(indices)      # the from-scratch evaluations rebuilt them each time


@pytest.mark.parametrize("algorithm", ["conv", "point"])
def test_litz_per_segment_regression_matches_the_cpu_path(algorithm):
    """The sibling task on the same surface: LitZ / SingleEndedZConv -- regular SparseConv2d
    layers with "same" padding, BatchNorm, ReLU, ToDense, and the segment loss built from SparseConvTensor.dense() --
    on the GPU against the CPU restatement with the same weights: loss 1e-5, gradients 1e-4."""
    import copy
    from test_host_mirror import _z_config
    from waveformml_amd.psd.config import load_config
    from waveformml_amd.psd.litz import LitZ
    cfg = _z_config(["waveformml_amd.spconv"])
    cfg["net_config"]["algorithm"] = algorithm
    torch.manual_seed(13)
    gpu = LitZ(load_config(copy.deepcopy(cfg)))
    cpu_cfg = copy.deepcopy(cfg)
    cpu_cfg["net_config"]["imports"] = ["torch.nn", "oracle.spconv"]
    cpu = LitZ(load_config(cpu_cfg))
    cpu.load_state_dict(gpu.state_dict())
    gpu = gpu.to(DEV)
    rng = np.random.default_rng(6)
    B = 9
    rows = sorted({(int(rng.integers(0, 14)), int(rng.integers(0, 11)), e) for e in range(B) for _ in range(4)},
                  key=lambda r: r[2])
    c = torch.tensor(rows, dtype=torch.int32)
    f = torch.from_numpy(rng.random((len(rows), 40)).astype(np.float32))
    z = torch.from_numpy(rng.standard_normal(len(rows)).astype(np.float32))
    lc = cpu.training_step(([c, f], z), 0)
    lg = gpu.training_step(([c.to(DEV), f.to(DEV)], z.to(DEV)), 0)
    lc.backward()
    lg.backward()
    assert abs(lg.item() - lc.item()) <= 1e-5 * abs(lc.item()), (lg.item(), lc.item())
    for (name, a), b in zip(gpu.model.named_parameters(), cpu.model.parameters()):
        if float(b.grad.abs().max()) < 1e-6:
            # a conv bias in front of BatchNorm: its gradient is identically zero up to rounding noise on both sides
            assert float(a.grad.abs().max()) < 1e-6, name
            continue
        _assert_close(a.grad.cpu().numpy(), b.grad.numpy(), 1e-4, name)


@pytest.mark.parametrize("cols,perm,C,dtype", [(4, [3, 0, 1, 2], 2, torch.float32), (3, [2, 0, 1], 301, torch.bfloat16)])
def test_batch_hand_over_kernel_copies_and_permutes(cols, perm, C, dtype):
    """wfs_load_batch: the one-launch hand-over of a batch into a captured step's fixed buffers (coordinates as they are
    and in the reference's batch-first order, features incl. a byte tail, labels, row count) against torch copies."""
    from waveformml_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(12)
    n, B, cap = 1237, 19, 1500
    coords = torch.from_numpy(rng.integers(0, 200, (n, cols)).astype(np.int32)).to(DEV)
    feats = torch.from_numpy(rng.standard_normal((n, C)).astype(np.float32)).to(DEV).to(dtype)
    labels = torch.from_numpy(rng.integers(0, 3, B)).to(DEV)
    cdst = torch.full((cap, cols), -7, dtype=torch.int32, device=DEV)
    idst = torch.full((cap, cols), -7, dtype=torch.int32, device=DEV)
    fdst = torch.zeros((cap, C), dtype=dtype, device=DEV)
    ldst = torch.zeros_like(labels)
    nv = torch.zeros((1,), dtype=torch.int64, device=DEV)
    # the batch index lives in the source column that the permutation moves to the front: make it a grouped event column
    ev_col = perm[0]
    evt = np.sort(rng.integers(0, B, n)).astype(np.int32)
    evt = evt[evt != 5]                                    # an event without rows
    n = len(evt)
    coords, feats = coords[:n].clone(), feats[:n].clone()
    coords[:, ev_col] = torch.from_numpy(evt).to(DEV)
    from waveformml_amd.spconv import ops
    events = torch.full((int(lib.wfs_event_offsets_ints(B)),), -3, dtype=torch.int32, device=DEV)
    _lib.check(lib.wfs_load_batch(_lib.ptr(coords), n, cols, _lib.i32_array(perm), _lib.ptr(cdst), _lib.ptr(idst),
                                  _lib.ptr(feats), _lib.ptr(fdst), feats.numel() * feats.element_size(), _lib.ptr(labels),
                                  _lib.ptr(ldst), B, _lib.ptr(nv), _lib.ptr(events), B, _lib.stream_ptr()))
    torch.cuda.synchronize()
    # the event offsets written by the same launch equal wfs_event_offsets on the batch-first copy, flag words included
    want = ops.event_offsets(idst[:n].contiguous(), B)
    assert torch.equal(events, want) and not bool(events[B + 1:].any())
    assert np.array_equal(events[:B + 1].cpu().numpy(), np.searchsorted(evt, np.arange(B + 1)))
    assert torch.equal(cdst[:n], coords) and bool((cdst[n:] == -7).all())
    assert torch.equal(idst[:n], coords[:, perm]) and bool((idst[n:] == -7).all())
    assert torch.equal(fdst.reshape(-1)[:n * C], feats.reshape(-1)) and float(fdst.reshape(-1)[n * C:].abs().sum()) == 0.0
    assert torch.equal(ldst, labels) and int(nv) == n






def _random_geometry(rng):
    ndim = int(rng.integers(1, 5))
    shape = [int(rng.integers(3, 10 if ndim > 2 else 24)) for _ in range(ndim)]
    subm = bool(rng.integers(0, 2))
    if subm:
        k = [int(rng.choice([1, 3, 3, 5])) for _ in range(ndim)]
        if ndim == 4:
            k = [min(v, 3) for v in k]
        s, p, d = [1] * ndim, [0] * ndim, [int(rng.choice([1, 1, 2])) for _ in range(ndim)]
    else:
        k = [int(rng.integers(1, 4 if ndim > 2 else 6)) for _ in range(ndim)]
        s = [int(rng.integers(1, 4)) for _ in range(ndim)]
        d = [1 if s[i] > 1 else int(rng.choice([1, 1, 2])) for i in range(ndim)]
        p = [int(rng.integers(0, k[i])) for i in range(ndim)]
        for i in range(ndim):                       # keep the output extent positive
            while (shape[i] + 2 * p[i] - d[i] * (k[i] - 1) - 1) // s[i] + 1 < 1:
                shape[i] += 1
    return ndim, shape, k, s, p, d, subm


@pytest.mark.parametrize("seed", range(24))
def test_rulebook_random_geometries_bitexact(seed):
    """Seeded random geometries (1-D .. 4-D, mixed kernel extents incl. kernel volumes > 32 that take the legacy kernels,
    strides, paddings, dilations, sparse and nearly full occupancy, hash and direct-grid site tables): out_indices,
    indice_pairs and indice_pair_num must equal the sequential CPU algorithm's bit for bit; every fourth case also
    feeds duplicate coordinates to the SubM build ("last row wins")."""
    rng = np.random.default_rng(9000 + seed)
    ndim, shape, k, s, p, d, subm = _random_geometry(rng)
    B = int(rng.integers(1, 6))
    vol = int(np.prod(shape))
    n = max(1, int(B * vol * rng.choice([0.02, 0.2, 0.9])))
    n = min(n, B * vol, 4000)
    idx = rand_coords(rng, B, shape, n)
    if subm and seed % 4 == 0 and n > 4:
        idx = np.concatenate([idx, idx[rng.integers(0, n, size=max(1, n // 10))]])
        idx = idx[np.argsort(idx[:, 0], kind="stable")]
    got, want = _rulebook_both(idx, B, shape, k, s, p, d, subm)
    for g, w, name in zip(got, want, ("out_indices", "indice_pairs", "indice_pair_num")):
        assert g.shape == w.shape, (name, ndim, shape, k, s, p, d, subm)
        assert np.array_equal(g, w), (name, ndim, shape, k, s, p, d, subm)


def test_fp16_storage_native_kernels():
    """The reference's ``half_precision`` feeds float16 features (src/datasets/HDF5Dataset.py:227-228).  fp16 rows have
    native kernels (fp16 storage and MFMA operands, fp32 accumulate, fp32 master weights).  A C2-shaped stack with
    fp16 features against the CPU restatement in fp32 fed the same fp16-rounded input: activations stay fp16 between
    layers, dense output within 5e-3 (eleven mantissa bits through three conv + BatchNorm layers), gradients flow in
    fp16 and the parameter gradients agree with the fp32 path's within 2e-2 of each tensor's scale."""
    from oracle import spconv as osp
    sp = _sp()
    rng = np.random.default_rng(404)
    B, T = 6, 32
    idx = _waveform_like(rng, B, T)
    feat = rng.random((len(idx), 2)).astype(np.float32)

    def build(m):
        return m.SparseSequential(
            m.SubMConv3d(2, 32, 3, 1, 0, 1, 1, False, "k0"), torch.nn.BatchNorm1d(32), torch.nn.ReLU(),
            m.SubMConv3d(32, 32, 3, 1, 0, 1, 1, False, "k0"), torch.nn.BatchNorm1d(32), torch.nn.ReLU(),
            m.SparseConv3d(32, 32, 3, (1, 1, 4), 0, 1, 1, False), torch.nn.BatchNorm1d(32), torch.nn.ReLU(), m.ToDense())

    torch.manual_seed(7)
    ref_net = build(osp)
    net = build(sp).to(DEV)
    net.load_state_dict(ref_net.state_dict())
    fin = torch.from_numpy(feat).half()
    fg = fin.to(DEV).requires_grad_(True)
    yg = net(sp.SparseConvTensor(fg, torch.from_numpy(idx).to(DEV), [14, 11, T], B))
    assert yg.dtype == torch.float16
    yr = ref_net(osp.SparseConvTensor(fin.float(), torch.from_numpy(idx), [14, 11, T], B))
    _assert_close(yg.detach().float().cpu().numpy(), yr.detach().numpy(), 5e-3, "dense output")
    yg.float().square().sum().backward()
    yr.square().sum().backward()
    assert fg.grad is not None and fg.grad.dtype == torch.float16 and bool(torch.isfinite(fg.grad).all())
    for (name, a), b_ in zip(net.named_parameters(), ref_net.parameters()):
        assert a.grad is not None and bool(torch.isfinite(a.grad).all()), name
        _assert_close(a.grad.float().cpu().numpy(), b_.grad.numpy(), 2e-2, name)


@pytest.mark.parametrize("dtype,tol_logits,tol_grad", [(torch.float32, 1e-5, 1e-4), (torch.float16, 5e-3, 0.15)],
                         ids=["f32", "f16"])
def test_c4_deep_stack_config_matches_the_cpu_path(dtype, tol_logits, tol_grad):
    """BASELINE.json configs[3] ("C4": config/psd_c4_deep_fp16.json, six SubMConv3d blocks sharing one rulebook + two
    strided layers + head) built by LitPSD from the config on both sides (GPU: waveformml_amd.spconv; CPU: the oracle's
    restatement, fp32): logits, loss and every parameter gradient of one training step.  fp32 rows meet the 1e-5 bar
    through the eight conv layers; fp16 rows (the config's half_precision) are compared with the fp32 CPU path on the
    same fp16-rounded input."""
    import copy
    import json
    from waveformml_amd.psd import synthetic
    from waveformml_amd.psd.config import DictionaryUtility
    from waveformml_amd.psd.lit import LitPSD
    T, B = 128, 24
    with open(os.path.join(HERE, "..", "config", "psd_c4_deep_fp16.json")) as f:
        cfg = json.load(f)
    cfg["system_config"]["n_samples"] = T
    cfg["net_config"]["algorithm"][-1] = [32 * 10 * 7 * 8, 3]
    torch.manual_seed(21)
    gpu = LitPSD(DictionaryUtility.to_object(copy.deepcopy(cfg)))
    cfg_ref = copy.deepcopy(cfg)
    cfg_ref["net_config"]["imports"] = ["oracle.spconv" if m == "waveformml_amd.spconv" else m
                                        for m in cfg_ref["net_config"]["imports"]]
    cpu = LitPSD(DictionaryUtility.to_object(cfg_ref))
    cpu.load_state_dict(gpu.state_dict())
    gpu = gpu.to(DEV)
    gpu.train(), cpu.train()
    c, f, y = synthetic.generate(B, T, 3, seed=99)
    fin = torch.from_numpy(f).to(dtype)
    loss_r = cpu.training_step(([torch.from_numpy(c), fin.float()], torch.from_numpy(y)), 0)
    loss_g = gpu.training_step(([torch.from_numpy(c).to(DEV), fin.to(DEV)], torch.from_numpy(y).to(DEV)), 0)
    assert abs(loss_g.item() - loss_r.item()) <= max(tol_logits, 1e-5) * abs(loss_r.item()), (loss_g.item(), loss_r.item())
    with torch.no_grad():
        lr = cpu.model([torch.from_numpy(c), fin.float()])
        lg = gpu.model([torch.from_numpy(c).to(DEV), fin.to(DEV)])
    _assert_close(lg.float().cpu().numpy(), lr.numpy(), tol_logits, "logits")
    loss_r.backward()
    loss_g.backward()
    for (name, a), b in zip(gpu.model.named_parameters(), cpu.model.parameters()):
        assert a.grad is not None and bool(torch.isfinite(a.grad).all()), name
        if dtype == torch.float32:
            _assert_close(a.grad.cpu().numpy(), b.grad.numpy(), tol_grad, name)
        else:
            # 16-bit activations: single elements of a filter gradient in front of a BatchNorm (a projection with heavy
            # cancellation) move by up to ~10 % of the tensor's scale -- from the ROUNDING of the stored activations, not
            # from fp16 underflow: tools/exp/f16_loss_scale.py gets the same error at loss scales 1 ... 65536.  The
            # tensor as a whole must agree: relative L2 error (measured 0.1 - 10 % per tensor for fp16, the sums over
            # rows that make dgamma / dbeta cancel heavily; bf16 rows: 1 - 26 %).
            err = float((a.grad.float().cpu() - b.grad).norm() / b.grad.norm().clamp_min(1e-30))
            assert err < tol_grad, (name, err)


@pytest.mark.parametrize("cout", [32, 12], ids=["c32", "c12"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("device_counts", [False, True], ids=["exact", "device_counts"])
def test_dense_through_the_conv_cell_map_equals_the_row_scatter(dtype, device_counts, cout):
    """dense() of a regular conv's output goes through the cell -> row map its rulebook build left behind
    (wfs_rulebook_cell_map + wfs_to_dense_mapped / _bwd_mapped: every cell written once, no zero fill).  It must equal
    the row-parallel dense() (the same tensor with the map taken away) bit for bit, forward and backward, and the CPU
    oracle's dense(); odd out volume per 64-cell tile (10 x 7 x 7 = 490 cells), capacity-padded rows with a device count."""
    from oracle import spconv as osp
    sp = _sp()
    rng = np.random.default_rng(99)
    B, T = 5, 30
    idx = _waveform_like(rng, B, T)
    n = len(idx)
    feat = rng.standard_normal((n, 32)).astype(np.float32)
    torch.manual_seed(4)
    ref_conv = osp.SparseConv3d(32, cout, 3, (1, 1, 4), 0, 1, 1, False)
    conv = sp.SparseConv3d(32, cout, 3, (1, 1, 4), 0, 1, 1, False).to(DEV)
    conv.load_state_dict(ref_conv.state_dict())
    fin = torch.from_numpy(feat).to(dtype)
    if device_counts:
        cap = n + 77
        pi = torch.zeros((cap, 4), dtype=torch.int32, device=DEV)
        pf = torch.full((cap, 32), float("nan"), dtype=dtype, device=DEV)
        pi[:n], pf[:n] = torch.from_numpy(idx).to(DEV), fin.to(DEV)
        x = sp.SparseConvTensor(pf.requires_grad_(True), pi, [14, 11, T], B)
        x.n_valid = torch.tensor([n], dtype=torch.int64, device=DEV)
        x.unique = True
    else:
        x = sp.SparseConvTensor(fin.to(DEV).requires_grad_(True), torch.from_numpy(idx).to(DEV), [14, 11, T], B)
    y = conv(x)
    assert y.cell_map is not None and y.cell_map[3] == 12 * 9 * 7
    dense_map = y.dense()
    plain = sp.SparseConvTensor(y.features, y.indices, y.spatial_shape, y.batch_size)
    plain.unique, plain.n_valid = y.unique, y.n_valid
    dense_rows = plain.dense()
    assert torch.equal(dense_map, dense_rows)
    w = torch.from_numpy(rng.standard_normal(tuple(dense_map.shape)).astype(np.float32)).to(DEV).to(dtype)
    (g_map,) = torch.autograd.grad((dense_map.float() * w.float()).sum(), y.features, retain_graph=True)
    (g_rows,) = torch.autograd.grad((dense_rows.float() * w.float()).sum(), y.features, retain_graph=True)
    m = int(y.n_valid) if y.n_valid is not None else y.features.shape[0]
    assert torch.equal(g_map[:m], g_rows[:m])
    if dtype == torch.float32:
        yr = ref_conv(osp.SparseConvTensor(torch.from_numpy(feat), torch.from_numpy(idx), [14, 11, T], B)).dense()
        _assert_close(dense_map.detach().cpu().numpy(), yr.detach().numpy(), 1e-5, "dense vs oracle")


def test_trainer_with_captured_steps_follows_the_eager_trainer():
    """Trainer(capture=True): the fit loop replays one captured HIP graph per step (psd/graph.GraphedTrainStep).  The
    capture's own calibration / warm-up steps must leave no trace (parameters, BatchNorm buffers and momentum are
    restored), so six steps over six different batches end where the eager trainer ends (update sizes within 2e-3, see
    test_graph_captured_step_matches_eager_steps); a batch larger than the captured capacity takes an eager step."""
    import copy
    from waveformml_amd.psd import data
    from waveformml_amd.psd.trainer import Trainer
    T = 64
    ds = data.SyntheticPulseDataset(6, 24, T, n_type=3, layout="3d", seed=77)
    loader = data.make_loader(ds, 1, shuffle=False, pin_memory=False)

    def run(capture):
        mod = _c2_module(T, 32 * 10 * 7 * 4)
        start = copy.deepcopy(mod.state_dict())
        tr = Trainer(max_epochs=1, device=DEV, capture=capture, check_every=2)
        hist = tr.fit(mod, loader, val_loader=loader)
        return mod, start, hist, tr

    mod_e, start, hist_e, _ = run(False)
    mod_g, _, hist_g, tr = run(True)
    assert tr.last_capacity > 0 and tr.eager_fallbacks == 0          # a step was captured (and closed at the end of fit)
    assert abs(hist_g[0]["train_loss"] - hist_e[0]["train_loss"]) <= 1e-4 * abs(hist_e[0]["train_loss"])
    assert abs(hist_g[0]["val_loss"] - hist_e[0]["val_loss"]) <= 1e-3 * abs(hist_e[0]["val_loss"])       # captured validation
    assert abs(hist_g[0]["val_acc"] - hist_e[0]["val_acc"]) <= 1.0 / 144 + 1e-9
    sd_e, sd_g = mod_e.state_dict(), mod_g.state_dict()
    for name in sd_e:
        a, b, p0 = sd_e[name].float().cpu(), sd_g[name].float().cpu(), start[name].float().cpu()
        if a.numel() == 1 and "num_batches_tracked" in name:
            assert int(a) == int(b) == 6, (name, int(a), int(b))          # warm-up steps of the capture left no trace
            continue
        _assert_close((b - p0).numpy(), (a - p0).numpy(), 2e-3, name)
    # batches that do not fit the captured step (twice the events): ordinary eager steps, training goes on
    mixed = [b for b in loader][:2] + [b for b in data.make_loader(ds, 2, shuffle=False, pin_memory=False)][:2]
    tr2 = Trainer(max_epochs=1, device=DEV, capture=True)
    tr2.fit(mod_g, mixed)
    assert tr2.eager_fallbacks == 2 and np.isfinite(tr2.history[-1]["train_loss"])


def test_captured_evaluation_matches_the_eager_loops():
    """psd/graph.GraphedEvalStep behind evaluate.test_loop(capture=True) and occlusion_sweep(capture=True): the eval-mode
    forward as replays of a captured graph (sweeps: rulebooks built by the first pass, a forward-only graph for every
    further occluded column) gives the losses / accuracies of the eager loops -- 3-D C2 net over several batches, one of
    them too large for the capture (eager fallback), and the 2-D GEP net for the sweep."""
    import copy
    import json
    from waveformml_amd.psd import data, synthetic
    from waveformml_amd.psd.config import load_config
    from waveformml_amd.psd.evaluate import occlusion_sweep, test_loop
    from waveformml_amd.psd.lit import LitPSD
    sp = _sp()
    T = 64
    mod = _c2_module(T, 32 * 10 * 7 * 4).to(DEV)
    with torch.no_grad():                                  # non-trivial running statistics for eval mode
        for m in mod.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.5, 1.5)
    ds = data.SyntheticPulseDataset(5, 24, T, n_type=3, layout="3d", seed=31)
    batches = [b for b in data.make_loader(ds, 1, shuffle=False, pin_memory=False)]
    batches.append([b for b in data.make_loader(ds, 2, shuffle=False, pin_memory=False)][0])      # 48 events: no fit
    eager = test_loop(mod, batches, DEV)
    graphed = test_loop(mod, batches, DEV, capture=True)
    assert graphed["events"] == eager["events"] == 5 * 24 + 48
    assert abs(graphed["test_loss"] - eager["test_loss"]) <= 1e-5 * abs(eager["test_loss"]), (graphed, eager)
    assert graphed["test_acc"] == eager["test_acc"]
    mod.occlude_index = 1
    assert abs(test_loop(mod, batches, DEV, capture=True)["test_loss"] - test_loop(mod, batches, DEV)["test_loss"]) <= 1e-5
    mod.occlude_index = None
    # occlusion sweep on the 2-D net
    cfg = json.load(open(os.path.join(HERE, "golden", "gep_config.json")))
    torch.manual_seed(8)
    gep = LitPSD(load_config(copy.deepcopy(cfg))).to(DEV)
    c, f, y = synthetic.generate(12, 150, 3, seed=4, layout="2d")
    batch = ([torch.from_numpy(c).to(DEV), torch.from_numpy(f).to(DEV)], torch.from_numpy(y).to(DEV))
    indices = [None, 0, 17, 150, 299]
    want = occlusion_sweep(gep, batch, indices)
    n0 = sp.ops.BUILD_COUNT
    got = occlusion_sweep(gep, batch, indices, capture=True)
    for idx in indices:
        assert abs(got[idx]["test_loss"] - want[idx]["test_loss"]) <= 1e-5 * abs(want[idx]["test_loss"]), (idx, got[idx], want[idx])
        assert got[idx]["test_acc"] == want[idx]["test_acc"]
    assert got[17]["test_loss"] != got[None]["test_loss"]
    # calibration (2 builds, exact sizes) + 2 builds in device-count mode; every later pass and the forward-only capture hit the cache
    assert sp.ops.BUILD_COUNT - n0 == 4, sp.ops.BUILD_COUNT - n0


def test_reducer_pack_with_gradients_partly_in_place():
    """FlatGradAllReducer._pack: the HIP operators write their parameters' gradients straight into the flat buffer
    (spconv/functional.grad_like), torch-side gradients (a conv bias: column sum of dY) arrive as separate tensors, and a
    parameter that took no part gets zeros -- the flat buffer must equal the per-parameter gradients of an unflattened
    twin, with and without deferred weight-gradient reductions."""
    from waveformml_amd.psd.ddp import FlatGradAllReducer
    from waveformml_amd.spconv import functional as Fsp
    sp = _sp()
    rng = np.random.default_rng(3)
    B, T = 4, 32
    idx = _waveform_like(rng, B, T)
    feat = torch.from_numpy(rng.standard_normal((len(idx), 2)).astype(np.float32)).to(DEV)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.body = sp.SparseSequential(
                sp.SubMConv3d(2, 32, 3, 1, 0, 1, 1, False, "k"), torch.nn.BatchNorm1d(32), torch.nn.ReLU(),
                sp.SubMConv3d(32, 32, 3, 1, 0, 1, 1, True, "k"), torch.nn.ReLU())      # bias=True: a torch-side gradient
            self.unused = torch.nn.Parameter(torch.ones(5))                              # never reached by backward

        def forward(self, x):
            return self.body(x).features.float().square().mean()

    torch.manual_seed(9)
    twin = Net().to(DEV)
    net = Net().to(DEV)
    net.load_state_dict(twin.state_dict())
    x = lambda: sp.SparseConvTensor(feat, torch.from_numpy(idx).to(DEV), [14, 11, T], B)      # noqa: E731
    twin(x()).backward()
    want = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in twin.named_parameters()}
    red = FlatGradAllReducer(net.parameters(), world_size=1)
    for deferred in (False, True):
        red.flat_grad.fill_(float("nan"))
        red.reset()
        Fsp.defer_dw(deferred)
        try:
            net(x()).backward()
            red.pack_all()
        finally:
            Fsp.defer_dw(False)
        in_place = 0
        for i, (n, p) in enumerate(net.named_parameters()):
            o, cnt = red.slices[i]
            got = red.flat_grad[o:o + cnt].view_as(p)
            _assert_close(got.cpu().numpy(), want[n].cpu().numpy(), 1e-6, "%s (deferred=%s)" % (n, deferred))
            in_place += int(p.grad is not None and p.grad.data_ptr() == got.data_ptr())
        assert in_place >= 4, in_place            # two conv weights, BatchNorm weight and bias were written in place


def test_gradient_slots_with_a_shared_module_and_with_accumulation():
    """spconv/functional.grad_like hands a parameter's slot of the flat gradient buffer out ONCE per backward pass and
    only while the parameter has no gradient: (1) a conv module called twice in one graph gets two gradients that
    autograd adds -- they must not alias (2 * g2 instead of g1 + g2); (2) a second backward() without reset()
    accumulates ``p.grad += new`` -- again the operands must not alias.  Both with and without deferred dW reductions,
    against an unflattened twin."""
    from waveformml_amd.psd.ddp import FlatGradAllReducer
    from waveformml_amd.spconv import functional as Fsp
    sp = _sp()
    rng = np.random.default_rng(12)
    B, T = 3, 24
    idx = _waveform_like(rng, B, T)
    feat = torch.from_numpy(rng.standard_normal((len(idx), 32)).astype(np.float32)).to(DEV)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = sp.SubMConv3d(32, 32, 3, 1, 0, 1, 1, False, "k")
            self.bn = torch.nn.BatchNorm1d(32)

        def forward(self, x):
            y = self.conv(x)
            z = sp.SparseConvTensor(torch.relu(self.bn(y.features)), y.indices, y.spatial_shape, y.batch_size)
            z.indice_dict = y.indice_dict
            return self.conv(z).features.float().square().mean()          # the SAME conv module again

    torch.manual_seed(2)
    twin, net = Net().to(DEV), Net().to(DEV)
    net.load_state_dict(twin.state_dict())
    x = lambda: sp.SparseConvTensor(feat, torch.from_numpy(idx).to(DEV), [14, 11, T], B)      # noqa: E731
    twin(x()).backward()
    once = {n: p.grad.clone() for n, p in twin.named_parameters()}
    twin(x()).backward()
    twice = {n: p.grad.clone() for n, p in twin.named_parameters()}
    red = FlatGradAllReducer(net.parameters(), world_size=1)
    for deferred in (False, True):
        red.reset()
        Fsp.defer_dw(deferred)
        try:
            net(x()).backward()
            red.pack_all()
        finally:
            Fsp.defer_dw(False)
        for i, (n, p) in enumerate(net.named_parameters()):
            o, cnt = red.slices[i]
            _assert_close(red.flat_grad[o:o + cnt].view_as(p).cpu().numpy(), once[n].cpu().numpy(), 1e-5,
                          "%s, module called twice (deferred=%s)" % (n, deferred))
    # gradient accumulation: a second backward without reset
    red.reset()
    net(x()).backward()
    net(x()).backward()
    red.pack_all()
    for i, (n, p) in enumerate(net.named_parameters()):
        o, cnt = red.slices[i]
        _assert_close(red.flat_grad[o:o + cnt].view_as(p).cpu().numpy(), twice[n].cpu().numpy(), 1e-5,
                      "%s, two backward passes" % n)
    with pytest.raises(TypeError):
        FlatGradAllReducer(torch.nn.Linear(4, 4).half().to(DEV).parameters(), world_size=1)


@pytest.mark.parametrize("R,C,dtype,padded", [(1, 5, torch.float32, False), (777, 32, torch.float32, True),
                                               (40000, 300, torch.bfloat16, True), (5000, 7, torch.float16, False),
                                               (0, 9, torch.float32, False)])
def test_column_sum_for_the_conv_bias_gradient(R, C, dtype, padded):
    """wfs_column_sum (the conv bias gradient: sum of dY over the valid rows) against a float64 sum of the same rounded
    rows; rows beyond a device-side count are ignored; deterministic (two calls agree bit for bit)."""
    from waveformml_amd.spconv import functional as Fsp
    rng = np.random.default_rng(3)
    cap = R + 100 if padded else R
    x = torch.from_numpy(rng.standard_normal((cap, C)).astype(np.float32)).to(DEV).to(dtype)
    if padded:
        x[R:] = 1e4                                              # must not be read into the sum
    r_dev = torch.tensor([R], dtype=torch.int64, device=DEV) if padded else None
    got = Fsp._masked_column_sum(x, r_dev)
    again = Fsp._masked_column_sum(x, r_dev)
    want = x[:R].double().sum(0).cpu().numpy()
    assert got.dtype == torch.float32 and got.shape == (C,)
    assert torch.equal(got, again)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=1e-5 * max(1.0, float(np.abs(x[:R].float().cpu().numpy()).sum(0).max() if R else 1.0)))


TRANSPOSED = [(2, (7, 6), 3, 2, 1, 1), (2, (14, 11), (2, 3), (1, 2), 0, 1), (3, (4, 5, 6), 3, (1, 1, 2), (1, 0, 1), 1),
              (2, (9, 9), 3, 1, 0, 2), (1, (33,), 4, 3, 1, 1), (3, (12, 9, 16), 3, (1, 1, 4), 0, 1)]


@pytest.mark.parametrize("case", TRANSPOSED, ids=lambda c: "D%d_k%s_s%s_p%s_d%s" % (c[0], c[2], c[3], c[4], c[5]))
def test_transposed_rulebook_bitexact_and_layer_parity(case):
    """spconv.SparseConvTranspose{2,3}d (the reference names them at src/utils/ModelValidation.py:30-31): rulebook -- output
    indices, indice_pairs, indice_pair_num -- bit for bit against the CPU restatement of get_indice_pairs(transpose=True);
    forward, dX, dW, db of the layer within 1e-5; device-count mode gives the same rows."""
    from oracle import ref
    from oracle import spconv as osp
    sp = _sp()
    ndim, shape, k, s, p, d = case
    rng = np.random.default_rng(77)
    B = 4
    n = min(600, B * int(np.prod(shape)) // 2)
    idx = rand_coords(rng, B, shape, n)
    want = ref.get_indice_pairs(idx, B, shape, k, s, p, d, 0, False, True)
    got = sp.ops.get_indice_pairs(torch.from_numpy(idx).to(DEV), B, list(shape), k, s, p, d, 0, False, True)
    for g, w, name in zip(got, want, ("out_indices", "indice_pairs", "indice_pair_num")):
        assert np.array_equal(g.cpu().numpy(), w), name
    Cin, Cout = 6, 9
    feat = rng.standard_normal((n, Cin)).astype(np.float32)
    torch.manual_seed(5)
    if ndim == 1:
        mk = lambda mod: mod.SparseConvolution(1, Cin, Cout, k, s, p, d, 1, True, transposed=True)      # noqa: E731
    else:
        mk = lambda mod: getattr(mod, "SparseConvTranspose%dd" % ndim)(Cin, Cout, k, s, p, d, 1, True)  # noqa: E731
    ref_layer = mk(osp)
    layer = mk(sp)
    layer.load_state_dict(ref_layer.state_dict())
    layer = layer.to(DEV)
    fr = torch.from_numpy(feat).requires_grad_(True)
    fg = torch.from_numpy(feat).to(DEV).requires_grad_(True)
    yr = ref_layer(osp.SparseConvTensor(fr, torch.from_numpy(idx), list(shape), B))
    yg = layer(sp.SparseConvTensor(fg, torch.from_numpy(idx).to(DEV), list(shape), B))
    assert list(yg.spatial_shape) == list(yr.spatial_shape)
    assert np.array_equal(yg.indices.cpu().numpy(), yr.indices.numpy())
    _assert_close(yg.features.detach().cpu().numpy(), yr.features.detach().numpy(), 1e-5, "forward")
    g = rng.standard_normal(tuple(yr.features.shape)).astype(np.float32)
    yr.features.backward(torch.from_numpy(g))
    yg.features.backward(torch.from_numpy(g).to(DEV))
    _assert_close(fg.grad.cpu().numpy(), fr.grad.numpy(), 1e-5, "dX")
    _assert_close(layer.weight.grad.cpu().numpy(), ref_layer.weight.grad.numpy(), 1e-5, "dW")
    _assert_close(layer.bias.grad.cpu().numpy(), ref_layer.bias.grad.numpy(), 1e-5, "db")
    # dense() of the transposed layer's output (cell map of the build) against the restatement's
    assert torch.equal(yg.dense().detach().cpu() != 0, yr.dense().detach() != 0)
    # device-count mode: capacity-padded rows, nothing read back
    cap = n + 37
    buf = torch.zeros((cap, ndim + 1), dtype=torch.int32, device=DEV)
    buf[:n] = torch.from_numpy(idx).to(DEV)
    n_dev = torch.tensor([n], dtype=torch.int64, device=DEV)
    kk, ss, pp, dd = (norm(v, ndim) for v in (k, s, p, d))
    rb = sp.ops.build_rulebook(buf, B, list(shape), kk, ss, pp, dd, False, n_dev=n_dev, transposed=True)
    m = int(rb.m_dev)
    assert m == len(want[0]) and int(rb.overflow) == 0
    assert np.array_equal(rb.out_indices[:m].cpu().numpy(), want[0])
    assert np.array_equal(rb.indice_pair_num.cpu().numpy(), want[2])


def test_three_piece_bf16_products_of_fp32_rows_over_a_wide_dynamic_range():
    """The fp32 32 -> 32 products run on the bf16 matrix cores with every number cut into three bf16 pieces
    (csrc/conv_mfma.hip k_gconv16_split / k_gdw32_split).  The cut is exact for every exponent, so the result must be
    as good as an fp32 dot product whatever the scale of a row: rows scaled by 2^-40 .. 2^40 (and filters by 2^-12 ..
    2^12), compared element by element with float64 against the fp32 bar of 1e-5 x sum |x| |w| -- a bar on each
    element's OWN scale, not the tensor's, which a lost low piece (2^-16 of the element) would miss by a factor 1.5."""
    from waveformml_amd.spconv import functional as Fsp
    rng = np.random.default_rng(4242)
    N, M, K = 1500, 1300, 27
    table = rng.integers(-1, N, size=(K, M)).astype(np.int32)
    table[rng.random((K, M)) < 0.6] = -1
    x = rng.standard_normal((N, 32)) * np.exp2(rng.integers(-40, 41, size=(N, 1)))
    w = rng.standard_normal((K, 32, 32)) * np.exp2(rng.integers(-12, 13, size=(K, 1, 1)))
    x32, w32 = x.astype(np.float32), w.astype(np.float32)
    X, W, T = torch.from_numpy(x32).to(DEV), torch.from_numpy(w32).to(DEV), torch.from_numpy(table).to(DEV)
    xd, wd = x32.astype(np.float64), w32.astype(np.float64)

    def bar(got, want, scale, what):
        err = np.abs(np.asarray(got, np.float64) - want)
        worst = float((err / np.maximum(scale, 1e-300)).max())
        assert worst <= 1e-5, "%s: %.3g of the element's own scale" % (what, worst)
        return worst

    for transpose in (False, True):
        wk = wd.transpose(0, 2, 1) if transpose else wd
        want = np.zeros((M, 32))
        scale = np.zeros((M, 32))
        for k in range(K):
            rows = np.nonzero(table[k] >= 0)[0]
            want[rows] += xd[table[k, rows]] @ wk[k]
            scale[rows] += np.abs(xd[table[k, rows]]) @ np.abs(wk[k])
        got = Fsp.gather_conv(T, None, K, -1, M, X, W, transpose, None)
        bar(got.cpu().numpy(), want, scale, "conv (transposed filters)" if transpose else "conv")
    # dW[k, a, b] = sum_r S[r, a] G[table[k, r], b] with rows of moderate scale on the stationary side (the sum runs over
    # rows: one row 2^80 above another would own the element in any arithmetic)
    s = rng.standard_normal((M, 32)) * np.exp2(rng.integers(-6, 7, size=(M, 1)))
    g = rng.standard_normal((N, 32)) * np.exp2(rng.integers(-6, 7, size=(N, 1)))
    s32, g32 = s.astype(np.float32), g.astype(np.float32)
    sd, gd = s32.astype(np.float64), g32.astype(np.float64)
    want = np.zeros((K, 32, 32))
    scale = np.zeros((K, 32, 32))
    for k in range(K):
        rows = np.nonzero(table[k] >= 0)[0]
        want[k] = sd[rows].T @ gd[table[k, rows]]
        scale[k] = np.abs(sd[rows]).T @ np.abs(gd[table[k, rows]])
    got = Fsp.gather_dw(T, K, -1, M, torch.from_numpy(s32).to(DEV), torch.from_numpy(g32).to(DEV), False)
    bar(got.cpu().numpy(), want, scale, "dW")

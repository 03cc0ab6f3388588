"""GPU checks of the wide-layer path (csrc/wide.hip; include/wfsparse.h wfs_wide_gather_conv and the wide arm of
wfs_gather_dw): layers with >= 128 channels on a side run as dense matrix-core products -- v_mfma_f32_32x32x16 for
16-bit rows, the exact-fp32 v_mfma_f32_32x32x2_f32 for fp32 rows (the 1e-5 path: compared to 1e-5 of the tensor scale
against fp64 on the unrounded operands).

Primitive level: random gather tables, odd channel counts (rows only 2-byte aligned), both product orders (the dense
product over the source rows + ordered sum, the gathered product over the destination rows, with and without the
split over kernel offsets), the SubM column map + identity offset, no table at all (1 x 1 conv), a device-side row count;
forward, dX (transposed filter) and dW (both orientations).  The reference is numpy fp64 on the SAME rounded operands
(rows and filters rounded to the 16-bit type, as the kernels do), so what is left is the fp32 accumulation order and
the rounding of the stored result: 16-bit results within one unit in the last place of the row type (2^-8 bf16 /
2^-11 fp16, relative) + 1e-4 of the tensor scale, fp32 results (dW) within 1e-4 of the tensor scale.

Layer level: the hybrid net's layer shapes (1 x 1 conv 300 -> 264, 3 x 3 conv 264 -> 130) forward and backward against
the CPU oracle's fp32 spconv restatement within the 16-bit tolerances of test_conv_bf16_storage_against_fp32_oracle.
"""
import numpy as np
import pytest
import torch

from helpers import rand_coords

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ULP = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11, torch.float32: 0.0}
TOL = {torch.bfloat16: 1e-4, torch.float16: 1e-4, torch.float32: 1e-5}
DTYPES = [torch.bfloat16, torch.float16, torch.float32]
DTYPE_IDS = ["bf16", "f16", "f32"]


def _round(a, dtype):
    return torch.from_numpy(a).to(dtype).float().numpy().astype(np.float64)


def _table(rng, K, R, src_rows, fill):
    t = rng.integers(0, src_rows, size=(K, R)).astype(np.int32)
    t[rng.random((K, R)) > fill] = -1
    return t


def _ref_conv(t, kmap, ident, R, X, W, transpose_w, bias, valid):
    K = W.shape[0]
    Cy = W.shape[1] if transpose_w else W.shape[2]
    Y = np.zeros((R, Cy))
    for k in range(K):
        src = np.arange(R) if (t is None or k == ident) else t[kmap[k] if kmap is not None else k]
        ok = (src >= 0) & (np.arange(R) < valid)
        Wk = W[k].T if transpose_w else W[k]
        Y[ok] += X[src[ok]] @ Wk
    if bias is not None:
        Y += bias
    return Y


CASES = [
    # K, R, X_rows, Cx, Cy, fill, kmap, ident, with_table
    (9, 300, 120, 257, 300, 0.6, False, -1, True),     # source side shorter: dense product + ordered sum
    (9, 120, 300, 300, 263, 0.5, False, -1, True),     # destination side shorter: gathered product, split over offsets
    (3, 700, 700, 512, 40, 0.7, False, -1, True),      # one column tile
    (9, 200, 200, 260, 264, 0.4, True, 4, True),       # SubM: mirrored column map + identity offset
    (1, 333, 333, 300, 252, 1.0, False, 0, False),     # 1 x 1 conv: no table
    (27, 90, 50, 64, 257, 0.3, False, -1, True),       # narrow input, wide output, 27 offsets
]


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
@pytest.mark.parametrize("case", CASES, ids=[str(c[:5]) for c in CASES])
@pytest.mark.parametrize("transpose_w", [False, True], ids=["fwd", "dx"])
def test_wide_gather_conv_against_fp64_on_rounded_operands(case, dtype, transpose_w):
    from waveformml_amd import _lib
    from waveformml_amd.spconv import functional as Fsp
    K, R, XR, Cx, Cy, fill, mirror, ident, with_table = case
    rng = np.random.default_rng(K * 1000 + R)
    X = rng.standard_normal((XR, Cx)).astype(np.float32)
    Wshape = (K, Cy, Cx) if transpose_w else (K, Cx, Cy)
    W = (rng.standard_normal(Wshape) * 0.05).astype(np.float32)
    bias = None if transpose_w else rng.standard_normal(Cy).astype(np.float32)
    t = _table(rng, K, R, XR, fill) if with_table else None
    kmap = list(range(K - 1, -1, -1)) if mirror else None
    assert _lib.load().wfs_wide_conv_ok(K, R, XR, Cx, Cy, _lib.dtype_code(torch.zeros(1, dtype=dtype)))
    for valid in (R, R - 37):
        Xg = torch.from_numpy(X).to(DEV).to(dtype)
        r_dev = None if valid == R else torch.tensor([valid], dtype=torch.int64, device=DEV)
        cmap = None if kmap is None else (_lib.ctypes.c_int32 * K)(*kmap)
        Y = Fsp.gather_conv(None if t is None else torch.from_numpy(t).to(DEV), cmap, K, ident, R, Xg,
                            torch.from_numpy(W).to(DEV), transpose_w, None if bias is None else torch.from_numpy(bias).to(DEV),
                            r_dev)
        torch.cuda.synchronize()
        assert Y.dtype == dtype and tuple(Y.shape) == (R, Cy)
        want = _ref_conv(t, kmap, ident, R, _round(X, dtype), _round(W, dtype), transpose_w, bias, valid)
        got = Y.float().cpu().numpy().astype(np.float64)[:valid]
        want = want[:valid]
        scale = np.abs(want).max()
        err = np.abs(got - want) - ULP[dtype] * np.abs(want)
        assert err.max() <= TOL[dtype] * scale, (err.max(), scale)


DW_CASES = [
    # K, R, G_rows, Cs, Cg, fill, ident, with_table
    (9, 300, 500, 257, 300, 0.6, -1, True),
    (9, 650, 200, 300, 131, 0.5, -1, True),
    (1, 333, 333, 300, 252, 1.0, 0, False),
    (27, 150, 150, 40, 260, 0.3, 13, True),
]


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
@pytest.mark.parametrize("case", DW_CASES, ids=[str(c[:5]) for c in DW_CASES])
@pytest.mark.parametrize("swap", [False, True], ids=["dw", "dw_swapped"])
def test_wide_gather_dw_against_fp64_on_rounded_operands(case, dtype, swap):
    from waveformml_amd.spconv import functional as Fsp
    K, R, GR, Cs, Cg, fill, ident, with_table = case
    rng = np.random.default_rng(K * 77 + R)
    S = rng.standard_normal((R, Cs)).astype(np.float32)
    G = rng.standard_normal((GR, Cg)).astype(np.float32)
    t = _table(rng, K, R, GR, fill) if with_table else None
    Sr, Gr = _round(S, dtype), _round(G, dtype)
    for valid in (R, R - 70):
        r_dev = None if valid == R else torch.tensor([valid], dtype=torch.int64, device=DEV)
        Sg = torch.from_numpy(S).to(DEV).to(dtype)
        if valid < R:
            Sg[valid:] = float("nan")                 # rows beyond the count must not be read into the sums
        dW = Fsp.gather_dw(None if t is None else torch.from_numpy(t).to(DEV), K, ident, R, Sg,
                           torch.from_numpy(G).to(DEV).to(dtype), swap, None, r_dev)
        torch.cuda.synchronize()
        want = np.zeros((K, Cs, Cg))
        for k in range(K):
            src = np.arange(R) if (t is None or k == ident) else t[k]
            ok = (src >= 0) & (np.arange(R) < valid)
            want[k] = Sr[ok].T @ Gr[src[ok]]
        if swap:
            want = want.transpose(0, 2, 1)
        got = dW.cpu().numpy().astype(np.float64)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= TOL[dtype] * np.abs(want).max()


def test_unaligned_views_are_read_correctly():
    """Rows that start on an odd element of a larger buffer (2-byte aligned only): the padded copy reads aligned dwords
    and shifts."""
    from waveformml_amd.spconv import functional as Fsp
    rng = np.random.default_rng(9)
    K, R, XR, Cx, Cy = 9, 100, 140, 257, 256
    big = torch.from_numpy(rng.standard_normal(XR * Cx + 3).astype(np.float32)).to(DEV).to(torch.bfloat16)
    W = (rng.standard_normal((K, Cx, Cy)) * 0.05).astype(np.float32)
    t = _table(rng, K, R, XR, 0.5)
    outs = []
    for off in (0, 1):
        Xv = big[off:off + XR * Cx].view(XR, Cx)
        assert Xv.data_ptr() % 4 == 2 * off
        Y = Fsp.gather_conv(torch.from_numpy(t).to(DEV), None, K, -1, R, Xv, torch.from_numpy(W).to(DEV), False, None)
        want = _ref_conv(t, None, -1, R, Xv.float().cpu().numpy().astype(np.float64), _round(W, torch.bfloat16), False, None, R)
        got = Y.float().cpu().numpy().astype(np.float64)
        err = np.abs(got - want) - ULP[torch.bfloat16] * np.abs(want)
        assert err.max() <= 1e-4 * np.abs(want).max()
        outs.append(got)


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 2e-2), (torch.float16, 3e-3), (torch.float32, 1e-5)],
                         ids=["bf16", "f16", "f32"])
def test_wide_layers_forward_backward_against_the_cpu_oracle(dtype, tol):
    """1 x 1 conv 300 -> 264 (spconv's torch.mm), BatchNorm, ReLU, 3 x 3 conv 264 -> 130 with a bias, as the reference's
    block generator stacks them (src/models/SPConvBlocks.py:450-516), against the fp32 CPU restatement on the same
    rounded input rows."""
    import waveformml_amd.spconv as sp
    from oracle import spconv as osp
    rng = np.random.default_rng(31)
    shape, B, n = (14, 11), 24, 420
    idx = rand_coords(rng, B, shape, n)
    idx = np.ascontiguousarray(idx[np.argsort(idx[:, 0], kind="stable")])
    feat = torch.from_numpy(rng.standard_normal((n, 300)).astype(np.float32)).to(dtype)
    torch.manual_seed(4)
    ref = osp.SparseSequential(osp.SparseConv2d(300, 264, 1, 1, 0, 1, 1, True), torch.nn.BatchNorm1d(264), torch.nn.ReLU(),
                               osp.SparseConv2d(264, 130, 3, 1, 0, 1, 1, True))
    net = sp.SparseSequential(sp.SparseConv2d(300, 264, 1, 1, 0, 1, 1, True), torch.nn.BatchNorm1d(264), torch.nn.ReLU(),
                              sp.SparseConv2d(264, 130, 3, 1, 0, 1, 1, True)).to(DEV)
    net.load_state_dict(ref.state_dict())
    fg = feat.to(DEV).requires_grad_(True)
    fr = feat.float().clone().requires_grad_(True)
    yr = ref(osp.SparseConvTensor(fr, torch.from_numpy(idx), list(shape), B))
    yg = net(sp.SparseConvTensor(fg, torch.from_numpy(idx).to(DEV), list(shape), B))
    assert yg.features.dtype == dtype
    assert np.array_equal(yg.indices.cpu().numpy(), yr.indices.numpy())

    def close(a, b, what, t=tol):
        a, b = a.detach().float().cpu().numpy(), b.detach().float().numpy()
        assert np.abs(a - b).max() <= t * np.abs(b).max(), (what, np.abs(a - b).max(), np.abs(b).max())

    def close_l2(a, b, what, t):
        a, b = a.detach().float().cpu().numpy().astype(np.float64), b.detach().float().numpy().astype(np.float64)
        assert np.linalg.norm(a - b) <= t * np.linalg.norm(b), (what, np.linalg.norm(a - b) / np.linalg.norm(b))

    close(yg.features, yr.features, "forward")
    g = torch.from_numpy(rng.standard_normal(tuple(yr.features.shape)).astype(np.float32))
    yr.features.backward(g)
    yg.features.backward(g.to(DEV).to(dtype))
    # Gradients that pass the ReLU are compared in the L2 norm: a 16-bit rounding of the BatchNorm output flips the
    # ReLU mask of the few elements that sit within a rounding step of zero, and each flip is a full-size error in one
    # element (measured, tools/exp/diag_wide2.py: the same net through the 32 x 32-tile kernels differs from this
    # path by 1e-2 in the L2 norm and 0.4 of the scale in single elements, with dX of the 3 x 3 layer equal to 3e-4).
    l2 = {torch.bfloat16: 6e-2, torch.float16: 2.5e-2, torch.float32: 1e-4}[dtype]
    close_l2(fg.grad, fr.grad, "dX", l2)
    for (name, a), (_n, b) in zip(net.named_parameters(), ref.named_parameters()):
        if name == "0.bias":
            continue          # a bias in front of a BatchNorm has a zero gradient: nothing to compare against
        if name.startswith("3."):
            close(a.grad, b.grad, name, 2 * tol)       # behind the ReLU: element-wise
        else:
            close_l2(a.grad, b.grad, name, l2)


@pytest.mark.parametrize("C,dtype", [(345, torch.bfloat16), (1697, torch.float32), (252, torch.float16)],
                         ids=["c345_bf16", "c1697_f32", "c252_f16"])
def test_wide_batchnorm_with_a_device_side_row_count(C, dtype):
    """The column-block BatchNorm kernels (csrc/bn.hip k_bnw_*) under a captured step's conditions: capacity rows
    beyond the device-side count hold NaN and must neither enter the statistics nor be touched; the valid rows must
    equal the same call on exactly the valid rows (bit for bit: same chunking is not required, so compare to 1e-6 of
    scale for fp32 and exactly-rounded 16-bit values within one unit in the last place)."""
    from waveformml_amd.spconv import functional as Fsp
    rng = np.random.default_rng(C)
    N, cap = 700, 900
    x = (rng.standard_normal((N, C)) * 2 + 3).astype(np.float32)
    g = rng.standard_normal((N, C)).astype(np.float32)
    outs = []
    for padded in (False, True):
        torch.manual_seed(1)
        bn = torch.nn.BatchNorm1d(C).to(DEV)
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.5, 0.5)
        xs = torch.from_numpy(x).to(DEV).to(dtype)
        gs = torch.from_numpy(g).to(DEV).to(dtype)
        n_dev = None
        if padded:
            xs = torch.cat([xs, torch.full((cap - N, C), float("nan"), device=DEV, dtype=dtype)])
            gs = torch.cat([gs, torch.full((cap - N, C), float("nan"), device=DEV, dtype=dtype)])
            n_dev = torch.tensor([N], dtype=torch.int64, device=DEV)
        xs.requires_grad_(True)
        y = Fsp.batch_norm_relu(xs, bn, True, n_dev)
        y.backward(gs)
        outs.append((y.detach()[:N].float().cpu().numpy(), xs.grad[:N].float().cpu().numpy(),
                     bn.weight.grad.cpu().numpy(), bn.bias.grad.cpu().numpy(), bn.running_var.cpu().numpy()))
    ulp = {torch.float32: 1e-6, torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}[dtype]
    for a, b in zip(outs[0], outs[1]):
        assert np.isfinite(b).all()
        assert np.abs(a - b).max() <= ulp * max(np.abs(a).max(), 1e-30)


@pytest.mark.parametrize("dtype", DTYPES, ids=DTYPE_IDS)
@pytest.mark.parametrize("B,I,O", [(256, 24150, 269), (37, 1000, 33), (300, 515, 700), (64, 512, 260)])
def test_wide_linear_against_fp64_on_rounded_operands(B, I, O, dtype):
    """wfs_linear16_fwd / _bwd (the hybrid net's Linear(24150, 269)): y, dX, dW, db against numpy fp64 on the operands
    as the kernels see them (x, W and dY rounded to the row type): fp32 results within 1e-4 of scale, dX within one unit
    in the last place of the row type."""
    from waveformml_amd.spconv import functional as Fsp
    rng = np.random.default_rng(B + I)
    x = rng.standard_normal((B, I)).astype(np.float32)
    lin = torch.nn.Linear(I, O).to(DEV)
    g = rng.standard_normal((B, O)).astype(np.float32)
    xg = torch.from_numpy(x).to(DEV).to(dtype).requires_grad_(True)
    assert Fsp.can_use_wide_linear(lin, xg)
    y = Fsp.wide_linear(xg, lin)
    assert y.dtype == torch.float32
    y.backward(torch.from_numpy(g).to(DEV))
    torch.cuda.synchronize()
    xr, wr, gr = _round(x, dtype), _round(lin.weight.detach().cpu().numpy(), dtype), _round(g, dtype)
    b = lin.bias.detach().cpu().numpy().astype(np.float64)

    def close(got, want, what, ulp=0.0):
        got = got.detach().float().cpu().numpy().astype(np.float64)
        err = np.abs(got - want) - ulp * np.abs(want)
        assert err.max() <= TOL[dtype] * np.abs(want).max(), (what, err.max(), np.abs(want).max())

    close(y, xr @ wr.T + b, "y")
    close(xg.grad, gr @ wr, "dX", ULP[dtype])
    close(lin.weight.grad, gr.T @ xr, "dW")
    close(lin.bias.grad, g.astype(np.float64).sum(0), "db")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)], ids=["f32", "bf16"])
def test_narrow_pointwise_conv_runs_in_the_library(dtype, tol):
    """A 1 x 1 SparseConv2d below the wide path's channel threshold (20 -> 12): spconv's torch.mm is replaced by the
    gather kernels with the identity map (no table); forward and backward against the CPU oracle."""
    import waveformml_amd.spconv as sp
    from oracle import spconv as osp
    rng = np.random.default_rng(5)
    shape, B, n = (14, 11), 6, 150
    idx = rand_coords(rng, B, shape, n)
    idx = np.ascontiguousarray(idx[np.argsort(idx[:, 0], kind="stable")])
    feat = torch.from_numpy(rng.standard_normal((n, 20)).astype(np.float32)).to(dtype)
    torch.manual_seed(2)
    ref = osp.SparseConv2d(20, 12, 1, 1, 0, 1, 1, True)
    layer = sp.SparseConv2d(20, 12, 1, 1, 0, 1, 1, True).to(DEV)
    layer.load_state_dict(ref.state_dict())
    fg = feat.to(DEV).requires_grad_(True)
    fr = feat.float().clone().requires_grad_(True)
    yr = ref(osp.SparseConvTensor(fr, torch.from_numpy(idx), list(shape), B))
    yg = layer(sp.SparseConvTensor(fg, torch.from_numpy(idx).to(DEV), list(shape), B))
    g = torch.from_numpy(rng.standard_normal(tuple(yr.features.shape)).astype(np.float32))
    yr.features.backward(g)
    yg.features.backward(g.to(DEV).to(dtype))
    for what, a, b in (("y", yg.features, yr.features), ("dX", fg.grad, fr.grad), ("dW", layer.weight.grad, ref.weight.grad),
                       ("db", layer.bias.grad, ref.bias.grad)):
        a, b = a.detach().float().cpu().numpy(), b.detach().float().numpy()
        assert np.abs(a - b).max() <= tol * np.abs(b).max(), (what, np.abs(a - b).max(), np.abs(b).max())

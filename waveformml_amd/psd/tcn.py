"""Causal dilated 1-D conv front end (hybrid config C5): mirror of the reference's TemporalConvNet
(src/models/ConvBlocks.py:105-173, itself the locuslab TCN): per level two weight-normed Conv1d with
left padding (k-1)*d chomped on the right, ReLU, dropout, residual; dilation doubles per level.

The reference only ever builds it with ONE channel (``TemporalConvNet(1, [1] * n_dil, ...)``,
src/models/SPConvNet.py:83-92).  In that shape every level is two k-tap causal FIR filters per waveform row, and the
whole net runs as one HIP launch per direction with the row resident in LDS (include/wfsparse.h, wfs_tcn_fwd /
wfs_tcn_bwd; csrc/tcn.hip), dropout included (training mode: masks from a counter-based hash of a seed drawn from torch's
generator -- the same distribution as nn.Dropout, not the same bits).  Same modules, parameters and state_dict as the
torch composition below, which remains the path for everything else (more channels, CPU tensors, rows too long for the
LDS-resident backward).
"""
import torch
from torch import nn
from torch.autograd import Function
from torch.nn.utils import weight_norm

from .. import _lib


class _Chomp(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.n = n

    def forward(self, x):
        return x[:, :, :-self.n].contiguous()


class TemporalBlock(nn.Module):
    def __init__(self, n_in, n_out, kernel_size, stride, dilation, padding, dropout=0.2):
        super().__init__()
        layers = []
        self.convs = []
        for cin in (n_in, n_out):
            conv = weight_norm(nn.Conv1d(cin, n_out, kernel_size, stride=stride, padding=padding, dilation=dilation))
            conv.weight.data.normal_(0, 0.01)
            self.convs.append(conv)
            layers += [conv, _Chomp(padding), nn.ReLU()]
            if dropout != 0:
                layers.append(nn.Dropout(dropout))
        self.net = nn.Sequential(*layers)
        self.downsample = nn.Conv1d(n_in, n_out, 1) if n_in != n_out else None
        if self.downsample is not None:
            self.downsample.weight.data.normal_(0, 0.01)
        self.relu = nn.ReLU()
        self.shape = (n_in, n_out, kernel_size, stride, dilation, padding, dropout)

    def forward(self, x):
        res = x if self.downsample is None else self.downsample(x)
        return self.relu(self.net(x) + res)


class FusedTCNFunction(Function):
    """The two fused kernels on EFFECTIVE taps (what the module's FusedNormedTCNFunction wraps with the weight norm):
    rows [N, L], effective taps [levels, 2, k] and biases [levels, 2] (fp32, on the GPU) -> rows [N, L].
    ``dropout`` > 0 with ``seed`` (int64 [1] on the GPU) applies the levels' dropout inside the kernels."""

    @staticmethod
    def forward(ctx, x, taps, bias, dropout=0.0, seed=None):
        lib = _lib.load()
        x = x.contiguous()
        taps, bias = taps.contiguous(), bias.contiguous()
        N, L = x.shape
        levels, _, k = taps.shape
        y = torch.empty_like(x)
        _lib.check(lib.wfs_tcn_fwd(_lib.ptr(x), N, L, _lib.ptr(taps), _lib.ptr(bias), levels, k, _lib.ptr(y),
                                   _lib.dtype_code(x), float(dropout), _lib.ptr(seed), _lib.stream_ptr()))
        ctx.save_for_backward(x, taps, bias)
        ctx.dropout, ctx.seed = float(dropout), seed
        return y

    @staticmethod
    def backward(ctx, grad_output):
        lib = _lib.load()
        x, taps, bias = ctx.saved_tensors
        N, L = x.shape
        levels, _, k = taps.shape
        dy = grad_output.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dx = torch.empty_like(x)
        partial = torch.empty((N, levels, 2, k + 1), dtype=torch.float32, device=x.device)
        _lib.check(lib.wfs_tcn_bwd(_lib.ptr(x), _lib.ptr(dy), N, L, _lib.ptr(taps), _lib.ptr(bias), levels, k, _lib.ptr(dx),
                                   _lib.ptr(partial), _lib.dtype_code(x), ctx.dropout, _lib.ptr(ctx.seed),
                                   _lib.stream_ptr()))
        sums = partial.sum(0)
        return dx, sums[:, :, :k].contiguous(), sums[:, :, k].contiguous(), None, None


def _ptr_table(cache, key, rows, device):
    """Device array of pointer records for wfs_tcn_taps_* (six int64 per convolution), cached by the addresses it holds:
    in a captured step parameters and gradient slots never move, so the table is built (one small H2D copy) during the
    eager warm-up only."""
    tab = cache.get(key)
    if tab is None:
        if len(cache) > 64:
            cache.clear()
        tab = torch.tensor(rows, dtype=torch.int64, device=device)
        cache[key] = tab
    return tab


class FusedNormedTCNFunction(Function):
    """rows [N, L] through the whole front end.  ``params``: per convolution (2 per level) the weight-norm parameters
    weight_v [1, 1, k], weight_g [1, 1, 1] and the bias [1] (or None) -- the effective taps w = g v / |v| are formed by ONE
    launch (wfs_tcn_taps_fwd), the backward's per-row partial sums are turned into d weight_v, d weight_g, d bias by ONE
    launch (wfs_tcn_taps_bwd) that writes straight into the parameters' gradient slots (spconv/functional.grad_like).
    ``dropout`` > 0 with ``seed`` (int64 [1] on the GPU) applies the levels' dropout inside the kernels."""

    @staticmethod
    def forward(ctx, x, k, dropout, seed, cache, *params):
        lib = _lib.load()
        x = x.contiguous()
        N, L = x.shape
        n_conv = len(params) // 3
        levels = n_conv // 2
        rows = []
        for c in range(n_conv):
            v, g, b = params[3 * c: 3 * c + 3]
            assert v.dtype == torch.float32 and v.is_contiguous() and v.numel() == k and g.numel() == 1
            rows.append([v.data_ptr(), g.data_ptr(), b.data_ptr() if b is not None else 0, 0, 0, 0])
        tab = _ptr_table(cache, ("fwd",) + tuple(r[0] for r in rows) + tuple(r[1] for r in rows) + tuple(r[2] for r in rows),
                         rows, x.device)
        taps = torch.empty((levels, 2, k), dtype=torch.float32, device=x.device)
        bias = torch.empty((levels, 2), dtype=torch.float32, device=x.device)
        _lib.check(lib.wfs_tcn_taps_fwd(_lib.ptr(tab), n_conv, k, _lib.ptr(taps), _lib.ptr(bias), _lib.stream_ptr()))
        y = torch.empty_like(x)
        _lib.check(lib.wfs_tcn_fwd(_lib.ptr(x), N, L, _lib.ptr(taps), _lib.ptr(bias), levels, k, _lib.ptr(y),
                                   _lib.dtype_code(x), float(dropout), _lib.ptr(seed), _lib.stream_ptr()))
        ctx.save_for_backward(x, taps, bias)
        ctx.params = params
        ctx.dropout, ctx.seed, ctx.k, ctx.cache = float(dropout), seed, k, cache
        return y

    @staticmethod
    def backward(ctx, grad_output):
        from ..spconv.functional import grad_like
        lib = _lib.load()
        x, taps, bias = ctx.saved_tensors
        params, k = ctx.params, ctx.k
        N, L = x.shape
        levels = taps.shape[0]
        n_conv = 2 * levels
        dy = grad_output.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dx = torch.empty_like(x)
        partial = torch.empty((N, levels, 2, k + 1), dtype=torch.float32, device=x.device)
        _lib.check(lib.wfs_tcn_bwd(_lib.ptr(x), _lib.ptr(dy), N, L, _lib.ptr(taps), _lib.ptr(bias), levels, k, _lib.ptr(dx),
                                   _lib.ptr(partial), _lib.dtype_code(x), ctx.dropout, _lib.ptr(ctx.seed),
                                   _lib.stream_ptr()))
        grads, rows = [], []
        for c in range(n_conv):
            v, g, b = params[3 * c: 3 * c + 3]
            need = ctx.needs_input_grad[5 + 3 * c: 5 + 3 * c + 3]
            dv = grad_like(v) if need[0] else None
            dg = grad_like(g) if need[1] else None
            db = grad_like(b) if (b is not None and need[2]) else None
            grads += [dv, dg, db]
            rows.append([v.data_ptr(), g.data_ptr(), b.data_ptr() if b is not None else 0,
                         dv.data_ptr() if dv is not None else 0, dg.data_ptr() if dg is not None else 0,
                         db.data_ptr() if db is not None else 0])
        if any(t is not None for t in grads):
            tab = _ptr_table(ctx.cache, ("bwd",) + tuple(x_ for r in rows for x_ in r), rows, x.device)
            _lib.check(lib.wfs_tcn_taps_bwd(_lib.ptr(tab), n_conv, k, _lib.ptr(partial), N, _lib.stream_ptr()))
        return (dx, None, None, None, None) + tuple(grads)


class TemporalConvNet(nn.Module):
    def __init__(self, num_inputs, num_channels, kernel_size=3, dropout=0.2):
        super().__init__()
        blocks = []
        for i, n_out in enumerate(num_channels):
            d = 2 ** i
            n_in = num_inputs if i == 0 else num_channels[i - 1]
            blocks.append(TemporalBlock(n_in, n_out, kernel_size, 1, d, (kernel_size - 1) * d, dropout))
        self.network = nn.Sequential(*blocks)
        self.kernel_size, self.dropout = kernel_size, dropout
        self.single_channel = num_inputs == 1 and all(c == 1 for c in num_channels)

    def _can_fuse(self, x):
        levels, k = len(self.network), self.kernel_size
        if not (self.single_channel and x.is_cuda and x.dim() == 3 and x.shape[1] == 1
                and x.dtype in (torch.float32, torch.bfloat16, torch.float16) and 1 <= levels <= 8 and 1 <= k <= 8
                and 1 <= x.shape[2] <= 4096 and x.shape[0] > 0):
            return False
        if self.training and not 0.0 <= self.dropout < 1.0:
            return False                                  # p = 1 (everything dropped) is torch's business
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            if _lib.load().wfs_tcn_lds_bytes(int(x.shape[2]), levels, 1) > 150 * 1024:
                return False                              # the backward keeps (3 levels + 4) rows in LDS
        return True

    def effective_taps(self):
        """[levels, 2, k] taps after weight norm (w = g v / |v|, differentiable) and [levels, 2] biases."""
        taps, bias = [], []
        for blk in self.network:
            for conv in blk.convs:
                taps.append(torch._weight_norm(conv.weight_v, conv.weight_g, 0).reshape(-1))
                bias.append(conv.bias.reshape(()) if conv.bias is not None else conv.weight_v.new_zeros(()))
        levels = len(self.network)
        return torch.stack(taps).reshape(levels, 2, -1).float(), torch.stack(bias).reshape(levels, 2).float()

    def _norm_params(self):
        """(weight_v, weight_g, bias) of every convolution, level by level -- None unless they are what the fused
        weight-norm kernels read: contiguous fp32 tensors on one device."""
        out = []
        for blk in self.network:
            for conv in blk.convs:
                v, g, b = conv.weight_v, conv.weight_g, conv.bias
                for t in (v, g) + ((b,) if b is not None else ()):
                    if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
                        return None
                out += [v, g, b]
        return out

    def forward(self, x):
        if self._can_fuse(x):
            params = self._norm_params()
            if params is not None:
                rows = x.reshape(x.shape[0], x.shape[2])
                seed, p = None, 0.0
                if self.training and self.dropout > 0:
                    # a fresh 64-bit seed per call from torch's CUDA generator (reproducible under torch.manual_seed, and a
                    # captured graph draws a new one per replay); the kernels derive every mask from it
                    seed = torch.randint(-2 ** 62, 2 ** 62, (1,), dtype=torch.int64, device=x.device)
                    p = self.dropout
                if not hasattr(self, "_ptr_cache"):
                    self._ptr_cache = {}
                return FusedNormedTCNFunction.apply(rows, self.kernel_size, p, seed, self._ptr_cache, *params).reshape(x.shape)
        return self.network(x)

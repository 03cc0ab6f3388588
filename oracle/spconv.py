"""oracle/spconv.py -- TEST INFRASTRUCTURE ONLY.

CPU (torch, fp32) restatement of the spconv 1.2.1 *Python layer* the reference drives
(SURVEY.md Appendix A.1 + A.4): SparseConvTensor, SparseConvolution and its thin subclasses,
SparseSequential, ToDense.  The rulebook comes from oracle/spconv_ref.c (A.3); the per-offset
gather -> torch.mm -> scatter-add loop is executed exactly as spconv's Native algo does on CPU
(A.4), which also makes this the ``cpu_baseline`` ("port") that bench.py times on the host cores.

It is importable as a module whose last name component is ``spconv`` so that the reference's
plugin loader (src/utils/util.py:79-99 ``ModuleUtility``) and its counterpart in this repo key
it exactly like the real package: a config that lists ``"oracle.spconv"`` in ``imports``
resolves ``"spconv.SubMConv3d"`` to the class below.

PARITY UNPINNED against the upstream spconv binary (absent offline; the reference holds no
fixtures for this path).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module.
"""
import math

import numpy as np
import torch
from torch import nn
from torch.autograd import Function

from . import ref as _ref


class SparseConvTensor(object):
    """A.1 container (call site: reference src/models/SPConvNet.py:64)."""

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features = features
        self.indices = indices
        if self.indices.dtype != torch.int32:
            self.indices.int()
        self.spatial_shape = spatial_shape
        self.batch_size = batch_size
        self.indice_dict = {}
        self.grid = grid

    @property
    def spatial_size(self):
        return int(np.prod(self.spatial_shape))

    def find_indice_pair(self, key):
        if key is None:
            return None
        if key in self.indice_dict:
            return self.indice_dict[key]
        return None

    def dense(self, channels_first=True):
        B = int(self.batch_size)
        shape = [B] + [int(s) for s in self.spatial_shape] + [self.features.shape[1]]
        res = torch.zeros(*shape, dtype=self.features.dtype)
        idx = self.indices.long()
        slices = [idx[:, i] for i in range(idx.shape[1])]
        res[tuple(slices)] = self.features          # assignment: duplicates, last wins
        if not channels_first:
            return res
        ndim = len(self.spatial_shape)
        perm = list(range(0, ndim + 1))
        perm.insert(1, ndim + 1)
        return res.permute(*perm).contiguous()

    @property
    def sparity(self):
        return self.indices.shape[0] / np.prod(self.spatial_shape) / self.batch_size


def _native_fwd(features, filters, pairs, pair_num, num_act_out, inverse, subm):
    K = pairs.shape[1]
    W = filters.reshape(K, filters.shape[-2], filters.shape[-1])
    out = features.new_zeros((num_act_out, W.shape[-1]))
    nums = pair_num.tolist()
    kmax = int(np.argmax(np.asarray(nums))) if K > 0 else 0
    if subm:
        torch.mm(features, W[kmax], out=out)
    src, dst = (1, 0) if inverse else (0, 1)
    for k in range(K):
        n = nums[k]
        if n <= 0 or (subm and k == kmax):
            continue
        buf = torch.mm(features[pairs[src, k, :n].long()], W[k])
        out.index_add_(0, pairs[dst, k, :n].long(), buf)
    return out


def _native_bwd(features, filters, grad_out, pairs, pair_num, inverse, subm):
    K = pairs.shape[1]
    W = filters.reshape(K, filters.shape[-2], filters.shape[-1])
    dW = torch.zeros_like(W)
    dX = torch.zeros_like(features)
    nums = pair_num.tolist()
    kmax = int(np.argmax(np.asarray(nums))) if K > 0 else 0
    if subm:
        dW[kmax] = torch.mm(features.t(), grad_out)
        dX = torch.mm(grad_out, W[kmax].t())
    src, dst = (1, 0) if inverse else (0, 1)
    for k in range(K):
        n = nums[k]
        if n <= 0 or (subm and k == kmax):
            continue
        i_in = pairs[src, k, :n].long()
        bi = features[i_in]
        bo = grad_out[pairs[dst, k, :n].long()]
        dW[k] = torch.mm(bi.t(), bo)
        dX.index_add_(0, i_in, torch.mm(bo, W[k].t()))
    return dX, dW.reshape(filters.shape)


class _ConvFunction(Function):
    @staticmethod
    def forward(ctx, features, filters, pairs, pair_num, num_act_out, inverse, subm):
        ctx.save_for_backward(pairs, pair_num, features, filters)
        ctx.flags = (inverse, subm)
        return _native_fwd(features, filters, pairs, pair_num, num_act_out, inverse, subm)

    @staticmethod
    def backward(ctx, grad_output):
        pairs, pair_num, features, filters = ctx.saved_tensors
        inverse, subm = ctx.flags
        dX, dW = _native_bwd(features, filters, grad_output.contiguous(), pairs, pair_num, inverse, subm)
        return dX, dW, None, None, None, None, None


def _listify(v, ndim):
    if isinstance(v, (list, tuple)):
        return [int(x) for x in v]
    return [int(v)] * ndim


class SparseModule(nn.Module):
    pass


class SparseConvolution(SparseModule):
    """A.1 ``SparseConvolution`` (ctor call sites: reference src/models/SPConvBlocks.py:75,134,498,803-810)."""

    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0,
                 dilation=1, groups=1, bias=True, subm=False, output_padding=0, transposed=False,
                 inverse=False, indice_key=None, fused_bn=False, use_hash=False, algo=None):
        super().__init__()
        assert groups == 1
        self.ndim = ndim
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _listify(kernel_size, ndim)
        self.conv1x1 = int(np.prod(self.kernel_size)) == 1
        self.stride = _listify(stride, ndim)
        self.padding = _listify(padding, ndim)
        self.dilation = _listify(dilation, ndim)
        self.output_padding = _listify(output_padding, ndim)
        self.transposed, self.inverse, self.subm = transposed, inverse, subm
        self.groups, self.indice_key = groups, indice_key
        for d, s in zip(self.dilation, self.stride):
            assert any([s == 1, d == 1]), "don't support this."
        self.weight = nn.Parameter(torch.Tensor(*self.kernel_size, in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        features, indices = input.features, input.indices
        spatial_shape, batch_size = input.spatial_shape, int(input.batch_size)
        if not self.subm:
            if self.transposed:
                out_spatial_shape = _ref.deconv_output_shape(spatial_shape, self.kernel_size, self.stride,
                                                             self.padding, self.dilation, self.output_padding)
            else:
                out_spatial_shape = _ref.conv_output_shape(spatial_shape, self.kernel_size, self.stride,
                                                           self.padding, self.dilation)
        else:
            out_spatial_shape = spatial_shape
        if self.conv1x1:
            feats = torch.mm(input.features, self.weight.view(self.in_channels, self.out_channels))
            if self.bias is not None:
                feats = feats + self.bias
            out = SparseConvTensor(feats, input.indices, input.spatial_shape, input.batch_size)
            out.indice_dict = input.indice_dict
            out.grid = input.grid
            return out
        datas = input.find_indice_pair(self.indice_key)
        if self.inverse:
            assert datas is not None and self.indice_key is not None
            _, outids, indice_pairs, indice_pair_num, out_spatial_shape = datas
            assert indice_pair_num.shape[0] == np.prod(self.kernel_size), "inverse conv must have same kernel size as its couple conv"
        else:
            if self.indice_key is not None and datas is not None:
                outids, _, indice_pairs, indice_pair_num, _ = datas
            else:
                o, p, n = _ref.get_indice_pairs(indices.numpy(), batch_size, spatial_shape, self.kernel_size,
                                                self.stride, self.padding, self.dilation,
                                                self.output_padding, self.subm, self.transposed)
                outids, indice_pairs, indice_pair_num = torch.from_numpy(o), torch.from_numpy(p), torch.from_numpy(n)
                input.indice_dict[self.indice_key] = (outids, indices, indice_pairs, indice_pair_num, spatial_shape)
        out_features = _ConvFunction.apply(features, self.weight, indice_pairs, indice_pair_num,
                                           outids.shape[0], bool(self.inverse), bool(self.subm))
        if self.bias is not None:
            out_features = out_features + self.bias
        out = SparseConvTensor(out_features, outids, out_spatial_shape, batch_size)
        out.indice_dict = input.indice_dict
        out.grid = input.grid
        return out


def _mk(name, ndim, **fixed):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None, use_hash=False, algo=None):
        SparseConvolution.__init__(self, ndim, in_channels, out_channels, kernel_size, stride, padding,
                                   dilation, groups, bias, indice_key=indice_key, **fixed)
    return type(name, (SparseConvolution,), {"__init__": __init__})


SparseConv1d = _mk("SparseConv1d", 1)
SparseConv2d = _mk("SparseConv2d", 2)
SparseConv3d = _mk("SparseConv3d", 3)
SparseConv4d = _mk("SparseConv4d", 4)
SubMConv1d = _mk("SubMConv1d", 1, subm=True)
SubMConv2d = _mk("SubMConv2d", 2, subm=True)
SubMConv3d = _mk("SubMConv3d", 3, subm=True)
SubMConv4d = _mk("SubMConv4d", 4, subm=True)


def _mk_inv(name, ndim):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=True, algo=None):
        SparseConvolution.__init__(self, ndim, in_channels, out_channels, kernel_size, bias=bias,
                                   inverse=True, indice_key=indice_key)
    return type(name, (SparseConvolution,), {"__init__": __init__})


SparseInverseConv2d = _mk_inv("SparseInverseConv2d", 2)
SparseInverseConv3d = _mk_inv("SparseInverseConv3d", 3)
SparseConvTranspose2d = _mk("SparseConvTranspose2d", 2, transposed=True)
SparseConvTranspose3d = _mk("SparseConvTranspose3d", 3, transposed=True)


class ToDense(SparseModule):
    def forward(self, x):
        return x.dense()


class SparseSequential(SparseModule):
    """A.1 container: spconv modules get the tensor, plain nn.Modules get ``.features``."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for idx, module in enumerate(args):
            self.add_module(str(idx), module)
        for name, module in kwargs.items():
            self.add_module(name, module)

    def __getitem__(self, idx):
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        self.add_module(name if name is not None else str(len(self._modules)), module)

    def forward(self, input):
        for k, module in self._modules.items():
            if isinstance(module, SparseModule):
                input = module(input)
            else:
                if isinstance(input, SparseConvTensor):
                    if input.indices.shape[0] != 0:
                        input.features = module(input.features)
                else:
                    input = module(input)
        return input

"""``spconv.conv`` counterpart: SparseConvolution and its thin subclasses (spconv 1.2.1,
SURVEY.md A.1; constructed by the reference at src/models/SPConvBlocks.py:75,134,191,249,298,335,
370,498,502,803-810 and from config strings via src/utils/util.py:93-137)."""
import math

import numpy as np
import torch
from torch import nn
from torch.nn import init

from . import functional as Fsp
from . import ops
from .modules import SparseModule
from .tensor import IndiceData, SparseConvTensor


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, subm=False, output_padding=0, transposed=False, inverse=False,
                 indice_key=None, fused_bn=False, use_hash=False, algo=None):
        super(SparseConvolution, self).__init__()
        assert groups == 1
        self.ndim = ndim
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = ops._listify(kernel_size, ndim)
        self.conv1x1 = int(np.prod(self.kernel_size)) == 1
        self.stride = ops._listify(stride, ndim)
        self.padding = ops._listify(padding, ndim)
        self.dilation = ops._listify(dilation, ndim)
        self.transposed = transposed
        self.inverse = inverse
        self.output_padding = ops._listify(output_padding, ndim)
        self.groups = groups
        self.subm = subm
        self.indice_key = indice_key
        self.fused_bn = fused_bn
        self.use_hash = use_hash
        self.algo = algo
        for d, s in zip(self.dilation, self.stride):
            assert any([s == 1, d == 1]), "don't support this."
        self.weight = nn.Parameter(torch.Tensor(*self.kernel_size, in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def _sticky_flags(self):
        """This layer's store of failure flags that rulebook builds only ever set (ops._sticky_flags): plain tensors kept
        across steps and graph replays, not parameters or buffers."""
        store = self.__dict__.get("_flag_store")
        if store is None:
            store = self.__dict__["_flag_store"] = {}
        return store

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        features = input.features
        indices = input.indices
        spatial_shape = input.spatial_shape
        batch_size = input.batch_size
        if self.transposed:
            out_spatial_shape = ops.get_deconv_output_size(spatial_shape, self.kernel_size, self.stride,
                                                           self.padding, self.dilation, self.output_padding)
        elif not self.subm:
            out_spatial_shape = ops.get_conv_output_size(spatial_shape, self.kernel_size, self.stride,
                                                         self.padding, self.dilation)
        else:
            out_spatial_shape = spatial_shape
        if self.conv1x1:
            # spconv: torch.mm(features, weight.view(in, out)) (+ bias); here the same product in libwfsparse (the wide
            # matrix-core path from 128 channels, the gather kernels with the identity map below that)
            features = Fsp.pointwise_conv(features, self.weight, self.bias, input.n_valid)
            out_tensor = SparseConvTensor(features, input.indices, input.spatial_shape, input.batch_size)
            out_tensor.indice_dict = input.indice_dict
            out_tensor.grid = input.grid
            out_tensor.unique = input.unique
            out_tensor.n_valid = input.n_valid
            out_tensor.prefetched = getattr(input, "prefetched", None)
            return out_tensor
        datas = input.find_indice_pair(self.indice_key)
        if self.inverse:
            assert datas is not None and self.indice_key is not None
            rb = datas.rulebook
            out_spatial_shape = datas.spatial_shape
            assert rb.K == np.prod(self.kernel_size), "inverse conv must have same kernel size as its couple conv"
            out_indices = rb.indices
            out_features = Fsp.indice_inverse_conv(features, self.weight, self.bias, rb)
            out_unique = None if rb.has_dup else True
            out_n_valid = rb.n_dev
        else:
            pre = getattr(input, "prefetched", None)
            if pre is not None and id(self) in pre:
                rb = pre[id(self)]                  # built on the side stream by SparseSequential's prefetch
                if rb.ready is not None:
                    rb.ready.wait()                 # modules._SideJoin: the branch's end, waited for once
                self.last_rulebook = rb
                input.unique = not rb.has_dup
                if datas is None:
                    input.indice_dict[self.indice_key] = IndiceData(rb, spatial_shape)
            elif self.indice_key is not None and datas is not None:
                rb = datas.rulebook
            else:
                rb = ops.build_rulebook(indices, batch_size, spatial_shape, self.kernel_size, self.stride,
                                        self.padding, self.dilation, self.subm, known_unique=input.unique,
                                        n_dev=input.n_valid, out_capacity=getattr(self, "out_capacity", None),
                                        transposed=self.transposed, output_padding=self.output_padding,
                                        events=getattr(input, "events", None), flags=self._sticky_flags(),
                                        want_cell_map=getattr(input, "dense_follows", True))
                if getattr(rb, "events_in", None) is not None:
                    input.events = rb.events_in          # the offsets of this row set, for the layers that follow
                self.last_rulebook = rb          # capacity calibration / overflow checks of graph-captured steps
                input.unique = not rb.has_dup
                input.indice_dict[self.indice_key] = IndiceData(rb, spatial_shape)
            out_indices = rb.out_indices
            if self.subm:
                out_features = Fsp.indice_subm_conv(features, self.weight, self.bias, rb)
                out_unique = input.unique
            else:
                out_features = Fsp.indice_conv(features, self.weight, self.bias, rb)
                out_unique = True      # a regular conv numbers DISTINCT output sites
            out_n_valid = rb.m_dev
        out_tensor = SparseConvTensor(out_features, out_indices, out_spatial_shape, batch_size)
        out_tensor.indice_dict = input.indice_dict
        out_tensor.grid = input.grid
        out_tensor.unique = out_unique
        out_tensor.n_valid = out_n_valid
        out_tensor.prefetched = getattr(input, "prefetched", None)
        if self.subm:
            out_tensor.events = getattr(input, "events", None)          # same row set, same event offsets
        elif not self.inverse and getattr(rb, "events_out", None) is not None:
            out_tensor.events = rb.events_out                           # the event-local build numbered them by event
        if not self.subm and not self.inverse:
            out_tensor.cell_map = getattr(rb, "cell_map", None)      # dense() of THIS row set can use the build's map
        return out_tensor


def _conv_class(name, ndim, **fixed):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, indice_key=None, use_hash=False, algo=None):
        SparseConvolution.__init__(self, ndim, in_channels, out_channels, kernel_size, stride, padding, dilation,
                                   groups, bias, indice_key=indice_key, use_hash=use_hash, algo=algo, **fixed)
    return type(name, (SparseConvolution,), {"__init__": __init__, "__module__": __name__})


def _inverse_class(name, ndim):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=True, algo=None):
        SparseConvolution.__init__(self, ndim, in_channels, out_channels, kernel_size, bias=bias, inverse=True,
                                   indice_key=indice_key, algo=algo)
    return type(name, (SparseConvolution,), {"__init__": __init__, "__module__": __name__})


def _transpose_class(name, ndim):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, indice_key=None, algo=None):
        SparseConvolution.__init__(self, ndim, in_channels, out_channels, kernel_size, stride, padding, dilation,
                                   groups, bias, transposed=True, indice_key=indice_key, algo=algo)
    return type(name, (SparseConvolution,), {"__init__": __init__, "__module__": __name__})


SparseConv1d = _conv_class("SparseConv1d", 1)
SparseConv2d = _conv_class("SparseConv2d", 2)
SparseConv3d = _conv_class("SparseConv3d", 3)
SparseConv4d = _conv_class("SparseConv4d", 4)
SubMConv1d = _conv_class("SubMConv1d", 1, subm=True)
SubMConv2d = _conv_class("SubMConv2d", 2, subm=True)
SubMConv3d = _conv_class("SubMConv3d", 3, subm=True)
SubMConv4d = _conv_class("SubMConv4d", 4, subm=True)
SparseInverseConv2d = _inverse_class("SparseInverseConv2d", 2)
SparseInverseConv3d = _inverse_class("SparseInverseConv3d", 3)
SparseConvTranspose2d = _transpose_class("SparseConvTranspose2d", 2)
SparseConvTranspose3d = _transpose_class("SparseConvTranspose3d", 3)

"""``spconv.modules`` counterpart: SparseModule, SparseSequential, ToDense (spconv 1.2.1,
SURVEY.md A.1; reference call sites src/models/SPConvBlocks.py:81-82,515-516,822)."""
from collections import OrderedDict

import torch
from torch import nn


class SparseModule(nn.Module):
    """place holder, all module subclass from this will take sptensor in SparseSequential."""
    pass


def _is_sparse_tensor(x):
    from .tensor import SparseConvTensor
    return isinstance(x, SparseConvTensor)


class SparseSequential(SparseModule):
    """Sequential container: spconv modules receive the SparseConvTensor, plain ``nn.Module``s are
    applied to ``.features`` (skipped when there are no active rows); after ``ToDense`` the value
    is a dense tensor and later modules receive it directly."""

    def __init__(self, *args, **kwargs):
        super(SparseSequential, self).__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError('index {} is out of range'.format(idx))
        if idx < 0:
            idx += len(self)
        it = iter(self._modules.values())
        for i in range(idx):
            next(it)
        return next(it)

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input):
        from . import functional as Fsp
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            module = mods[i]
            if isinstance(module, SparseModule):
                input = module(input)
            elif _is_sparse_tensor(input):
                if input.indices.shape[0] != 0:
                    if isinstance(module, nn.BatchNorm1d) and Fsp.can_fuse_batch_norm(module, input.features):
                        # BatchNorm1d [+ ReLU] over the active rows: one fused pair of HIP launches
                        relu = i + 1 < len(mods) and type(mods[i + 1]) is nn.ReLU
                        input.features = Fsp.batch_norm_relu(input.features, module, relu, input.n_valid)
                        if relu:
                            i += 1
                    else:
                        if input.n_valid is not None and type(module) is not nn.ReLU:      # ReLU is row-local
                            raise RuntimeError("device-count (padded) SparseConvTensors only support the fused "
                                               "BatchNorm1d/ReLU pair between sparse layers, got %s" % type(module).__name__)
                        input.features = module(input.features)
            else:
                input = module(input)
            i += 1
        return input


class ToDense(SparseModule):
    """convert SparseConvTensor to NCHW dense tensor."""

    def forward(self, x):
        return x.dense()


class RemoveGrid(SparseModule):
    """remove pre-allocated grid buffer."""

    def forward(self, x):
        x.grid = None
        return x

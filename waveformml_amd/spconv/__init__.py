"""MI355X-native counterpart of the ``spconv`` 1.2.1 Python surface that WaveformML drives
(reference: ``import spconv`` at src/models/SPConvBlocks.py:4, src/engineering/LitBase.py:5 and
config strings such as "spconv.SubMConv3d" resolved by src/utils/util.py:74-137).

Same names, argument meaning and error behaviour; the arithmetic is libwfsparse.so (HIP, gfx950).
There is no CPU fallback: CPU tensors raise RuntimeError.

To let unmodified reference code ``import spconv`` pick this package up, put
``waveformml_amd/compat`` on PYTHONPATH (it holds a one-line ``spconv`` alias) or call
``waveformml_amd.install_as_spconv()`` before the reference's modules are imported.
"""
from . import functional, ops
from .conv import (SparseConv1d, SparseConv2d, SparseConv3d, SparseConv4d, SparseConvolution,
                   SparseConvTranspose2d, SparseConvTranspose3d, SparseInverseConv2d, SparseInverseConv3d,
                   SubMConv1d, SubMConv2d, SubMConv3d, SubMConv4d)
from .modules import RemoveGrid, SparseModule, SparseSequential, ToDense
from .tensor import IndiceData, SparseConvTensor

__version__ = "1.2.1+wfsparse"

__all__ = ["SparseConvTensor", "SparseConvolution", "SparseConv1d", "SparseConv2d", "SparseConv3d",
           "SparseConv4d", "SubMConv1d", "SubMConv2d", "SubMConv3d", "SubMConv4d", "SparseInverseConv2d",
           "SparseInverseConv3d", "SparseConvTranspose2d", "SparseConvTranspose3d", "SparseSequential",
           "SparseModule", "ToDense", "RemoveGrid", "ops", "functional", "IndiceData"]

"""CPU checks of the drop-in boundary: libwfsparse.so loads and exports every symbol that
include/wfsparse.h declares, the ctypes table mirrors the header, and the host-side front door
(geometry validation) behaves like spconv's.  No kernel is launched here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions(header="wfsparse.h", prefix="wfs_"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s[a-z0-9_]+)\s*\(" % prefix, text)))


def test_library_exports_every_declared_symbol():
    from waveformml_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _header_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), "libwfsparse.so does not export %s" % n


def test_ctypes_table_mirrors_header():
    from waveformml_amd import _lib
    assert sorted(_lib.SIGNATURES) == _header_functions()
    assert _lib.load().wfs_abi_version() == _lib.WFS_ABI_VERSION == 6


def test_h5_reader_library_exports_and_ctypes_table_mirror_its_header():
    from waveformml_amd.psd import h5data
    names = _header_functions("wfh5.h", "wfh5_")
    assert len(names) == 11 and sorted(h5data.SIGNATURES) == names
    lib = ctypes.CDLL(h5data.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libwfh5.so does not export %s" % n
    assert ctypes.sizeof(h5data.Info) == 40          # struct wfh5_info: 3 x int64 + 4 x int32


def test_geometry_front_door():
    from waveformml_amd import _lib
    g = _lib.make_geometry(3, 256, [14, 11, 256], [3, 3, 3], [1, 1, 4], [0, 0, 0], [1, 1, 1], False)
    assert [g.out_shape[i] for i in range(3)] == [12, 9, 64] and g.K == 27
    g = _lib.make_geometry(3, 256, [14, 11, 256], [3, 3, 3], [2, 2, 2], [5, 5, 5], [1, 1, 1], True)
    assert [g.out_shape[i] for i in range(3)] == [14, 11, 256]          # SubM ignores stride / padding (A.2)
    assert [g.padding[i] for i in range(3)] == [1, 1, 1] and [g.stride[i] for i in range(3)] == [1, 1, 1]
    with pytest.raises(RuntimeError, match="2\\^31"):
        _lib.make_geometry(3, 4096, [1024, 1024, 1024], [3] * 3, [1] * 3, [0] * 3, [1] * 3, True)
    with pytest.raises(RuntimeError, match="stride>1 together with dilation>1"):
        _lib.make_geometry(2, 1, [8, 8], [3, 3], [2, 2], [0, 0], [2, 2], False)
    assert _lib.load().wfs_rulebook_workspace_bytes(ctypes.byref(g), 1000) > 0


def test_surface_matches_what_the_reference_calls():
    """Names / positional signatures the reference uses."""
    import torch
    import waveformml_amd.spconv as sp
    for name in ["SparseConvTensor", "SparseConv1d", "SparseConv2d", "SparseConv3d", "SparseConv4d", "SubMConv2d",
                 "SubMConv3d", "SparseInverseConv2d", "SparseConvTranspose2d", "SparseConvTranspose3d",
                 "SparseSequential", "ToDense"]:
        assert hasattr(sp, name), name
    # reference src/models/SPConvBlocks.py:498 passes 8 positionals; the 8th lands in `bias`
    layer = sp.SparseConv2d(300, 252, 1, 1, 0, 1, 1, False)
    assert layer.bias is None and layer.conv1x1 and tuple(layer.weight.shape) == (1, 1, 300, 252)
    assert isinstance(layer, sp.SparseConv2d) and layer.in_channels == 300 and layer.kernel_size == [1, 1]
    inv = sp.SparseInverseConv2d(8, 4, 3, "ind_0", bias=False)
    assert inv.inverse and inv.indice_key == "ind_0" and tuple(inv.weight.shape) == (3, 3, 8, 4)
    with pytest.raises(AssertionError):
        sp.SparseConv2d(4, 4, 3, 2, 0, 2)
    with pytest.raises(AssertionError):
        sp.SparseConv2d(4, 4, 3, groups=2)
    seq = sp.SparseSequential(sp.SubMConv3d(2, 4, 3, indice_key="k"), torch.nn.ReLU(), sp.ToDense())
    assert len(seq) == 3 and isinstance(seq[0], sp.SubMConv3d)
    # weights use torch's generic fan computation on [*k, Cin, Cout] (A.1)
    w = sp.SubMConv3d(2, 32, 3).weight
    assert float(w.detach().abs().max()) <= 1.0 / 24 + 1e-7      # fan_in = 3 * (3*2*32) = 576
    # no CPU path
    x = sp.SparseConvTensor(torch.zeros(3, 2), torch.zeros(3, 4, dtype=torch.int32), [4, 4, 4], torch.tensor(1))
    assert x.batch_size == 1
    with pytest.raises(RuntimeError):
        seq(x)


def test_graft_entry_build_runs():
    """The driver's build check: __graft_entry__.build() (incremental make of both libraries and the oracle, import of
    the package, ABI version of library and binding equal) must not raise."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    entry = importlib.import_module("__graft_entry__")
    entry.build()

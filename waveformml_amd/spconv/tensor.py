"""``spconv.SparseConvTensor`` counterpart (spconv 1.2.1; reference call sites
src/models/SPConvNet.py:23-25,64, src/engineering/LitBase.py:138-146)."""
import numpy as np
import torch

from . import functional as Fsp


class IndiceData(object):
    """What spconv caches in ``indice_dict[key]``: the 5-tuple
    ``(outids, indices, indice_pairs, indice_pair_num, spatial_shape)``.

    It unpacks / indexes like that tuple; ``indice_pairs`` and ``indice_pair_num`` (spconv's own
    rulebook encoding) are produced on first access, the kernels use ``.rulebook`` directly."""

    def __init__(self, rulebook, spatial_shape):
        self.rulebook = rulebook
        self.spatial_shape = spatial_shape

    def _item(self, i):
        rb = self.rulebook
        if i == 0:
            return rb.out_indices
        if i == 1:
            return rb.indices
        if i == 2:
            return rb.indice_pairs
        if i == 3:
            return rb.indice_pair_num
        if i == 4:
            return self.spatial_shape
        raise IndexError(i)

    def __len__(self):
        return 5

    def __getitem__(self, i):
        if isinstance(i, slice):
            return tuple(self._item(j) for j in range(*i.indices(5)))
        return self._item(i + 5 if i < 0 else i)

    def __iter__(self):
        return (self._item(i) for i in range(5))


class SparseConvTensor(object):
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        """
        Args:
            features: [num_points, num_features] float32 / bfloat16 tensor on the GPU
            indices: [num_points, ndim + 1] int32, batch index first
            spatial_shape: spatial shape of the dense grid
            batch_size: int (the reference hands in a 0-dim tensor, src/models/SPConvNet.py:63)
            grid: unused, kept for signature compatibility
        """
        self._features = features
        self._pending = None        # functional.RowAffine: BatchNorm (+ ReLU) still to be applied to the raw rows
        self.bn_link = None         # functional.BnLink of the fused BatchNorm that produced `features`, if any
        self.indices = indices
        if self.indices.dtype != torch.int32:
            self.indices = self.indices.int()
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {}
        self.grid = grid
        self.unique = None          # True/False once a rulebook build has looked; None = unknown
        # device-count mode: `features`/`indices` hold a CAPACITY of rows, the first n_valid[0] are real
        # (int64 [1] device tensor).  None = every row is valid (spconv's normal contract).
        self.n_valid = None

    # ``features`` is what spconv exposes.  Inside a SparseSequential a training-mode BatchNorm1d (+ ReLU) that follows
    # a conv may be DEFERRED to the next reader of the rows (functional.RowAffine: the next conv / dense() applies it
    # while gathering); whoever asks for ``.features`` gets the normalised rows (materialised on first access).
    @property
    def features(self):
        if self._pending is not None:
            spec, self._pending = self._pending, None
            self._features = Fsp.materialize_affine(self._features, spec, self.n_valid)
        return self._features

    @features.setter
    def features(self, value):
        self._features = value
        self._pending = None
        self.bn_link = None         # whoever assigns new rows says if a BatchNorm produced them (SparseSequential)

    def defer_affine(self, spec):
        """Keep the raw rows and remember the BatchNorm (+ ReLU) the next reader has to apply."""
        assert self._pending is None
        self._pending = spec

    def take_pending(self):
        """(raw rows, RowAffine or None) for a reader that applies the map itself; clears the deferral."""
        spec, self._pending = self._pending, None
        return self._features, spec

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
        cell_map = getattr(self, "cell_map", None)
        if (self._pending is not None and cell_map is not None and self._features.shape[0] > 0
                and (self.unique is True or self.n_valid is not None)
                and Fsp._dense_map_ok(cell_map, self.spatial_shape, int(self.batch_size), self._features.shape[1],
                                      self._features)):
            raw, spec = self.take_pending()
            out = Fsp.affine_to_dense(raw, spec, cell_map, self.spatial_shape, self.batch_size, self.n_valid)
            self._features, self._pending = raw, spec          # the tensor itself still stands for the normalised rows
        else:
            out = Fsp.to_dense(self.features, self.indices, self.spatial_shape, self.batch_size,
                               self.unique is True or self.n_valid is not None, self.n_valid, cell_map)
        if channels_first:
            return out
        ndim = len(self.spatial_shape)
        return out.permute(0, *range(2, ndim + 2), 1).contiguous()

    @property
    def sparity(self):
        return self.indices.shape[0] / np.prod(self.spatial_shape) / self.batch_size

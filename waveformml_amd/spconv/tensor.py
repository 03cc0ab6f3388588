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
        self.features = features
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
        out = Fsp.to_dense(self.features, self.indices, self.spatial_shape, self.batch_size,
                           self.unique is True or self.n_valid is not None, self.n_valid, cell_map)
        if channels_first:
            return out
        ndim = len(self.spatial_shape)
        return out.permute(0, *range(2, ndim + 2), 1).contiguous()

    @property
    def sparity(self):
        return self.indices.shape[0] / np.prod(self.spatial_shape) / self.batch_size

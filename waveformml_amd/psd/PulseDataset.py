"""Config-facing dataset classes: the constructors the reference's data module calls,
``dataset_class(config, dataset_type, n_per_dir, device, **dataset_params)`` (reference
src/engineering/PSDDataModule.py:52-57), for ``"dataset_class": "PulseDataset.PulseDataset2D"`` /
``"PulseDataset.PulseDataset3D"`` with ``"waveformml_amd.psd.PulseDataset"`` in ``dataset_config.imports``
(reference config/examples/GEP.json:81-86 names ``src.datasets.PulseDataset``).

What they do is reference src/datasets/PulseDataset.py:88-131 + :543-621: directories =
``dataset_config.base_path`` joined with each of ``dataset_config.paths`` (one class label per directory), features
normalised by 1 / (2^14 - 1), file mask / table / batch column fixed per layout; the reading itself is the native
chunk-parallel reader behind psd/h5data.py (libwfh5).  The reference's offline preparation (``write_shuffled``, chunked
re-writes, ``retrieve_config`` of a pickled file list: SURVEY.md 3.3, not on the timed path) is not provided.
"""
import os

from . import h5data


def _dirs(config):
    dc = config.dataset_config
    return [os.path.join(dc.base_path, p) for p in dc.paths]


class _ConfigMixin(object):
    def _init_from_config(self, base, config, dataset_type, n_per_dir, device, file_excludes, label_name,
                          label_file_pattern, data_cache_size, use_half, normalize=True, label_map=None):
        self.config = config.dataset_config
        self.dataset_type = dataset_type                      # "train" | "validate" | "test" (bookkeeping only)
        self.n_paths = self.n_categories = len(self.config.paths)
        self.use_half = use_half
        base.__init__(self, _dirs(config), n_per_dir, device, file_excludes=file_excludes, label_name=label_name,
                      label_file_pattern=label_file_pattern, data_cache_size=data_cache_size, normalize=normalize,
                      use_half=use_half, label_map=label_map)

    def get_file_list(self):
        """Files this dataset draws from -- what the data module passes to the next split as ``file_excludes``
        (reference src/datasets/HDF5Dataset.py get_file_list, used at PSDDataModule.py:58,91-93,110-113)."""
        return list(self.ordered_file_set)

    @classmethod
    def retrieve_config(cls, config_path, device, use_half=False):
        raise NotImplementedError("datasets restored from a saved file list (train_config / val_config) belong to the "
                                  "reference's offline preparation; build the dataset from dataset_config.paths")


class PulseDataset2D(_ConfigMixin, h5data.PulseDataset2D):
    """[N, 2 * nsamples] rows, N = PMT pairs fired in the item's events (reference PulseDataset.py:543-579)."""

    def __init__(self, config, dataset_type, n_per_dir, device, file_excludes=None, label_name=None,
                 label_file_pattern=None, data_cache_size=3, model_dir=None, data_dir=None, dataset_dir=None,
                 use_half=False):
        self._init_from_config(h5data.PulseDataset2D, config, dataset_type, n_per_dir, device, file_excludes, label_name,
                               label_file_pattern, data_cache_size, use_half)


class PulseDataset3D(_ConfigMixin, h5data.PulseDataset3D):
    """[N, 2] rows, N = active (cell, sample) voxels of the item's events (reference PulseDataset.py:582-621)."""

    def __init__(self, config, dataset_type, n_per_dir, device, file_excludes=None, label_name=None,
                 label_file_pattern=None, data_cache_size=3, model_dir=None, data_dir=None, dataset_dir=None,
                 use_half=False):
        self._init_from_config(h5data.PulseDataset3D, config, dataset_type, n_per_dir, device, file_excludes, label_name,
                               label_file_pattern, data_cache_size, use_half)

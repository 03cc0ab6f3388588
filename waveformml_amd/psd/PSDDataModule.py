"""``PSDDataModule(config, device)``: the data module the reference's main.py hands to its trainer
(reference src/engineering/PSDDataModule.py:22-151), without the pytorch_lightning base class: same constructor, same
``setup`` / ``train_dataloader`` / ``val_dataloader`` / ``test_dataloader``, same config keys
(``dataset_config.{imports, dataset_class, dataset_params, test_dataset_params, dataloader_params, paths, base_path,
n_train, n_validate, n_test}``, ``system_config.half_precision``), same split logic: the validation set excludes the
training set's files, the test set excludes both.  Batches are built by the reference's collate (psd/data.py).
One deliberate difference: the reference passes ``collate_fn`` to every loader, which offsets coordinate column 2 even
for the 4-column 3-D layout, where column 2 is the TIME sample and the event id sits in column 3;
here 3-D datasets are collated on column 3 (``collate_fn_3d``), 2-D datasets exactly as the reference does.

Out of scope here (offline preparation): ``data_prep: "shuffle"`` re-writes and datasets restored from
``train_config`` / ``val_config`` / ``test_config`` file lists -- both raise.
"""
import logging

from torch.utils.data import DataLoader

from .config import DictionaryUtility, ModuleUtility
from .data import PackedLoader, collate_fn, collate_fn_3d, rank_sampler  # noqa: F401  (collate_fn is the reference's name)


class PSDDataModule(object):
    def __init__(self, config, device):
        self.log = logging.getLogger(__name__)
        self.config = config
        self.device = device
        dc = config.dataset_config
        if hasattr(config.system_config, "half_precision"):
            self.half_precision = config.system_config.half_precision
            if not hasattr(dc, "dataset_params"):
                setattr(dc, "dataset_params", DictionaryUtility.to_object({}))
            if not hasattr(dc.dataset_params, "use_half"):
                setattr(dc.dataset_params, "use_half", bool(self.half_precision))
        else:
            self.half_precision = False
        for key in ("train_config", "val_config", "test_config"):
            if hasattr(dc, key):
                raise NotImplementedError("dataset_config.%s (a saved file list) belongs to the reference's offline "
                                          "preparation; give dataset_config.paths" % key)
        if getattr(dc, "data_prep", None) == "shuffle":
            raise NotImplementedError("data_prep = 'shuffle' (offline re-write of the files) is not provided")
        self.ntype = len(dc.paths)
        self.total_train = dc.n_train * self.ntype
        self.modules = ModuleUtility(dc.imports)
        self.dataset_class = self.modules.retrieve_class(dc.dataset_class)

    def _params(self, key="dataset_params"):
        dc = self.config.dataset_config
        return DictionaryUtility.to_dict(getattr(dc, key)) if hasattr(dc, key) else {}

    def prepare_data(self):
        pass

    def gen_train_dataset(self):
        if not hasattr(self, "train_dataset"):
            self.train_dataset = self.dataset_class(self.config, "train", self.config.dataset_config.n_train,
                                                    self.device, **self._params())
            self.train_excludes = self.train_dataset.get_file_list()

    def setup(self, stage=None):
        dc = self.config.dataset_config
        if stage in ("fit", "train", None):
            self.gen_train_dataset()
        if stage in ("test", None):
            self.gen_train_dataset()
            if not hasattr(self, "val_dataset"):
                n_validate = dc.n_validate if hasattr(dc, "n_validate") else dc.n_test
                self.val_dataset = self.dataset_class(self.config, "validate", n_validate, self.device,
                                                      file_excludes=self.train_excludes, **self._params())
            if not hasattr(self, "test_dataset"):
                excludes = self.train_excludes + self.val_dataset.get_file_list()
                key = "test_dataset_params" if hasattr(dc, "test_dataset_params") else "dataset_params"
                self.test_dataset = self.dataset_class(self.config, "test", dc.n_test, self.device,
                                                       file_excludes=excludes, **self._params(key))

    def _collate(self, dataset):
        return collate_fn_3d if getattr(dataset, "layout", "2d") == "3d" else collate_fn

    def _loader(self, dataset, shuffle):
        # one process per GPU: each rank reads its own 1/N share of the items, as under the reference's Lightning DDP
        # (which replaces the loaders' samplers with DistributedSamplers, src/utils/util.py:228-239)
        sampler = rank_sampler(dataset, shuffle)
        params = self._params("dataloader_params")
        if params.get("num_workers", 0) > 0 and getattr(self.config.dataset_config, "pack_batches", True):
            # worker processes hand a batch over as ONE shared-memory buffer (psd/data.PackedLoader); same batches
            return PackedLoader(dataset, self._collate(dataset), shuffle=shuffle and sampler is None, sampler=sampler,
                                **params)
        return DataLoader(dataset, shuffle=shuffle and sampler is None, sampler=sampler,
                          collate_fn=self._collate(dataset), **params)

    def train_dataloader(self):
        if not hasattr(self, "train_dataset"):
            self.setup("train")
        return self._loader(self.train_dataset, True)

    def val_dataloader(self):
        if not hasattr(self, "val_dataset"):
            self.setup("test")
        return self._loader(self.val_dataset, False)

    def test_dataloader(self):
        if not hasattr(self, "test_dataset"):
            self.setup("test")
        return self._loader(self.test_dataset, False)

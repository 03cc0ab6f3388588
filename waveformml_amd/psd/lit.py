"""``LitPSD``: host-side mirror of the reference's LightningModule for PSD classification
(src/engineering/LitBase.py:13-55, src/engineering/LitPSD.py:20-151).

pytorch_lightning 1.2 (which the reference subclasses) is not installable here, so this class keeps
the same constructor, attributes (``model``, ``criterion``, ``lr``, ``modules``) and step semantics on
a plain ``nn.Module``; ``waveformml_amd.psd.trainer.Trainer`` plays the part of ``pl.Trainer`` for it.
Where Lightning IS present, INTEGRATION.md shows the two-line change that makes the reference's own
LitPSD use the MI355X operators instead (only the ``imports`` list changes).
"""
import logging

import torch
from torch import nn

from .config import DictionaryUtility, ModuleUtility


class LitPSD(nn.Module):
    def __init__(self, config, trial=None):
        super().__init__()
        self.trial = trial
        self.pylog = logging.getLogger(__name__)
        self.config = config
        self.n_type = config.system_config.n_type
        self.lr = config.optimize_config.lr
        self.modules_util = ModuleUtility(config.net_config.imports + config.dataset_config.imports +
                                          config.optimize_config.imports)
        self.model = self.modules_util.retrieve_class(config.net_config.net_class)(config)
        criterion_class = self.modules_util.retrieve_class(config.net_config.criterion_class)
        self.criterion = criterion_class(*config.net_config.criterion_params, reduction="mean")
        self.occlude_index = getattr(config.dataset_config, "occlude_index", None)
        self.softmax = nn.LogSoftmax(dim=1)
        self.logged = {}
        self.optimizer_parameters = None      # set to [flat parameter] by ddp.FlatGradAllReducer(flatten=True)

    def forward(self, x):
        return self.model(x)

    def _predict(self, c, f, target, n_valid=None):
        # the reference reads the batch size back from the last coordinate row (SPConvNet.py:63, a device->host
        # sync); one label per event means it equals len(target), which is known on the host
        if hasattr(self.model, "batch_size_hint"):
            self.model.batch_size_hint = int(target.shape[0])
        return self.model([c, f, n_valid] if n_valid is not None else [c, f])

    @staticmethod
    def _unpack(batch):
        """((coords, feats), target) as the reference's collate_fn builds it; a third element of the inner
        list is the device-side count of valid rows of a capacity-padded batch (psd/graph.py)."""
        inputs, target = batch
        return inputs[0], inputs[1], (inputs[2] if len(inputs) > 2 else None), target

    def log(self, name, value, **kwargs):
        self.logged[name] = value.detach() if torch.is_tensor(value) else value

    def log_dict(self, d, **kwargs):
        for k, v in d.items():
            self.log(k, v)

    # reference LitPSD.configure_optimizers, :60-76
    def configure_optimizers(self):
        oc = self.config.optimize_config
        params = self.optimizer_parameters if self.optimizer_parameters is not None else self.model.parameters()
        kwargs = DictionaryUtility.to_dict(oc.optimizer_params)
        opt_class = self.modules_util.retrieve_class(oc.optimizer_class)
        if self.optimizer_parameters is not None and len(self.optimizer_parameters) == 1:
            # one flat tensor: torch's multi-tensor ("foreach") kernels would run it on a handful of blocks (and so
            # does its "fused" SGD: measured 74 us against 19 us for the four single-tensor launches)
            import inspect
            if opt_class is torch.optim.SGD and all(p.is_cuda for p in self.optimizer_parameters):
                from .optim import FlatSGD
                opt_class = FlatSGD               # the whole update in one HIP launch, lr in device memory
            elif "foreach" in inspect.signature(opt_class.__init__).parameters and "foreach" not in kwargs:
                kwargs["foreach"] = False
        optimizer = opt_class(params, lr=self.lr, **kwargs)
        if getattr(oc, "scheduler_class", None):
            if not hasattr(oc, "scheduler_params"):
                raise IOError("Optimizer config has a learning scheduler class specified. You must also set "
                              "lr_schedule_parameters (dictionary of key value pairs).")
            scheduler = self.modules_util.retrieve_class(oc.scheduler_class)(
                optimizer, **DictionaryUtility.to_dict(oc.scheduler_params))
            return [optimizer], [scheduler]
        return optimizer

    def _loss(self, predictions, target):
        """``self.criterion.forward`` -- through the one-launch HIP kernel when the criterion is a plain
        CrossEntropyLoss(mean) on GPU logits (same value; torch needs six launches for loss + gradient)."""
        if predictions.is_cuda:
            from ..spconv import functional as Fsp
            if Fsp.can_fuse_cross_entropy(self.criterion, predictions, target):
                return Fsp.cross_entropy_mean(predictions, target, self.criterion.ignore_index)
        return self.criterion.forward(predictions, target)

    # reference LitPSD.training_step, :94-104
    def training_step(self, batch, batch_idx):
        c, f, n_valid, target = self._unpack(batch)
        predictions = self._predict(c, f, target, n_valid)
        loss = self._loss(predictions, target)
        self.log("train_loss", loss, on_epoch=True, prog_bar=True, logger=True)
        return loss

    # reference LitPSD.validation_step, :106-128
    def validation_step(self, batch, batch_idx):
        c, f, n_valid, target = self._unpack(batch)
        predictions = self._predict(c, f, target, n_valid)
        loss = self.criterion.forward(predictions, target)
        pred = torch.argmax(self.softmax(predictions), dim=1)
        acc = (pred == target).float().mean()
        results = {"val_loss": loss, "val_acc": acc}
        self.log_dict(results, on_epoch=True, prog_bar=True, logger=True)
        return results

    # reference LitPSD.test_step, :130-151 (evaluator plumbing is out of scope)
    def test_step(self, batch, batch_idx):
        c, f, n_valid, target = self._unpack(batch)
        if self.occlude_index:                       # falsy for index 0, exactly as the reference (:134)
            f[:, self.occlude_index] = 0
        predictions = self._predict(c, f, target, n_valid)
        loss = self.criterion.forward(predictions, target)
        pred = torch.argmax(self.softmax(predictions), dim=1)
        acc = (pred == target).float().mean()
        results = {"test_loss": loss, "test_acc": acc}
        self.log_dict(results, on_epoch=True, logger=True)
        return results

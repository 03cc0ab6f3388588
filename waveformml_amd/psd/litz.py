"""``LitZ``: host-side mirror of the reference's per-segment regression module (src/engineering/LitZ.py:31-139 on
src/engineering/LitBase.py:13-55,124-174): ``SingleEndedZConv`` predicts a dense [B, 1, 14, 11] map, the loss compares
it with the targets of the ACTIVE segments only -- both the mask and the dense target come from
``SparseConvTensor(...).dense()`` (wfs_to_dense) -- with a sum-reduced criterion divided by the number of rows.

As psd/lit.LitPSD this is a plain ``nn.Module`` with Lightning's step methods (Lightning is not installable offline);
the evaluators (ZEvaluator*, histogram / plot plumbing) are out of scope, SURVEY.md 2.
"""
import logging

import torch
from torch import nn

from .config import DictionaryUtility, ModuleUtility
from .znet import SingleEndedZConv


class LitZ(nn.Module):
    def __init__(self, config, trial=None):
        super().__init__()
        self.trial = trial
        self.pylog = logging.getLogger(__name__)
        self.config = config
        self.nx, self.ny = 14, 11
        self.lr = config.optimize_config.lr
        self.modules_util = ModuleUtility(config.net_config.imports + config.dataset_config.imports +
                                          config.optimize_config.imports)
        self.model = SingleEndedZConv(config)
        criterion_class = self.modules_util.retrieve_class(config.net_config.criterion_class)
        self.criterion = criterion_class(*config.net_config.criterion_params, reduction="sum")   # event_predictions=False
        self.occlude_index = getattr(config.dataset_config, "occlude_index", None)
        if getattr(config.net_config, "SELoss", False):
            raise NotImplementedError("the single-ended-only loss needs the evaluator's segment status table")
        if hasattr(config.net_config, "UseFFT"):
            raise NotImplementedError("UseFFT feeds complex features; not on the mirrored path")
        self.logged = {}

    def forward(self, x):
        return self.model(x)

    def log(self, name, value, **kwargs):
        self.logged[name] = value.detach() if torch.is_tensor(value) else value

    def log_dict(self, d, **kwargs):
        for k, v in d.items():
            self.log(k, v)

    def configure_optimizers(self):
        oc = self.config.optimize_config
        optimizer = self.modules_util.retrieve_class(oc.optimizer_class)(
            self.model.parameters(), lr=self.lr, **DictionaryUtility.to_dict(oc.optimizer_params))
        if getattr(oc, "scheduler_class", None):
            if not hasattr(oc, "scheduler_params"):
                raise IOError("Optimizer config has a learning scheduler class specified. You must also set "
                              "lr_schedule_parameters (dictionary of key value pairs).")
            scheduler = self.modules_util.retrieve_class(oc.scheduler_class)(
                optimizer, **DictionaryUtility.to_dict(oc.scheduler_params))
            return [optimizer], [scheduler]
        return optimizer

    # reference LitBase._calc_segment_loss, :124-174 (SE_only branch excluded)
    def _calc_segment_loss(self, coo, predictions, target, use_float=True, target_index=None, sparse_mask=None):
        sp = self.model.spconv
        batch_size = int(coo[-1, -1]) + 1
        num_predictions = coo.shape[0]
        if target.shape[0] != num_predictions:
            raise ValueError("if using segment loss, target must have same number of elements in first dimension as "
                             "coordinate tensor")
        idx = coo[:, self.model.permute_tensor].contiguous()
        if sparse_mask is None:
            ones = torch.ones((num_predictions, predictions.shape[1]), dtype=torch.float32, device=predictions.device)
            sparse_mask = sp.SparseConvTensor(ones, idx, self.model.spatial_size, batch_size).dense()
        t = target.unsqueeze(1) if target.dim() == 1 else target
        target_tensor = sp.SparseConvTensor(t, idx, self.model.spatial_size, batch_size).dense()
        predictions = sparse_mask * predictions
        if target_index is None:
            want = target_tensor if use_float else target_tensor.squeeze(1)
        else:
            want = target_tensor[:, target_index, :, :]
            want = want.unsqueeze(1) if use_float else want
        loss = self.criterion.forward(predictions, want)
        return loss / num_predictions, target_tensor, predictions, sparse_mask

    # reference LitZ._process_batch, :89-108
    def _process_batch(self, batch, target_index=None):
        (c, f), target = batch
        additional_fields = None
        if isinstance(f, list):
            additional_fields, f = f[1:], f[0]
        if self.occlude_index:
            f[:, self.occlude_index] = 0
        predictions = self.model([c, f])
        loss, target_tensor, predictions, _ = self._calc_segment_loss(c, predictions, target, target_index=target_index)
        return loss, predictions, target_tensor, c, f, additional_fields

    def training_step(self, batch, batch_idx):
        loss = self._process_batch(batch)[0]
        self.log("train_loss", loss, on_epoch=True, prog_bar=True, logger=True)
        return loss

    def validation_step(self, batch, batch_idx):
        loss = self._process_batch(batch)[0]
        self.log("val_loss", loss, on_epoch=True, prog_bar=True, logger=True)
        return loss

    def test_step(self, batch, batch_idx):
        loss = self._process_batch(batch)[0]
        results = {"test_loss": loss}
        self.log_dict(results, on_epoch=True, logger=True)
        return results

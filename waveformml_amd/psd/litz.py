"""``LitZ`` and ``LitEZ``: host-side mirrors of the reference's per-segment regression modules
(src/engineering/LitZ.py:31-139, src/engineering/LitEZ.py:8-91, on src/engineering/LitBase.py:13-55,110-174): the net
predicts a dense [B, out, 14, 11] map, the loss compares it with the targets of the ACTIVE segments only -- both the mask
and the dense target come from ``SparseConvTensor(...).dense()`` (wfs_to_dense) -- with a sum-reduced criterion divided by
the number of rows (``net_config.SELoss``: of the single-ended segments only, psd/segments.py).

As psd/lit.LitPSD these are plain ``nn.Module``s with Lightning's step methods (Lightning is not installable offline);
the evaluators (ZEvaluator*, EZEvaluator*, histogram / plot plumbing) are out of scope, SURVEY.md 2.
"""
import logging

import torch
from torch import nn

from .config import DictionaryUtility, ModuleUtility
from .segments import SE_DEAD_PMTS, segment_status, single_ended_mask
from .znet import SingleEndedEZConv, SingleEndedZConv


class LitSegmentBase(nn.Module):
    """What LitZ and LitEZ share of the reference's LitBase with ``event_predictions=False`` (LitBase.py:13-55)."""
    model_class = None

    def __init__(self, config, trial=None):
        super().__init__()
        self.trial = trial
        self.pylog = logging.getLogger(__name__)
        self.config = config
        self.nx, self.ny = 14, 11
        self.lr = config.optimize_config.lr
        self.modules_util = ModuleUtility(config.net_config.imports + config.dataset_config.imports +
                                          config.optimize_config.imports)
        self.model = type(self).model_class(config)
        criterion_class = self.modules_util.retrieve_class(config.net_config.criterion_class)
        self.criterion = criterion_class(*config.net_config.criterion_params, reduction="sum")   # event_predictions=False
        self.occlude_index = getattr(config.dataset_config, "occlude_index", None)
        self.SE_only = bool(getattr(config.net_config, "SELoss", False))
        if self.SE_only:
            # the reference takes the table from its SingleEndedEvaluator; `net_config.SE_dead_pmts` replaces the default
            dead = getattr(config.net_config, "SE_dead_pmts", SE_DEAD_PMTS)
            self.register_buffer("SE_mask", single_ended_mask(segment_status(dead, self.nx, self.ny)))
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

    # reference LitBase._calc_segment_loss, :124-174
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
        if self.SE_only:
            loss = self.criterion.forward(self.SE_mask * predictions, self.SE_mask * want)
            num_predictions = torch.sum(self.SE_mask * sparse_mask)
        else:
            loss = self.criterion.forward(predictions, want)
        return loss / num_predictions, target_tensor, predictions, sparse_mask


class LitZ(LitSegmentBase):
    model_class = SingleEndedZConv

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


class LitEZ(LitSegmentBase):
    """Energy + z regression: a 2-plane prediction map, the loss the sum of the two per-plane segment losses
    (reference LitEZ._process_batch, :57-73; the second call reuses the first call's active-segment mask)."""
    model_class = SingleEndedEZConv

    def __init__(self, config, trial=None):
        super().__init__(config, trial)
        nc = config.net_config
        self.zscale = getattr(nc, "zscale", 1200.)
        self.escale = getattr(nc, "escale", 12.)
        self.e_adjust = getattr(nc, "e_adjust", 12.)
        self.e_factor = self.escale / self.e_adjust
        self.phys_coord = nc.algorithm == "features"

    def _process_batch(self, batch):
        (c, f), target = batch
        if self.phys_coord and self.e_factor != 1.:
            f[:, 0] *= self.e_factor
            f[:, 2] *= self.e_factor
            f[:, 3] *= self.e_factor
        if self.occlude_index:
            f[:, self.occlude_index] = 0
        predictions = self.model([c, f])
        ZLoss, target_z, predictions_z, sparse_mask = self._calc_segment_loss(c, predictions[:, 0].unsqueeze(1),
                                                                              target[:, 0])
        ELoss, target_E, predictions_E, _ = self._calc_segment_loss(c, predictions[:, 1].unsqueeze(1), target[:, 1],
                                                                    sparse_mask=sparse_mask)
        predictions = torch.cat((predictions_z, predictions_E), dim=1)
        target_tensor = torch.cat((target_z, target_E), dim=1)
        return c, f, predictions, target_tensor, ZLoss + ELoss, ELoss, ZLoss

    def training_step(self, batch, batch_idx):
        loss = self._process_batch(batch)[4]
        self.log("train_loss", loss, on_epoch=True, prog_bar=True, logger=True)
        return loss

    def validation_step(self, batch, batch_idx):
        _, _, _, _, loss, ELoss, ZLoss = self._process_batch(batch)
        results = {"val_loss": loss, "val_MAE_E": ELoss, "val_MAE_z": ZLoss}
        self.log_dict(results, on_epoch=True, prog_bar=True, logger=True)
        return results

    def test_step(self, batch, batch_idx):
        _, _, _, _, loss, ELoss, ZLoss = self._process_batch(batch)
        results = {"test_loss": loss, "test_MAE_E": ELoss, "test_MAE_z": ZLoss}
        self.log_dict(results, on_epoch=True, logger=True)
        return results

"""``LitSegClassifier``: host-side mirror of the reference's per-SEGMENT classifier (src/engineering/LitSegClassifier.py:15-100
on src/engineering/LitBase.py:13-55): the net -- ``SPConvNet.SPConvPreserveNet`` in config/examples/IoniClassifierCNN.json --
returns one logit row per ACTIVE segment [N, n_type], the target holds one label per row, the criterion is mean-reduced
(``net_config.SELoss``: over the rows of single-ended segments only).  The torch_geometric ``Data`` batch form, the
torchmetrics objects and the PIDEvaluator / ROC plumbing are out of scope.
"""
import torch

from .lit import LitPSD
from .segments import SE_DEAD_PMTS, segment_status, single_ended_mask


class LitSegClassifier(LitPSD):
    per_row_targets = True            # one label per active segment (row): psd/graph.GraphedTrainStep pads them per row

    def __init__(self, config, trial=None):
        super().__init__(config, trial)
        self.softmax = torch.nn.Softmax(dim=1)
        self.SE_only = bool(getattr(config.net_config, "SELoss", False))
        if self.SE_only:
            dead = getattr(config.net_config, "SE_dead_pmts", SE_DEAD_PMTS)
            self.register_buffer("SE_mask", single_ended_mask(segment_status(dead)))

    # reference LitSegClassifier._process_batch, :36-63
    def _process_batch(self, batch):
        inputs, target = batch
        c, f = inputs[0], inputs[1]
        n_valid = inputs[2] if len(inputs) > 2 else None       # capacity-padded batch of a captured step (psd/graph.py)
        additional_fields = None
        if isinstance(f, list):
            additional_fields, f = f[1:], f[0]
        if self.occlude_index:
            f[:, self.occlude_index] = 0
        predictions = self.model([c, f, n_valid] if n_valid is not None else [c, f])
        logits = predictions if predictions.dtype == torch.float32 else predictions.float()
        if self.SE_only:
            # the reference indexes predictions[se_inds] / target[se_inds]; the same mean over the same rows with a
            # static shape: every other row's target becomes the criterion's ignore_index
            se_inds = self.SE_mask[0, 0, c[:, 0].long(), c[:, 1].long()] == 1.0
            ignore = torch.full_like(target, getattr(self.criterion, "ignore_index", -100))
            loss = self._loss(logits, torch.where(se_inds, target, ignore))
        else:
            loss = self._loss(logits, target)
        return loss, predictions, target, c, f, additional_fields

    def training_step(self, batch, batch_idx):
        loss = self._process_batch(batch)[0]
        self.log("train_loss", loss, on_epoch=True, prog_bar=True, logger=True)
        return loss

    def validation_step(self, batch, batch_idx):
        loss, predictions, target = self._process_batch(batch)[:3]
        pred = torch.argmax(self.softmax(predictions), dim=1)
        results = {"val_loss": loss, "val_acc": (pred == target).float().mean()}
        self.log_dict(results, on_epoch=True, prog_bar=True, logger=True)
        return results

    def test_step(self, batch, batch_idx):
        loss, predictions, target = self._process_batch(batch)[:3]
        pred = torch.argmax(self.softmax(predictions), dim=1)
        results = {"test_loss": loss, "test_acc": (pred == target).float().mean()}
        self.log_dict(results, on_epoch=True, logger=True)
        return results

"""Minimal stand-in for ``pytorch_lightning.Trainer`` on the PSD path (reference main.py:206-214):
fit loop over a LitPSD-style module, per-epoch scheduler step, validation, best-checkpoint save.
One process per GPU; gradients are exchanged by ddp.FlatGradAllReducer when world_size > 1."""
import os

import torch
import torch.distributed as dist

from .data import DevicePrefetcher, to_device
from .ddp import FlatGradAllReducer, broadcast_parameters


class Trainer(object):
    def __init__(self, max_epochs=1, device="cuda:0", default_root_dir=None, feature_dtype=None, log_every=0):
        self.max_epochs, self.device = max_epochs, torch.device(device)
        self.root = default_root_dir
        self.feature_dtype = feature_dtype
        self.log_every = log_every
        self.history = []

    def training_step(self, module, reducer, optimizer, batch, batch_idx):
        reducer.reset()
        loss = module.training_step(batch, batch_idx)
        loss.backward()
        reducer.finish()
        optimizer.step()
        return loss

    def fit(self, module, train_loader, val_loader=None):
        module.to(self.device)
        broadcast_parameters(module)
        reducer = FlatGradAllReducer(module.model.parameters())
        module.optimizer_parameters = reducer.optimizer_parameters()
        opt = module.configure_optimizers()
        optimizer, scheduler = (opt[0][0], opt[1][0]) if isinstance(opt, tuple) else (opt, None)
        best = float("inf")
        for epoch in range(self.max_epochs):
            module.train()
            for i, batch in enumerate(DevicePrefetcher(train_loader, self.device, self.feature_dtype)):
                loss = self.training_step(module, reducer, optimizer, batch, i)
                if self.log_every and i % self.log_every == 0:
                    print("epoch %d step %d train_loss %.5f" % (epoch, i, loss.item()), flush=True)
            if scheduler is not None:
                scheduler.step()
            rec = {"epoch": epoch, "train_loss": float(loss.item())}
            if val_loader is not None:
                rec.update(self.validate(module, val_loader))
                if self.root and rec["val_loss"] < best and (not dist.is_initialized() or dist.get_rank() == 0):
                    best = rec["val_loss"]
                    os.makedirs(self.root, exist_ok=True)
                    torch.save(module.state_dict(), os.path.join(self.root, "epoch=%d-val_loss=%.2f.ckpt" % (epoch, best)))
            self.history.append(rec)
        reducer.remove()
        return self.history

    @torch.no_grad()
    def validate(self, module, loader):
        module.eval()
        tot, n, acc = 0.0, 0, 0.0
        for i, batch in enumerate(loader):
            batch = to_device(batch, self.device, self.feature_dtype)
            res = module.validation_step(batch, i)
            b = batch[1].shape[0]
            tot += float(res["val_loss"]) * b
            acc += float(res["val_acc"]) * b
            n += b
        return {"val_loss": tot / max(n, 1), "val_acc": acc / max(n, 1)}

    def test(self, module, loader):
        """``pl.Trainer.test`` for the PSD module (reference Evaluate.py:84): see psd/evaluate.test_loop."""
        from .evaluate import test_loop
        return test_loop(module, loader, self.device, self.feature_dtype)

"""Minimal stand-in for ``pytorch_lightning.Trainer`` on the PSD path (reference main.py:206-214):
fit loop over a LitPSD-style module, per-epoch scheduler step, validation, best-checkpoint save.
One process per GPU; gradients are exchanged by ddp.FlatGradAllReducer when world_size > 1.

``capture=True`` runs the training steps as replays of ONE captured HIP graph (psd/graph.GraphedTrainStep: rulebook
builds, forward, backward, gradient packing and -- on one GPU -- the optimizer in a single launch of the whole step,
~3x the eager rate at the PSD batch sizes).  The graph is captured on the first batch with row capacities a few sigma
above it; the capture's own calibration / warm-up steps are undone (parameters, BatchNorm buffers and optimizer state
are restored), a batch with more voxels than the capacity takes an ordinary eager step, and every ``check_every``
steps (and at the end of an epoch) the strided layers' overflow flags are read back: an overflow raises, it is never
silently truncated."""
import os

import torch
import torch.distributed as dist

from .data import DevicePrefetcher, to_device
from .ddp import FlatGradAllReducer, broadcast_parameters


class Trainer(object):
    def __init__(self, max_epochs=1, device="cuda:0", default_root_dir=None, feature_dtype=None, log_every=0,
                 capture=False, check_every=100):
        self.max_epochs, self.device = max_epochs, torch.device(device)
        self.root = default_root_dir
        self.feature_dtype = feature_dtype
        self.log_every = log_every
        self.capture, self.check_every = bool(capture), int(check_every)
        self.history = []
        self._graph = None
        self.eager_fallbacks = 0

    def _capture(self, module, reducer, optimizer, batch):
        """Capture the step on ``batch`` without letting the capture's calibration / warm-up steps train the model."""
        from .graph import GraphedTrainStep
        params = reducer.flat_param.detach().clone() if reducer.flat_param is not None else \
            [p.detach().clone() for p in reducer.params]
        buffers = [b.detach().clone() for b in module.buffers()]
        had_state = {id(p): (p in optimizer.state and len(optimizer.state[p]) > 0)
                     for g in optimizer.param_groups for p in g["params"]}
        graph = GraphedTrainStep(module, optimizer, reducer, batch)
        with torch.no_grad():
            if reducer.flat_param is not None:
                reducer.flat_param.copy_(params)
            else:
                for p, q in zip(reducer.params, params):
                    p.copy_(q)
            for b, q in zip(module.buffers(), buffers):
                b.copy_(q)
            for g in optimizer.param_groups:
                for p in g["params"]:
                    if not had_state[id(p)]:
                        for v in optimizer.state.get(p, {}).values():
                            if torch.is_tensor(v):
                                v.zero_()      # in place: the graph holds these addresses (zero momentum == no history)
        return graph

    def _captured_step(self, module, reducer, optimizer, batch, batch_idx):
        (coords, _feats), _labels = batch
        if self._graph is None:
            self._graph = self._capture(module, reducer, optimizer, batch)
        if coords.shape[0] > self._graph.n_cap or _labels.shape != self._graph.labels.shape:
            self.eager_fallbacks += 1           # more voxels than the capacity, or another number of events
            return self.training_step(module, reducer, optimizer, batch, batch_idx)
        loss = self._graph(batch)
        if self.check_every > 0 and (batch_idx + 1) % self.check_every == 0:
            self._graph.check()
        return loss

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
        self._graph = None                      # a captured step belongs to this fit's optimizer / reducer
        for epoch in range(self.max_epochs):
            module.train()
            for i, batch in enumerate(DevicePrefetcher(train_loader, self.device, self.feature_dtype)):
                if self.capture:
                    loss = self._captured_step(module, reducer, optimizer, batch, i)
                else:
                    loss = self.training_step(module, reducer, optimizer, batch, i)
                if self.log_every and i % self.log_every == 0:
                    print("epoch %d step %d train_loss %.5f" % (epoch, i, loss.item()), flush=True)
            if self._graph is not None:
                self._graph.check()
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
        if self.capture and not module.occlude_index:
            # same numbers as validation_step (loss + accuracy of the eval-mode forward), from replays of a captured
            # forward (psd/graph.GraphedEvalStep)
            from .evaluate import test_loop
            r = test_loop(module, loader, self.device, self.feature_dtype, capture=True)
            return {"val_loss": r["test_loss"], "val_acc": r["test_acc"]}
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

"""Inference-only loops of the PSD path.

``test_loop``        what ``pytorch_lightning.Trainer.test(runner, datamodule)`` does for the reference's Evaluate.py
                     (Evaluate.py:69-84): eval mode, no gradients, ``LitPSD.test_step`` per batch (which zeroes feature
                     column ``occlude_index`` when it is set, src/engineering/LitPSD.py:130-151), event-weighted means.
``occlusion_sweep``  the reference's occlusion study (scripts/RunOcclusionStudy.py) runs Evaluate.py once per feature
                     index, i.e. re-reads the data and rebuilds every rulebook for each index although the geometry of a
                     batch never changes.  Here ONE batch already in HBM is run through the net for a list of indices
                     with its rulebooks built once (spconv.ops.reuse_rulebooks); only the feature column differs.
"""
import torch

from ..spconv import ops
from .data import to_device


@torch.no_grad()
def test_loop(module, loader, device, feature_dtype=None):
    module.to(device)
    module.eval()
    tot, acc, n = 0.0, 0.0, 0
    for i, batch in enumerate(loader):
        batch = to_device(batch, device, feature_dtype)
        res = module.test_step(batch, i)
        b = int(batch[1].shape[0])
        tot += float(res["test_loss"]) * b
        acc += float(res["test_acc"]) * b
        n += b
    return {"test_loss": tot / max(n, 1), "test_acc": acc / max(n, 1), "events": n}


@torch.no_grad()
def occlusion_sweep(module, batch, occlude_indices):
    """{index: {"test_loss", "test_acc"}} for one device-resident batch ``([coords, feats], labels)``.
    ``None`` in ``occlude_indices`` is the unoccluded pass.  Index 0 is passed through as it is: the reference's
    ``if self.occlude_index:`` treats it as "no occlusion" (LitPSD.py:134), and so does the mirror."""
    (coords, feats), labels = batch
    module.eval()
    saved = module.occlude_index
    out = {}
    try:
        with ops.reuse_rulebooks():
            for idx in occlude_indices:
                module.occlude_index = idx
                res = module.test_step(([coords, feats.clone()], labels), 0)      # test_step zeroes the column in place
                out[idx] = {"test_loss": float(res["test_loss"]), "test_acc": float(res["test_acc"])}
    finally:
        module.occlude_index = saved
    return out

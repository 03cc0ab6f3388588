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


def _score(module, logits, labels):
    """test_loss / test_acc of reference LitPSD.test_step (:136-141) from the logits."""
    loss = module.criterion.forward(logits, labels)
    pred = torch.argmax(module.softmax(logits), dim=1)
    return loss.detach().float(), (pred == labels).float().mean()          # device scalars: no host sync per batch


@torch.no_grad()
def test_loop(module, loader, device, feature_dtype=None, capture=False):
    """``capture=True``: the forward runs as replays of one captured HIP graph (psd/graph.GraphedEvalStep); a batch that
    does not fit the captured capacities takes the ordinary ``test_step``."""
    module.to(device)
    module.eval()
    tot, acc, n = 0.0, 0.0, 0          # tot / acc become device scalars: one read-back at the end of the loop
    step = None
    for i, batch in enumerate(loader):
        batch = to_device(batch, device, feature_dtype)
        b = int(batch[1].shape[0])
        if capture:
            if step is None:
                from .graph import GraphedEvalStep
                step = GraphedEvalStep(module, batch)
            if step.fits(batch):
                loss, a = _score(module, step(batch, module.occlude_index), batch[1])
                tot, acc, n = tot + loss * b, acc + a * b, n + b
                continue
        res = module.test_step(batch, i)
        tot = tot + res["test_loss"].detach().float() * b
        acc = acc + res["test_acc"].detach().float() * b
        n += b
    if step is not None:
        step.check()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        # every rank ran its own share of the items (data.rank_sampler): event-weighted sums over all ranks
        sums = torch.stack([torch.as_tensor(tot, dtype=torch.float64, device=device).reshape(()),
                            torch.as_tensor(acc, dtype=torch.float64, device=device).reshape(()),
                            torch.tensor(float(n), dtype=torch.float64, device=device)])
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        tot, acc, n = float(sums[0]), float(sums[1]), int(sums[2])
    return {"test_loss": float(tot) / max(n, 1), "test_acc": float(acc) / max(n, 1), "events": n}


@torch.no_grad()
def occlusion_sweep(module, batch, occlude_indices, capture=False):
    """{index: {"test_loss", "test_acc"}} for one device-resident batch ``([coords, feats], labels)``.
    ``None`` in ``occlude_indices`` is the unoccluded pass.  Index 0 is passed through as it is: the reference's
    ``if self.occlude_index:`` treats it as "no occlusion" (LitPSD.py:134), and so does the mirror.
    ``capture=True``: one captured graph builds the batch's rulebooks and runs the first pass, a second, forward-only
    graph is replayed for every further index (psd/graph.GraphedEvalStep(sweep=True))."""
    (coords, feats), labels = batch
    module.eval()
    out = {}
    if capture and len(occlude_indices) > 0:
        from .graph import GraphedEvalStep
        step = GraphedEvalStep(module, batch, sweep=True)
        try:
            scores = []
            for n, idx in enumerate(occlude_indices):
                logits = step(batch, idx) if n == 0 else step.rerun(idx)
                scores.append(torch.stack(_score(module, logits, labels)))
            host = torch.stack(scores).cpu()              # one read-back for the whole sweep
            for idx, row in zip(occlude_indices, host):
                out[idx] = {"test_loss": float(row[0]), "test_acc": float(row[1])}
            step.check()
        finally:
            step.close()
        return out
    saved = module.occlude_index
    try:
        with ops.reuse_rulebooks():
            for idx in occlude_indices:
                module.occlude_index = idx
                res = module.test_step(([coords, feats.clone()], labels), 0)      # test_step zeroes the column in place
                out[idx] = {"test_loss": float(res["test_loss"]), "test_acc": float(res["test_acc"])}
    finally:
        module.occlude_index = saved
    return out

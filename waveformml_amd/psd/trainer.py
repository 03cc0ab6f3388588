"""Minimal stand-in for ``pytorch_lightning.Trainer`` on the PSD path (reference main.py:206-214):
fit loop over a LitPSD-style module, per-epoch scheduler step, validation, best-checkpoint save.
One process per GPU; gradients are exchanged by ddp.FlatGradAllReducer when world_size > 1.

``capture=True`` runs the training steps as replays of ONE captured HIP graph (psd/graph.GraphedTrainStep: rulebook
builds, forward, backward, gradient packing and -- on one GPU -- the optimizer in a single launch of the whole step,
~3x the eager rate at the PSD batch sizes).  The graph is captured on the first batch with row capacities a few sigma
above it; the capture's own calibration / warm-up steps are undone (parameters, BatchNorm buffers and optimizer state
are restored), a batch with more voxels than the capacity takes an ordinary eager step, and every ``check_every``
steps (and at the end of an epoch) the strided layers' overflow flags are read back: an overflow raises, it is never
silently truncated.

More than one rank (one process per GPU): each rank reads its own share of the items (data.rank_sampler, re-seeded per
epoch), whether a step is replayed or taken eagerly is decided by ALL ranks together (a one-element MAX all-reduce of
"my batch does not fit"), the eager step issues the same collectives as a replay, validation sums are all-reduced
before the best-checkpoint decision, overflow flags are all-reduced before anybody raises.

Checkpoints are dictionaries shaped like Lightning's (``state_dict``, ``epoch``, ``global_step``, ``optimizer_states``,
``lr_schedulers``) under Lightning's file name pattern; ``load_from_checkpoint`` / ``Trainer(resume_from_checkpoint=)``
read them -- and a reference-written ``.ckpt`` 's ``state_dict`` -- with ``torch.load(weights_only=True)``
(reference: Evaluate.py:72 ``load_from_checkpoint``, main.py ``--load_checkpoint`` -> ``resume_from_checkpoint``)."""
import os
import time

import torch
import torch.distributed as dist

from .data import DevicePrefetcher, to_device
from .ddp import FlatGradAllReducer, broadcast_parameters


def _world():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def read_checkpoint(path, map_location="cpu"):
    """A checkpoint file as a dictionary with at least ``state_dict``: this Trainer's files, a reference / Lightning
    ``.ckpt`` (same keys; anything that needs unpickling of foreign classes is refused by ``weights_only=True``), or a
    bare ``state_dict`` as round 1 of this repository wrote them."""
    ck = torch.load(path, map_location=map_location, weights_only=True)
    if not isinstance(ck, dict):
        raise ValueError("%s is not a checkpoint dictionary" % path)
    if "state_dict" not in ck:
        ck = {"state_dict": ck}
    return ck


def load_from_checkpoint(path, config, module_class=None, map_location="cpu", strict=True):
    """``LitPSD.load_from_checkpoint(path, config=config)`` of the reference (Evaluate.py:72): builds the module from
    ``config`` and loads the checkpoint's ``state_dict``."""
    if module_class is None:
        from .lit import LitPSD as module_class
    module = module_class(config)
    module.load_state_dict(read_checkpoint(path, map_location)["state_dict"], strict=strict)
    return module


def _offsets_in_flat(flat, params):
    """Element offset of every model parameter inside the flat parameter it is a view of (the reducer lays buckets out
    in reverse layer order, so this is NOT the running sum of the sizes); the running sum for parameters that are not
    views of it."""
    base, size, out, run = flat.data_ptr(), flat.element_size(), [], 0
    for p in params:
        off = (p.data_ptr() - base) // size if p.device == flat.device else -1
        if not (0 <= off <= flat.numel() - p.numel() and (p.data_ptr() - base) % size == 0):
            off = run
        out.append(int(off))
        run += p.numel()
    return out


def per_parameter_optimizer_state(optimizer, params):
    """``optimizer.state_dict()`` in the layout a torch optimizer over ``model.parameters()`` writes -- what Lightning puts
    into ``optimizer_states`` for the reference (src/engineering/LitPSD.py:60-76 builds the optimizer over
    ``self.model.parameters()``): one state entry per parameter, indexed in parameter order.  This trainer's optimizer
    holds ONE flat parameter (psd/ddp.FlatGradAllReducer); its per-element state tensors (``momentum_buffer``, Adam's
    ``exp_avg`` ...) are cut at the parameters' offsets, scalar entries (``step``) are repeated."""
    sd = optimizer.state_dict()
    groups = sd["param_groups"]
    total = sum(p.numel() for p in params)
    flat = [p for g in optimizer.param_groups for p in g["params"]]
    if len(flat) != 1 or flat[0].numel() != total or len(params) == 1:
        return sd                                   # already one entry per parameter
    state = {}
    src = sd["state"].get(0, {})
    offsets = _offsets_in_flat(flat[0], params)
    for i, p in enumerate(params):
        n, off = p.numel(), offsets[i]
        entry = {}
        for name, v in src.items():
            if torch.is_tensor(v) and v.numel() == total:
                entry[name] = v.reshape(-1)[off:off + n].reshape(p.shape).detach().cpu().clone()
            else:
                entry[name] = v.detach().cpu().clone() if torch.is_tensor(v) else v
        if entry:
            state[i] = entry
    group = {k: v for k, v in groups[0].items() if k != "params"}
    group["params"] = list(range(len(params)))
    return {"state": state, "param_groups": [group]}


def load_optimizer_state(optimizer, sd, params):
    """Inverse of per_parameter_optimizer_state: accepts the optimizer's own layout or one entry per model parameter
    (this trainer's checkpoints, or a checkpoint the reference's Lightning run wrote with the same optimizer class) and
    gathers the per-parameter tensors into the flat parameter's state.  Anything else raises with both layouts named."""
    flat = [p for g in optimizer.param_groups for p in g["params"]]
    n_saved = sum(len(g["params"]) for g in sd["param_groups"])
    if n_saved == len(flat):
        optimizer.load_state_dict(sd)
        return
    if len(flat) != 1 or n_saved != len(params) or len(sd["param_groups"]) != 1:
        raise RuntimeError("checkpoint optimizer state covers %d parameters in %d group(s); this run's optimizer has %d "
                           "(the model has %d parameters): load the weights only (state_dict) or resume with the same "
                           "optimizer configuration" % (n_saved, len(sd["param_groups"]), len(flat), len(params)))
    names = set()
    for st in sd["state"].values():
        names.update(st.keys())
    entry = {}
    for name in names:
        per_elem = any(torch.is_tensor(st.get(name)) and st[name].numel() > 1 for st in sd["state"].values())
        if not per_elem:
            entry[name] = next(st[name] for st in sd["state"].values() if name in st)
            continue
        buf = torch.zeros(flat[0].numel(), dtype=flat[0].dtype)
        offsets = _offsets_in_flat(flat[0], params)
        for i, p in enumerate(params):
            v = sd["state"].get(i, {}).get(name)
            if v is None:
                continue
            if tuple(v.shape) != tuple(p.shape):
                raise RuntimeError("checkpoint optimizer state '%s' of parameter %d has shape %s, the parameter %s"
                                   % (name, i, tuple(v.shape), tuple(p.shape)))
            buf[offsets[i]:offsets[i] + p.numel()] = v.reshape(-1).to(flat[0].dtype).cpu()
        entry[name] = buf.reshape(flat[0].shape)
    group = {k: v for k, v in sd["param_groups"][0].items() if k != "params"}
    group["params"] = [0]
    optimizer.load_state_dict({"state": {0: entry} if entry else {}, "param_groups": [group]})


class Trainer(object):
    def __init__(self, max_epochs=1, device="cuda:0", default_root_dir=None, feature_dtype=None, log_every=0,
                 capture=False, check_every=100, resume_from_checkpoint=None, recapture_after=4, agree_block=4):
        self.max_epochs, self.device = max_epochs, torch.device(device)
        self.root = default_root_dir
        self.feature_dtype = feature_dtype
        self.log_every = log_every
        self.capture, self.check_every = bool(capture), int(check_every)
        self.resume_from_checkpoint = resume_from_checkpoint
        self.history = []
        self._graph = None
        # a captured step sizes its row capacities on its first batch + a few sigma of an event-level mix; batches whose
        # files differ systematically (one class per file, classes with different pulse lengths) can exceed them often:
        # after `recapture_after` misfits by SIZE the step is captured again on the batch that did not fit
        self.recapture_after = int(recapture_after)
        # several ranks: batch shapes are agreed `agree_block` batches per (asynchronous, host-side) all-reduce, staged that
        # many batches ahead (graph.ShapeAgreement); 0 = one blocking all-reduce per step, as in round 2
        self.agree_block = int(agree_block)
        self.recaptures = 0
        self.last_capacity = 0
        self._size_misfits = 0
        self._label_misfits = 0
        self._rows_stats = [0, 0.0, 0.0, 0]          # batches seen, sum of rows, sum of squares, largest
        self.eager_fallbacks = 0
        self.global_step = 0
        self.last_checkpoint = None

    def _observe_rows(self, rows):
        """Row counts of the batches seen so far (with several ranks: the agreed maximum over the ranks), for sizing a
        RE-capture: by then the loop knows the data's spread, which the first capture had to guess."""
        st = self._rows_stats
        st[0] += 1
        st[1] += float(rows)
        st[2] += float(rows) * float(rows)
        st[3] = max(st[3], int(rows))

    def _recapture_rows(self, batch_rows):
        """Row capacity of a re-capture: room for the largest batch seen and for mean + 4 sigma of the row counts seen,
        plus 2 % -- instead of the first capture's 1 + 3 / sqrt(events) on top of an already large batch (from files
        that compounded to 1.40 x the mean: 121 k rows of capacity for batches of 85 k, +20 us per step).  None until
        enough batches have been seen.  An absolute row count: with several ranks it is computed from the AGREED maxima,
        so every rank arrives at the same capacity."""
        n, s1, s2, mx = self._rows_stats
        if n < 8:
            return None
        mean = s1 / n
        sigma = max(0.0, s2 / n - mean * mean) ** 0.5
        return int(max(float(mx), mean + 4.0 * sigma, float(batch_rows)) * 1.02) + 1

    def _capture(self, module, reducer, optimizer, batch, min_rows=0, headroom=None):
        """Capture the step on ``batch`` without letting the capture's calibration / warm-up steps train the model."""
        from .graph import GraphedTrainStep
        params = reducer.flat_param.detach().clone() if reducer.flat_param is not None else \
            [p.detach().clone() for p in reducer.params]
        buffers = [b.detach().clone() for b in module.buffers()]
        # optimizer state that exists already (a resumed checkpoint's momentum): kept aside and copied back IN PLACE after
        # the capture -- its calibration and warm-up steps would otherwise be folded into it.  State the capture creates
        # is zeroed (zero momentum == no history).
        kept = {id(p): {k: v.detach().clone() for k, v in optimizer.state[p].items() if torch.is_tensor(v)}
                for g in optimizer.param_groups for p in g["params"] if p in optimizer.state and len(optimizer.state[p]) > 0}
        graph = GraphedTrainStep(module, optimizer, reducer, batch, min_rows=min_rows, headroom=headroom)
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
                    old = kept.get(id(p))
                    for k, v in optimizer.state.get(p, {}).items():
                        if torch.is_tensor(v):           # in place: the graph holds these addresses
                            if old is not None and k in old and old[k].shape == v.shape:
                                v.copy_(old[k])
                            else:
                                v.zero_()
        if hasattr(optimizer, "mark_fresh") and not kept:
            optimizer.mark_fresh()             # dampening != 0: the first real step must be torch's "buf = g" (eager)
        return graph

    def _captured_step(self, module, reducer, optimizer, batch, batch_idx, agreed=None):
        # several ranks deciding from agreed counts must hold the SAME capacity: a capture is sized on the largest batch
        # any rank has at that step (each rank's own example batch would give each rank its own capacity)
        floor = 0
        if agreed is not None:
            from .graph import GraphedTrainStep
            floor = GraphedTrainStep.capacity_for(agreed[0], agreed[1])     # largest rows, smallest label count (= largest headroom)
        per_row = bool(getattr(module, "per_row_targets", False))
        self._observe_rows(agreed[0] if agreed is not None else batch[0][0].shape[0])
        if self._graph is None:
            if agreed is not None and not per_row and agreed[1] != agreed[2]:
                # the ranks' batches hold different numbers of events (a partial last file): a step captured on this
                # rank's odd count would misfit on every later batch -- and only on this rank.  Everybody
                # steps eagerly (same agreed counts -> same decision); the first batch with an agreed count is captured
                self.eager_fallbacks += 1
                return self.training_step(module, reducer, optimizer, batch, batch_idx)
            self._graph = self._capture(module, reducer, optimizer, batch, min_rows=floor)
        # more voxels than the capacity, or another number of events -> an ordinary step.  With several ranks the
        # decision is taken together: a rank replaying while another steps eagerly must never depend on the two paths
        # happening to issue the same collectives.
        # Both inputs of the decision are host-side facts (tensor shapes, the optimizer's own bookkeeping), so the ranks
        # agree over a gloo side group of host integers: no device collective, no stream synchronisation per step -- the
        # host keeps queueing the next batch's copies and the next replay.  Bit 0: an ordinary step is
        # needed; bit 1: because the batch has more rows than the captured capacity.
        misfit = not self._graph.fits(batch) or (hasattr(optimizer, "has_fresh") and optimizer.has_fresh())
        too_big = batch[0][0].shape[0] > self._graph.n_cap
        if reducer.world > 1 and reducer.exchange and agreed is not None:
            # the counts were agreed when the batch was staged (graph.ShapeAgreement: asynchronous, batches ahead): every
            # rank derives the same decision from them, whatever capacity the step has by now; the optimizer's
            # bookkeeping (has_fresh) is the same on every rank by construction
            rows_max, labels_min, labels_max, rows_ne_labels = agreed
            fresh = hasattr(optimizer, "has_fresh") and optimizer.has_fresh()
            misfit = fresh or not self._graph.fits_counts(rows_max, labels_min, labels_max, rows_ne_labels)
            too_big = rows_max > self._graph.n_cap
        elif reducer.world > 1 and reducer.exchange:
            from .graph import _agree_max
            code = _agree_max((1 if misfit else 0) | (2 if (misfit and too_big) else 0), reducer.group)
            # MAX of the codes: 1 < 2 < 3, and a rank reporting 2 cannot exist (too_big implies misfit)
            misfit, too_big = code > 0, code >= 2
        # a batch that fits the rows but holds another number of labels than the capture (and every rank agrees on that
        # number): counted like the size misfits -- `recapture_after` of them in a row re-capture on the new shape
        fresh_now = hasattr(optimizer, "has_fresh") and optimizer.has_fresh()
        labels_agree = agreed is None or per_row or agreed[1] == agreed[2]
        label_misfit = (misfit and not too_big and not fresh_now and labels_agree
                        and (agreed is not None or not (reducer.world > 1 and reducer.exchange)))
        self._label_misfits = self._label_misfits + 1 if label_misfit else 0
        recapture_labels = label_misfit and self.recapture_after > 0 and self._label_misfits >= self.recapture_after
        if misfit:
            self._size_misfits += 1 if too_big else 0
            if recapture_labels or (too_big and self.recapture_after > 0 and self._size_misfits >= self.recapture_after):
                # the capacity is too small for this data: capture again, sized on this batch (never smaller than before)
                old = self._graph
                self._graph = None
                n_old = old.n_cap
                old.close(remove_hooks=False)          # the old graph goes (device idle) before the new one is captured
                del old
                rows_now = agreed[0] if agreed is not None else batch[0][0].shape[0]
                target = self._recapture_rows(rows_now)
                if recapture_labels and not too_big:
                    target = n_old                   # only the label shape changed: the row capacity stays
                if target is not None:
                    self._graph = self._capture(module, reducer, optimizer, batch, min_rows=target, headroom=1.0)
                else:
                    self._graph = self._capture(module, reducer, optimizer, batch, min_rows=max(n_old, floor))
                self._size_misfits = 0
                self._label_misfits = 0
                self.recaptures += 1
                return self._graph(batch)
            self.eager_fallbacks += 1
            return self._graph.eager_step(batch)
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
        first_epoch = 0
        if self.resume_from_checkpoint:
            first_epoch = self._resume(module, optimizer, scheduler, self.resume_from_checkpoint)
        for epoch in range(first_epoch, self.max_epochs):
            t_epoch = time.perf_counter()
            module.train()
            sampler = getattr(train_loader, "sampler", None)
            if hasattr(sampler, "set_epoch"):
                sampler.set_epoch(epoch)        # DistributedSampler: another permutation per epoch, the same on all ranks
            agree = None
            if self.capture and reducer.world > 1 and reducer.exchange and self.agree_block > 0:
                from .graph import ShapeAgreement
                agree = ShapeAgreement(reducer.group, self.agree_block)
            prefetch = DevicePrefetcher(train_loader, self.device, self.feature_dtype,
                                        depth=2 if agree is None else agree.block + 2,
                                        on_stage=agree.stage if agree is not None else None,
                                        on_exhausted=agree.flush if agree is not None else None)
            for i, batch in enumerate(prefetch):
                if self.capture:
                    loss = self._captured_step(module, reducer, optimizer, batch, i,
                                               agree.next() if agree is not None else None)
                else:
                    loss = self.training_step(module, reducer, optimizer, batch, i)
                self.global_step += 1
                if self.log_every and i % self.log_every == 0:
                    print("epoch %d step %d train_loss %.5f" % (epoch, i, loss.item()), flush=True)
            if self._graph is not None:
                self._graph.check()
            if scheduler is not None:
                scheduler.step()
            rec = {"epoch": epoch, "train_loss": float(loss.item()),           # .item(): the epoch's work is done
                   "train_seconds": time.perf_counter() - t_epoch, "steps": i + 1,
                   "loader_wait_seconds": round(prefetch.wait_seconds, 4)}   # of which: waiting for the next batch
            if val_loader is not None:
                rec.update(self.validate(module, val_loader))        # all-reduced: every rank sees the same numbers
                if self.root and rec["val_loss"] < best:
                    best = rec["val_loss"]
                    if not dist.is_initialized() or dist.get_rank() == 0:
                        self.save_checkpoint(module, optimizer, scheduler, epoch,
                                             os.path.join(self.root, "epoch=%d-val_loss=%.2f.ckpt" % (epoch, best)))
            self.history.append(rec)
        # orderly end: hooks off, device idle, the captured graph destroyed -- before anybody destroys the process group
        # (the reference relies on Lightning's DDP teardown, src/utils/util.py:236)
        if self._graph is not None:
            self.last_capacity = int(self._graph.n_cap)      # row capacity of the last captured step (diagnostics, tests)
            self._graph.close()
            self._graph = None
        reducer.remove()
        return self.history

    def save_checkpoint(self, module, optimizer, scheduler, epoch, path):
        """Lightning's checkpoint layout (what the reference's ModelCheckpoint callback writes, main.py:190-204):
        tensors, numbers, strings, lists and dictionaries only, so ``weights_only=True`` loads it."""
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        ck = {"epoch": int(epoch), "global_step": int(self.global_step),
              "state_dict": {k: v.detach().cpu().clone() for k, v in module.state_dict().items()},
              "optimizer_states": [per_parameter_optimizer_state(optimizer, list(module.model.parameters()))],
              "lr_schedulers": [scheduler.state_dict()] if scheduler is not None else []}
        torch.save(ck, path)
        self.last_checkpoint = path
        return path

    def _resume(self, module, optimizer, scheduler, path):
        """``resume_from_checkpoint`` (reference main.py ``--load_checkpoint``): weights, optimizer state (momentum; one
        entry per model parameter, as a Lightning run of the reference writes it, or this optimizer's own flat layout),
        scheduler state and the epoch / step counters.  Epoch convention of THIS trainer's checkpoints: ``epoch`` = index of
        the last finished epoch, training continues with ``epoch + 1``.  (pytorch_lightning is not installable here, so
        which of its versions store ``current_epoch + 1`` instead could not be checked against a real file: a reference
        checkpoint resumes at most one epoch late, never early.)  The parameters are
        views of the reducer's flat buffer at this point, so ``load_state_dict`` copies into it in place."""
        ck = read_checkpoint(path, map_location=self.device)
        module.load_state_dict(ck["state_dict"])
        if ck.get("optimizer_states"):
            load_optimizer_state(optimizer, ck["optimizer_states"][0], list(module.model.parameters()))
        if scheduler is not None and ck.get("lr_schedulers"):
            scheduler.load_state_dict(ck["lr_schedulers"][0])
        self.global_step = int(ck.get("global_step", 0))
        return int(ck["epoch"]) + 1 if "epoch" in ck else 0

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
        if _world() > 1:
            # each rank validated its own share of the items: event-weighted sums over all ranks
            sums = torch.tensor([tot, acc, float(n)], dtype=torch.float64, device=self.device)
            dist.all_reduce(sums, op=dist.ReduceOp.SUM)
            tot, acc, n = float(sums[0]), float(sums[1]), int(sums[2])
        return {"val_loss": tot / max(n, 1), "val_acc": acc / max(n, 1)}

    def test(self, module, loader):
        """``pl.Trainer.test`` for the PSD module (reference Evaluate.py:84): see psd/evaluate.test_loop."""
        from .evaluate import test_loop
        return test_loop(module, loader, self.device, self.feature_dtype)

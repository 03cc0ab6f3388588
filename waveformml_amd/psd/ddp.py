"""Data-parallel gradient exchange for the PSD step: one process per GPU, RCCL over xGMI.

The reference gets data parallelism from Lightning's DDPPlugin -> torch DDP -> NCCL
(src/utils/util.py:228-239; no explicit collective call sites, SURVEY.md 2 "Parallelism inventory").
Events are independent units: each rank runs its own shard of the global batch, the
only exchange is the gradient all-reduce.  The PSD nets are tiny (0.03-1 M parameters), so the
exchange is latency-bound.  Layout:

  * all parameters live in ONE flat fp32 buffer (the module's parameters are views of it), so the
    optimizer updates a single tensor -- a handful of launches instead of one per parameter;
  * all gradients land in ONE flat buffer cut into contiguous buckets in reverse layer order.
    A bucket is packed (one concatenation kernel) and its all-reduce launched asynchronously -- RCCL
    runs it on its own stream -- the moment its last gradient has been produced, so with several
    buckets the exchange overlaps the rest of backward.  DEFAULT: ONE bucket (WFS_GRAD_BUCKETS or the
    argument for more).  Inside a captured step every bucket is a fork / join pair between streams of
    the HIP graph, and those cost more than overlapping a 0.5-MB all-reduce can return: the structure
    alone (one rank, real RCCL communicator, `WFS_BENCH_ONE_RANK_RCCL=1 python bench.py`) costs
    +3 us per step with 1 bucket, +26 us with 2, +107 us with 4 on a 0.515-ms step
    (profiles/r03_one_rank_rccl_structure.txt); the last bucket's all-reduce is exposed either way;
  * BatchNorm statistics stay per rank, exactly as under the reference's DDP (no SyncBN).
"""
import os

import torch
import torch.distributed as dist


class FlatGradAllReducer(object):
    def __init__(self, parameters, n_buckets=None, process_group=None, world_size=None, flatten=True, exchange=None):
        self.params = [p for p in parameters if p.requires_grad]
        if n_buckets is None:
            n_buckets = int(os.environ.get("WFS_GRAD_BUCKETS", "1"))
        if not self.params:
            raise ValueError("no trainable parameters")
        self.group = process_group
        self.world = world_size if world_size is not None else (
            dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1)
        # exchange: issue the collectives at all.  Default: only when there is somebody to exchange with; True with a
        # one-rank group runs the whole bucket / hook / collective machinery against the real backend (tests: the RCCL
        # call path on a one-GPU box, where two ranks cannot share the card)
        self.exchange = (self.world > 1) if exchange is None else bool(exchange)
        dev, dtype = self.params[0].device, self.params[0].dtype
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError("FlatGradAllReducer keeps fp32 master parameters / gradients (got %s): the HIP "
                                "operators address gradient slots as 4-byte elements" % p.dtype)
        # the reduction the backend runs: RCCL averages inside the collective (ncclAvg: one launch less than sum +
        # divide on a latency-bound step); probed ONCE here on a scratch tensor -- never by catching errors in a step,
        # which would retry on a communicator that may be dead
        self._avg = False
        if self.exchange and dist.is_available() and dist.is_initialized() and dist.get_backend(process_group) == "nccl":
            try:
                probe = torch.ones(8, dtype=dtype, device=dev)
                dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=process_group)
                self._avg = bool((probe == 1).all().item())
            except (RuntimeError, ValueError):
                self._avg = False
        total = sum(p.numel() for p in self.params)
        # reverse order: the LAST layer's gradients are produced first and sit at the front
        order = list(reversed(range(len(self.params))))
        off = 0
        self.slices = {}
        for i in order:
            self.slices[i] = (off, self.params[i].numel())
            off += self.params[i].numel()
        self.flat_grad = torch.zeros(total, dtype=dtype, device=dev)
        self.flat_param = None
        if flatten:
            flat = torch.empty(total, dtype=dtype, device=dev)
            with torch.no_grad():
                for i, p in enumerate(self.params):
                    o, n = self.slices[i]
                    flat[o:o + n].copy_(p.data.reshape(-1))
                    p.data = flat[o:o + n].view_as(p)
            self.flat_param = torch.nn.Parameter(flat)
            self.flat_param.grad = self.flat_grad
            if dev.type == "cuda":
                # backward passes of the HIP operators write parameter gradients straight into their slots of
                # flat_grad (spconv/functional.grad_like): _pack then finds them in place and copies nothing
                from ..spconv import functional as _fsp
                _fsp.register_grad_slots(self.flat_param, self.flat_grad)
        # contiguous buckets of roughly equal size over that order
        if not self.exchange:
            n_buckets = 1              # nothing to overlap: pack with one concatenation
        n_buckets = max(1, min(n_buckets, len(self.params)))
        target = total / n_buckets
        self.buckets = []          # (start, end, [param indices in flat order])
        cur, start, acc = [], 0, 0
        for i in order:
            cur.append(i)
            acc += self.slices[i][1]
            if acc >= target * (len(self.buckets) + 1) and len(self.buckets) < n_buckets - 1:
                end = self.slices[i][0] + self.slices[i][1]
                self.buckets.append((start, end, cur))
                cur, start = [], end
        if cur:
            self.buckets.append((start, total, cur))
        self.bucket_of = {i: b for b, (_, _, idxs) in enumerate(self.buckets) for i in idxs}
        self._pending = [0] * len(self.buckets)
        self._handles = []
        self._hooks = []
        if self.exchange:
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self.reset()

    def _reduce_op(self):
        return dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM

    def _all_reduce_bucket(self, b):
        s, e, _ = self.buckets[b]
        self._handles.append(dist.all_reduce(self.flat_grad[s:e], op=self._reduce_op(), group=self.group, async_op=True))

    # flat view of the gradient buffer (tests / diagnostics)
    @property
    def flat(self):
        return self.flat_grad

    def optimizer_parameters(self):
        """What to hand the optimizer: the single flat parameter (or the original list if not flattened)."""
        return [self.flat_param] if self.flat_param is not None else self.params

    def _pack(self, b):
        from ..spconv import functional as _fsp
        _fsp.join_side_streams()          # weight gradients may still be in flight on the dW side stream
        _fsp.flush_deferred_dw()          # ... or wait for their (deferred) second stage
        s, e, idxs = self.buckets[b]
        base = self.flat_grad.data_ptr()
        placed = []                       # gradients that already sit in their slot (written there by the kernels)
        for i in idxs:
            g = self.params[i].grad
            slot = base + 4 * self.slices[i][0]
            placed.append(_fsp.was_deferred(slot) or (g is not None and g.is_contiguous() and g.data_ptr() == slot
                                                      and g.dtype == self.flat_grad.dtype))
        if all(placed):
            return
        if not any(placed):
            grads = []
            for i in idxs:
                g = self.params[i].grad
                grads.append(g.reshape(-1) if g is not None else self.flat_grad.new_zeros(self.slices[i][1]))
            torch.cat(grads, out=self.flat_grad[s:e])
            return
        for i, ok in zip(idxs, placed):   # mixed: move only what is elsewhere
            if ok:
                continue
            o, n = self.slices[i]
            g = self.params[i].grad
            if g is None:
                self.flat_grad[o:o + n].zero_()
            else:
                self.flat_grad[o:o + n].copy_(g.reshape(-1))

    def _make_hook(self, i):
        def hook(param):
            b = self.bucket_of[i]
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._pack(b)
                self._all_reduce_bucket(b)
        return hook

    def reset(self):
        """Call before each backward: drop the per-parameter gradients (autograd then ASSIGNS fresh ones
        instead of launching an add per parameter) and re-arm the buckets."""
        for p in self.params:
            p.grad = None
        if self.flat_param is not None and self.flat_grad.is_cuda:
            from ..spconv import functional as _fsp
            _fsp.reset_grad_slots()
        self._pending = [len(idxs) for (_, _, idxs) in self.buckets]
        self._handles = []

    def finish(self):
        """Call after backward, before optimizer.step(): pack what is not packed yet, wait for the
        exchanges, average.  Afterwards flat_grad (== flat_param.grad) holds the step's gradient."""
        for b, left in enumerate(self._pending):
            if left > 0 or not self._hooks:
                # no hooks (nothing to exchange, or remove()d): nothing is packed yet; with hooks: a parameter without
                # gradient never fires its hook
                self._pack(b)
                if self.exchange:
                    self._all_reduce_bucket(b)
                self._pending[b] = 0
        for h in self._handles:
            h.wait()
        self._handles = []
        if self.exchange and not self._avg and self.world > 1:
            self.flat_grad.div_(self.world)
        if self.flat_param is None:
            for i, p in enumerate(self.params):
                o, n = self.slices[i]
                p.grad = self.flat_grad[o:o + n].view_as(p)

    def pack_all(self):
        """Pack every bucket into the flat gradient buffer without communicating (graph-captured steps pack
        inside the graph and exchange afterwards)."""
        for b in range(len(self.buckets)):
            self._pack(b)
            self._pending[b] = 0

    def exchange_packed(self):
        """All-reduce + average an already packed flat gradient buffer (one collective: the PSD nets' gradients
        are well under a megabyte, i.e. latency-bound)."""
        if self.exchange:
            dist.all_reduce(self.flat_grad, op=self._reduce_op(), group=self.group)
            if not self._avg and self.world > 1:
                self.flat_grad.div_(self.world)

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def broadcast_parameters(module, src=0, group=None):
    """One-time parameter/buffer broadcast from rank 0 (what torch DDP does at wrap time)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)

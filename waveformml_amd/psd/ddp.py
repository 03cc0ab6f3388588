"""Data-parallel gradient exchange for the PSD step: one process per GPU, RCCL over xGMI.

The reference gets data parallelism from Lightning's DDPPlugin -> torch DDP -> NCCL
(src/utils/util.py:228-239; no explicit collective call sites, SURVEY.md 2 "Parallelism inventory").
Events are independent units: each rank runs its own shard of the global batch, the
only exchange is the gradient all-reduce.  The PSD nets are tiny (0.03-1 M parameters), so the
exchange is latency-bound: all gradients live in ONE flat fp32 buffer cut into a few contiguous
buckets in reverse layer order; a bucket's all-reduce is launched asynchronously (RCCL runs it on its
own stream) the moment its last gradient has been accumulated, so it overlaps the rest of backward.
BatchNorm statistics stay per rank, exactly as under the reference's DDP (no SyncBN).
"""
import torch
import torch.distributed as dist


class FlatGradAllReducer(object):
    def __init__(self, parameters, n_buckets=2, process_group=None, world_size=None):
        self.params = [p for p in parameters if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        self.group = process_group
        self.world = world_size if world_size is not None else (
            dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1)
        dev, dtype = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=dtype, device=dev)
        # reverse order: the LAST layer's gradients are produced first and sit at the front
        order = list(reversed(range(len(self.params))))
        off = 0
        self.slices = {}
        for i in order:
            p = self.params[i]
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            self.slices[i] = (off, n)
            off += n
        # contiguous buckets of roughly equal size over that order
        n_buckets = max(1, min(n_buckets, len(self.params)))
        target = total / n_buckets
        self.buckets = []          # (start, end, [param indices])
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
        if self.world > 1:
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self.reset()

    def _make_hook(self, i):
        def hook(param):
            if param.grad.data_ptr() != self.flat.data_ptr() + self.slices[i][0] * self.flat.element_size():
                # something re-pointed .grad (e.g. zero_grad(set_to_none=True)): copy into the flat buffer
                off, n = self.slices[i]
                self.flat[off:off + n].copy_(param.grad.reshape(-1))
                param.grad = self.flat[off:off + n].view_as(param)
            b = self.bucket_of[i]
            self._pending[b] -= 1
            if self._pending[b] == 0:
                s, e, _ = self.buckets[b]
                self._handles.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group,
                                                     async_op=True))
        return hook

    def reset(self):
        """Call before each backward: zero the flat gradient buffer and re-arm the buckets."""
        self.flat.zero_()
        for i, p in enumerate(self.params):
            off, n = self.slices[i]
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + off * self.flat.element_size():
                p.grad = self.flat[off:off + n].view_as(p)
        self._pending = [len(idxs) for (_, _, idxs) in self.buckets]
        self._handles = []

    def finish(self):
        """Call after backward, before optimizer.step(): wait for the exchanges, average."""
        if self.world <= 1:
            return
        # a parameter that received no gradient never fires its hook: flush whatever is left
        for b, left in enumerate(self._pending):
            if left > 0:
                s, e, _ = self.buckets[b]
                self._handles.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group,
                                                     async_op=True))
                self._pending[b] = 0
        for h in self._handles:
            h.wait()
        self._handles = []
        self.flat.div_(self.world)

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

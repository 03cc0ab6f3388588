"""``FlatSGD``: torch.optim.SGD whose update of ONE flat fp32 GPU parameter is a single HIP launch
(include/wfsparse.h, wfs_sgd_step).  Drop-in for the optimizer the reference's configs name
(``"optimizer_class": "optim.SGD"``, config/examples/GEP.json:51-69; built at src/engineering/LitPSD.py:60-76):
same constructor, same ``state_dict`` layout (``momentum_buffer``), same arithmetic; anything it does not cover
(several tensors, CPU tensors, ``maximize``, closures) falls back to torch's own step.

The learning rate lives in a device scalar, so a scheduler can change ``param_groups[i]["lr"]`` between replays of a
captured HIP graph: call ``sync_hyperparameters()`` (psd/graph.GraphedTrainStep does) and the next replay uses it.
"""
import torch

from .. import _lib


class FlatSGD(torch.optim.SGD):
    def _fast_groups(self):
        for g in self.param_groups:
            ps = g["params"]
            if (len(ps) != 1 or not ps[0].is_cuda or ps[0].dtype != torch.float32 or not ps[0].is_contiguous()
                    or g.get("maximize", False) or g.get("differentiable", False)):
                return False
        return True

    def sync_hyperparameters(self):
        """Copy each group's current ``lr`` to its device scalar if it changed (one tiny fill, only then)."""
        for g in self.param_groups:
            dev = g["params"][0].device
            if g.get("_lr_dev") is None or g["_lr_dev"].device != dev:
                g["_lr_dev"] = torch.empty((1,), dtype=torch.float32, device=dev)
                g["_lr_host"] = None
            if g["_lr_host"] != float(g["lr"]):
                g["_lr_dev"].fill_(float(g["lr"]))
                g["_lr_host"] = float(g["lr"])

    def state_dict(self):
        sd = super().state_dict()
        for g in sd["param_groups"]:
            g.pop("_lr_dev", None)
            g.pop("_lr_host", None)
        sd["state"] = {k: {n: v for n, v in st.items() if n != "_fresh"} for k, st in sd["state"].items()}
        return sd

    def mark_fresh(self):
        """The momentum buffers exist (a captured graph holds their addresses) but carry no history: the next EAGER
        step treats them as torch treats a missing buffer (buf = g, no dampening).  Only matters with dampening != 0 --
        with dampening 0 a zeroed buffer gives the same first step."""
        for g in self.param_groups:
            if g["dampening"] != 0 and g["momentum"] != 0:
                for p in g["params"]:
                    if "momentum_buffer" in self.state.get(p, {}):
                        self.state[p]["_fresh"] = True

    def has_fresh(self):
        return any(st.get("_fresh", False) for st in self.state.values())

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None or not self._fast_groups():
            return super().step(closure)
        lib = _lib.load()
        self.sync_hyperparameters()
        for g in self.param_groups:
            p = g["params"][0]
            if p.grad is None:
                continue
            grad = p.grad
            if grad.is_sparse or grad.dtype != torch.float32 or not grad.is_contiguous():
                return super().step(closure)
            state = self.state[p]
            buf, first = state.get("momentum_buffer"), False
            if g["momentum"] != 0 and buf is None:
                buf = state["momentum_buffer"] = torch.empty_like(p, memory_format=torch.contiguous_format)
                first = True
            if state.pop("_fresh", False) and not torch.cuda.is_current_stream_capturing():
                first = True
            _lib.check(lib.wfs_sgd_step(_lib.ptr(p), _lib.ptr(grad), _lib.ptr(buf) if g["momentum"] != 0 else None,
                                        p.numel(), _lib.ptr(g["_lr_dev"]), float(g["momentum"]), float(g["dampening"]),
                                        float(g["weight_decay"]), 1 if g["nesterov"] else 0, 1 if first else 0,
                                        _lib.stream_ptr()))
        return None

"""``SPConvNet``: host-side mirror of the reference's PSD network wrapper
(src/models/SPConvNet.py:29-157).  Builds the sparse stack + linear head from either an explicit
``algorithm`` list or ``hparams``, wraps ``[coords, feats]`` into a SparseConvTensor (batch column
moved first), runs the stack, flattens and applies the head.

The operator package is whatever module the config's ``imports`` binds to the key ``spconv``
(waveformml_amd.spconv on the GPU; tests bind the CPU oracle to compare).
"""
import logging

import torch
from torch import nn

from .blocks import LinearBlock, SparseConv2DBlock, SparseConv2DPreserve
from .config import DictionaryUtility, ModuleUtility


class SPConvPreserveNet(nn.Module):
    """Mirror of the reference's ``SPConvPreserveNet`` (src/models/SPConvNet.py:8-25): a SparseConv2DPreserve stack
    from 2 * n_samples channels down to n_type on the 14 x 11 grid; the output is the per-ROW feature matrix [N, n_type]
    (per-segment predictions; config/examples/IoniClassifierCNN.json).  The operator package is the module the config's
    ``imports`` binds to ``spconv`` (default: waveformml_amd.spconv)."""

    def __init__(self, config):
        super().__init__()
        self.log = logging.getLogger(__name__)
        self.system_config = config.system_config
        self.net_config = config.net_config
        self.nsamples = self.system_config.n_samples
        self.ntype = self.system_config.n_type
        self.modules_util = ModuleUtility(self.net_config.imports)
        if "spconv" in self.modules_util.modules:
            self.spconv = self.modules_util.modules["spconv"]
        else:
            import waveformml_amd.spconv as sp
            self.spconv = sp
        hparams = DictionaryUtility.to_dict(self.net_config.hparams.conv_params)
        self.model = SparseConv2DPreserve(self.spconv, self.nsamples * 2, self.ntype, self.net_config.hparams.n_conv,
                                          **hparams)
        self.spatial_size = [14, 11]
        self.register_buffer("permute_tensor", torch.LongTensor([2, 0, 1]), persistent=False)   # batch index first
        self.batch_size_hint = None            # an upper bound on the events of a batch (captured steps, psd/graph.py)

    def forward(self, x, batch_size=None):
        coords, feats = x[0], x[1]
        if batch_size is None:
            batch_size = self.batch_size_hint
        if batch_size is None:
            batch_size = int(coords[-1, -1]) + 1          # one device->host read, as the reference's
        handed = getattr(self, "batch_first_indices", None)
        if handed is not None and handed[0] is coords:
            indices = handed[1]                # psd/graph.py wrote the batch-first columns next to the coordinates
        else:
            indices = coords[:, self.permute_tensor].contiguous()
        st = self.spconv.SparseConvTensor(feats, indices, self.spatial_size, batch_size)
        if len(x) > 2 and x[2] is not None:    # [coords, feats, n_valid]: rows beyond n_valid[0] are padding
            st.n_valid = x[2]
        return self.model(st).features


class SPConvNet(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.log = logging.getLogger(__name__)
        self.system_config = config.system_config
        self.net_config = config.net_config
        self.nsamples = self.system_config.n_samples
        self.ntype = self.system_config.n_type
        self.modules_util = ModuleUtility(self.net_config.imports)
        self.spconv = self.modules_util.retrieve_module("spconv")
        self.sequence_class = self.modules_util.retrieve_class(self.net_config.sequence_class)
        self.batch_size_hint = None
        self.batch_events = None              # (coords, event offsets) written by a captured step's hand-over launch
        self._build()
        net_type = getattr(self.net_config, "net_type", "2DConvolution")
        if net_type == "3DConvolution":
            self.ndim = 3
            self.spatial_size = [14, 11, int(self.nsamples)]
            self.register_buffer("permute_tensor", torch.LongTensor([3, 0, 1, 2]), persistent=False)
        else:
            if net_type != "2DConvolution":
                self.log.warning("Warning: unknown net_type in net_config: {}".format(net_type))
            self.ndim = 2
            self.spatial_size = [14, 11]
            self.register_buffer("permute_tensor", torch.LongTensor([2, 0, 1]), persistent=False)

    # reference SPConvNet.forward, :54-69
    def forward(self, x, batch_size=None):
        coords, feats = x[0], x[1]
        if hasattr(self, "waveformLayer"):
            feats = self.waveformLayer(feats.unsqueeze(1)).squeeze(1)
        if batch_size is None:
            batch_size = getattr(self, "batch_size_hint", None)
        if batch_size is None:
            batch_size = int(coords[-1, -1]) + 1          # one device->host read, as the reference's
        handed = getattr(self, "batch_first_indices", None)
        if handed is not None and handed[0] is coords:
            indices = handed[1]            # psd/graph.py's hand-over kernel already wrote the batch-first columns
        else:
            make = lambda: coords[:, self.permute_tensor].contiguous()          # noqa: E731
            reused = getattr(getattr(self.spconv, "ops", None), "reused", None)
            indices = reused("batch_first", coords, make) if reused is not None else make()
        st = self.spconv.SparseConvTensor(feats, indices, self.spatial_size,
                                          batch_size)
        if len(x) > 2 and x[2] is not None:       # [coords, feats, n_valid]: rows beyond n_valid[0] are padding
            st.n_valid = x[2]
        ev = getattr(self, "batch_events", None)
        if ev is not None and ev[0] is coords and handed is not None and handed[0] is coords:
            st.events = ev[1]                     # first row of every event: what the event-local rulebook build starts from
        fsp = getattr(self.spconv, "functional", None)
        if fsp is not None and hasattr(fsp, "sparse_head") and hasattr(self.sparseModel, "run"):
            # ToDense -> view -> Linear off the sparse rows (csrc/shead.hip) when the tail has that shape
            layers = list(self.linear)
            out = self.sparseModel.run(st, list(self.sparseModel._modules.values()), stop_before_dense=True)
            if hasattr(out, "features") and hasattr(out, "dense"):
                if fsp.can_use_sparse_head(out, layers):
                    return fsp.sparse_head(out, layers[0])
                out = out.dense()
        else:
            out = self.sparseModel(st)
        out = out.view(-1, self.n_linear)
        if fsp is not None and hasattr(fsp, "head_forward"):
            return fsp.head_forward(out, self.linear)         # per layer: streaming / matrix-core HIP kernels or torch
        head_dtype = next(self.linear.parameters()).dtype
        if out.dtype != head_dtype:          # bf16 activations, fp32 master weights in the dense head
            out = out.to(head_dtype)
        return self.linear(out)

    def _build(self):
        if hasattr(self.net_config, "algorithm"):
            self._from_algorithm(self.net_config.algorithm)
        elif hasattr(self.net_config, "hparams"):
            try:
                self._from_hparams(self.net_config.hparams)
            except AssertionError as e:
                raise AssertionError("Parameters {0} \nlead to error : {1}".format(
                    DictionaryUtility.to_dict(self.net_config.hparams), e))
        else:
            raise IOError("net_config must contain one of either 'algorithm' or 'hparams'")

    # reference get_algorithm, :124-157: [optional nn.Conv1d front end] sparse layers ... "nn.Linear" head
    def _from_algorithm(self, algorithm):
        waveform, sparse, linear = [], [], []
        in_wf = False
        for i, f in enumerate(algorithm):
            if i == 0 and isinstance(f, str) and f == "nn.Conv1d":
                in_wf = True
                waveform.append(f)
                continue
            if in_wf:
                if isinstance(f, str) and not f.startswith("nn."):
                    in_wf = False
                    sparse.append(f)
                else:
                    waveform.append(f)
                continue
            if isinstance(f, str) and f == "nn.Linear":
                linear = algorithm[i:]
                break
            sparse.append(f)
        if waveform:
            self.waveformLayer = nn.Sequential(*self.modules_util.create_class_instances(waveform))
        self.sparseModel = self.sequence_class(*self.modules_util.create_class_instances(sparse))
        self.linear = nn.Sequential(*self.modules_util.create_class_instances(linear))
        self.n_linear = linear[1][0]

    # reference create_algorithm, :71-109 (2-D only there as well)
    def _from_hparams(self, hparams):
        for rq in ["n_dil", "n_conv", "n_lin", "out_planes"]:
            if not hasattr(hparams, rq):
                raise IOError(rq + " is required to create the sparse conv algorithm.")
        size = [14, 11, int(self.nsamples * 2)]
        if hparams.n_dil > 0:
            from .tcn import TemporalConvNet
            params = DictionaryUtility.to_dict(hparams.wf_params) if hasattr(hparams, "wf_params") else {}
            self.waveformLayer = TemporalConvNet(1, [1] * hparams.n_dil, **params)
        params = DictionaryUtility.to_dict(hparams.conv_params) if hasattr(hparams, "conv_params") else {}
        block = SparseConv2DBlock(self.spconv, size[2], hparams.out_planes, hparams.n_conv, size, True, **params)
        self.sparseModel = block.func
        self.schedule = block.schedule
        flat = 1
        for s in block.out_size:
            flat *= s
        self.n_linear = flat
        head = LinearBlock(flat, self.ntype, hparams.n_lin)
        self.linear = head.func

"""``SingleEndedZConv`` / ``SingleEndedEZConv``: host-side mirrors of reference src/models/SingleEndedZConv.py:10-45 and
src/models/SingleEndedEZConv.py:13-67 -- the per-segment (z position / energy) regression nets: [N, 2T] waveform rows on the
14 x 11 grid -> dense [B, out, 14, 11].

The reference hard-imports ``spconv``; here the module registered under the plugin key ``spconv`` in
``net_config.imports`` is used when there is one (so tests can bind the CPU oracle), else ``waveformml_amd.spconv``.
"""
import numpy as np
import torch
from torch import nn

from .config import DictionaryUtility, ModuleUtility
from .zblocks import Pointwise2DForZ, SparseConv2DForEZ, SparseConv2DForZ


class SingleEndedZConv(nn.Module):
    def __init__(self, config):
        super().__init__()
        if config.net_config.net_type != "2DConvolution":
            raise IOError("config.net_config.net_type must be 2DConvolution")
        self.system_config, self.net_config = config.system_config, config.net_config
        self.nsamples = self.system_config.n_samples
        self.modules_util = ModuleUtility(self.net_config.imports)
        if "spconv" in self.modules_util.modules:
            self.spconv = self.modules_util.modules["spconv"]
        else:
            import waveformml_amd.spconv as sp
            self.spconv = sp
        if not hasattr(self.net_config, "algorithm"):
            setattr(self.net_config, "algorithm", "conv")
        self.version = getattr(self.net_config, "version", 0)
        algo, hp = self.net_config.algorithm, self.net_config.hparams
        if algo in ("conv", "features"):
            planes = self.nsamples * 2 if algo == "conv" else self.nsamples
            if self.version == 0:
                self.model = SparseConv2DForZ(self.spconv, planes, **DictionaryUtility.to_dict(hp.conv))
            else:
                self.model = SparseConv2DForEZ(self.spconv, planes, out_planes=1, **DictionaryUtility.to_dict(hp))
        elif algo == "point":
            self.model = Pointwise2DForZ(self.spconv, self.nsamples * 2, **DictionaryUtility.to_dict(hp.point))
        else:
            raise IOError("unknown net_config.algorithm %r" % (algo,))
        self.spatial_size = np.array([14, 11])
        self.register_buffer("permute_tensor", torch.LongTensor([2, 0, 1]), persistent=False)   # batch index first

    def forward(self, x, batch_size=None):
        coords, feats = x[0], x[1]
        if batch_size is None:
            batch_size = int(coords[-1, -1]) + 1
        st = self.spconv.SparseConvTensor(feats, coords[:, self.permute_tensor].contiguous(), self.spatial_size,
                                          batch_size)
        return self.model(st)


class SingleEndedEZConv(nn.Module):
    """z + energy: one SparseConv2DForEZ with 2 output planes; or, with ``net_config.z_weights`` / ``z_config``, a frozen
    LitZ model for z next to a 1-plane energy stack, concatenated [energy, z] as the reference does (:55-62)."""

    def __init__(self, config):
        super().__init__()
        if config.net_config.net_type != "2DConvolution":
            raise IOError("config.net_config.net_type must be 2DConvolution")
        self.system_config, self.net_config = config.system_config, config.net_config
        self.nsamples = self.system_config.n_samples
        self.modules_util = ModuleUtility(self.net_config.imports)
        if "spconv" in self.modules_util.modules:
            self.spconv = self.modules_util.modules["spconv"]
        else:
            import waveformml_amd.spconv as sp
            self.spconv = sp
        self.use_z_model = hasattr(self.net_config, "z_weights")
        if self.use_z_model:
            if not hasattr(self.net_config, "z_config"):
                raise ValueError("if specifying z_weights, you must also specify corresponding z_config")
            from .config import load_config
            from .litz import LitZ
            from .trainer import load_from_checkpoint
            self.z_model = load_from_checkpoint(self.net_config.z_weights, load_config(self.net_config.z_config), LitZ)
            self.z_model.eval()              # LightningModule.freeze(); a later .train() of the parent reaches it again,
                                             # in the reference too
            for p in self.z_model.parameters():
                p.requires_grad_(False)
        if not hasattr(self.net_config, "algorithm"):
            setattr(self.net_config, "algorithm", "conv")
        algo = self.net_config.algorithm
        if algo in ("conv", "features"):
            planes = self.nsamples * 2 if algo == "conv" else self.nsamples
            extra = dict(out_planes=1) if self.use_z_model else {}
            self.model = SparseConv2DForEZ(self.spconv, planes, **extra, **DictionaryUtility.to_dict(self.net_config.hparams))
        self.spatial_size = np.array([14, 11])
        self.register_buffer("permute_tensor", torch.LongTensor([2, 0, 1]), persistent=False)   # batch index first

    def forward(self, x, batch_size=None):
        coords, feats = x[0], x[1]
        if batch_size is None:
            batch_size = int(coords[-1, -1]) + 1
        st = self.spconv.SparseConvTensor(feats, coords[:, self.permute_tensor].contiguous(), self.spatial_size,
                                          batch_size)
        out = self.model(st)
        if self.use_z_model:
            out = torch.cat((out, self.z_model(x)), dim=1)
        return out

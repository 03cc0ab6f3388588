"""Alias module so that reference-style config strings (``"net_class": "SPConvNet.SPConvNet"`` with
``"waveformml_amd.psd.SPConvNet"`` in ``imports``; cf. reference config/examples/GEP.json:24-29)
resolve here."""
from .net import SPConvNet, SPConvPreserveNet  # noqa: F401

"""Alias module: ``"imports": ["waveformml_amd.psd.LitPSD"], "run_class": "LitPSD"`` (cf. reference
config/examples/GEP.json:2-8)."""
from .lit import LitPSD  # noqa: F401

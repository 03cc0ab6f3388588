"""Alias module: ``"imports": ["waveformml_amd.psd.LitEZ"], "run_class": "LitEZ"`` (reference src/engineering/LitEZ.py)."""
from .litz import LitEZ  # noqa: F401

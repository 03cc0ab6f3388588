"""Alias module: ``"imports": ["waveformml_amd.psd.LitSegClassifier"], "run_class": "LitSegClassifier"`` (cf. reference
config/examples/IoniClassifierCNN.json:2-8)."""
from .litseg import LitSegClassifier  # noqa: F401

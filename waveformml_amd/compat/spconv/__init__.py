"""Alias so that unmodified reference code (`import spconv`, reference src/models/SPConvBlocks.py:4)
resolves to the MI355X implementation: put <repo>/waveformml_amd/compat on PYTHONPATH."""
from waveformml_amd.spconv import *  # noqa: F401,F403
from waveformml_amd.spconv import functional, ops, conv, modules, tensor, __version__  # noqa: F401

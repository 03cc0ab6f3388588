"""waveformml_amd -- MI355X-native sparse-convolution PSD training path for WaveformML.

Sub-packages
    spconv   the spconv 1.2.1 operator surface the reference imports, over libwfsparse.so (HIP)
    psd      host-side mirror of the reference's LitPSD / SPConvNet / PSDDataModule interface
"""
import sys

__version__ = "0.1.0"


def install_as_spconv():
    """Register waveformml_amd.spconv under the module name ``spconv`` (reference: `import spconv`)."""
    from . import spconv as _sp
    sys.modules.setdefault("spconv", _sp)
    return sys.modules["spconv"]

"""datum_amd -- MI355X-native replacement of datum's FFT-ocean compute path.

The product is the C-ABI shared library built from datum_amd/csrc (include/datum_ocean_hip.h) and the
C++ host shim in datum_amd/host that mirrors src/renderer/ocean.h.  This Python package is only a thin
ctypes loader for tests and bench.py; it contains no compute and no CPU fallback: loading fails loudly
when the HIP library has not been built.
"""

from . import capi  # noqa: F401

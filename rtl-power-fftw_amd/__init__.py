"""MI355X-native power-spectrum engine: drop-in for the FFT-and-accumulate
worker of rtl_power_fftw (``Datastore::fftThread``,
/root/reference/src/datastore.cxx:48-96).

The product is the C-ABI shared library ``librpf_engine.so`` (hand-written HIP
for gfx950, see ``csrc/`` and ``include/rpf_engine.h``).  This Python package is
only the host-side mirror used by tests and ``bench.py``: a ``Datastore``-shaped
class over that C-ABI, the synthetic IQ source, and the output formatter.
"""
from ._lib import RPFError, ReturnValue, build, lib_path, load  # noqa: F401
from .datastore import Datastore, Params  # noqa: F401
from . import sharding, synth  # noqa: F401

"""MI355X-native TSM training hot path for background-debiased video class-incremental learning.

Import as ``bdvcil_amd`` (see ``bdvcil_amd.py`` at the repo root).  Layout:

* ``csrc/``      hand-written HIP kernels for gfx950 + the C ABI (``include/bdvcil_hip.h``)
* ``_lib``       ctypes binding (lazy, per process, no fallback)
* ``kernels``    tensor-level wrappers (shape checks + launch on torch's current stream)
"""
from . import _lib, kernels  # noqa: F401

__version__ = '0.1.0'

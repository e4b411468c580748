"""gmat_amd — MI355X-native pixel-transform path (libgpuscale + GPU filters) of NVIDIA/GMAT.

The product is the C-ABI shared library ``gmat_amd/lib/libgmat_hip.so`` (HIP kernels for gfx950 +
a C++ host layer, declared in ``include/gmat_hip.h``).  This package only binds it with ctypes,
the way the reference binds its own ``CSwscale.so`` (metrans/python/swscale.py:1-23).

There is no CPU fallback: importing :mod:`gmat_amd.lib` raises if the library has not been built.
"""
from .lib import load, lib_path, GmatError  # noqa: F401

__all__ = ["load", "lib_path", "GmatError"]

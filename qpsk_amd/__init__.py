"""qpsk_amd -- MI355X (gfx950) implementation of the MonsieurETM/QPSK receive path.

The product is the C-ABI shared library ``qpsk_amd/libqpsk_hip.so`` (sources in ``qpsk_amd/csrc``,
interface in ``include/qpsk_hip.h`` and ``include/qpsk_dropin.h``).  This Python package is only
plumbing around it for the tests and the benchmark: it loads the library with ctypes and passes
device pointers of torch tensors (PyTorch is used for device memory and streams, nothing else).

There is no CPU fallback: without the built library, or without a GPU, everything raises.
"""
from .lib import (QpskError, Modem, MultiJob, Params, TIMING_FIXED, TIMING_FFT, TIMING_HIST, build, lib_path, load,  # noqa: F401
                  TAU, version)

"""fastq_utils_amd - MI355X-native implementation of fastq_utils' per-read hot path.

The work is done by libfqgpu.so (hand-written HIP kernels for gfx950 behind the C-ABI declared
in include/fqg.h).  This package is the thin Python binding used by the tests and bench.py; the
command-line programs in bin/ bind the same library from C++.
"""
from . import abi  # noqa: F401
from .abi import Context, Accumulator, FileState, LibraryMissing  # noqa: F401

__all__ = ["abi", "Context", "Accumulator", "FileState", "LibraryMissing"]

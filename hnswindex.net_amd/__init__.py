"""
hnswindex.net_amd -- MI355X-native distance backend behind HNSWIndex.Net's own surfaces.

`Index` mirrors the reference's Python class (bindings/bindings.py:172-521); `lib` is the
ctypes handle of the C-ABI library (include/hnsw_mi355x.h).  Distances are computed only by
the HIP kernels in csrc/; nothing in this package falls back to the CPU.
"""
from .bindings import Index, DeviceBackend, lib, last_error, host_parallelism, set_options, default_options, Options, LIB_PATH  # noqa: F401
from . import distributed  # noqa: F401

"""gzp_amd -- MI355X-native per-block encoder for gzp's ParCompress<Bgzf/Mgzip> hot path.

The package holds only what the path needs: the HIP kernels + C ABI (csrc/, include/gzpx.h),
their ctypes binding (_native.py) and the host-side mirror of the reference's builder/writer
interface (par.py).  There is no CPU fallback anywhere in here.
"""
from ._native import (COMPAT_1_10, COMPAT_1_24, FORMAT_BGZF, FORMAT_MGZIP, Compressor, Context,
                      GzpxError, GzpxLib, crc32, load)

__all__ = ["Context", "Compressor", "GzpxError", "GzpxLib", "crc32", "load", "FORMAT_BGZF",
           "FORMAT_MGZIP", "COMPAT_1_10", "COMPAT_1_24"]

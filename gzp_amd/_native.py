"""ctypes binding of the C ABI in include/gzpx.h (gzp_amd/lib/libgzpx.so, built by hipcc for gfx950).

There is no fallback: if the HIP library is missing or no GPU is present, loading / context
creation raises -- the product path never routes through a CPU implementation.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgzpx.so")

OK = 0
ERR_INVALID_ARG = 1
ERR_BUFFER_SIZE = 2
ERR_COMPRESSION_LEVEL = 3
ERR_INSUFFICIENT_SPACE = 4
ERR_BLOCK_SIZE_EXCEEDED = 5
ERR_DEVICE = 6
ERR_NO_DEVICE = 7
ERR_UNSUPPORTED = 8
ERR_NUM_THREADS = 9
ERR_IO = 10
ERR_CHANNEL = 11
ERR_INVALID_HEADER = 12
ERR_INVALID_CHECK = 13
ERR_BAD_DATA = 14
ERR_BUSY = 15

SLAB_FULL_BLOCKS = 0
SLAB_LAST = 1
SLAB_FLUSH = 2

FORMAT_BGZF = 0
FORMAT_MGZIP = 1
COMPAT_1_24 = 0
COMPAT_1_10 = 1
STREAM_NONE = ctypes.c_void_p(-1).value  # GZPX_STREAM_NONE: the caller has synchronized, no stream dependency
N_STAGES = 9
INFLATE_SEG, INFLATE_WAVE = 0, 1

EXPORTS = [
    "gzpx_config_default", "gzpx_ctx_create", "gzpx_ctx_destroy", "gzpx_slab_bound",
    "gzpx_compress_slab", "gzpx_compress_slab_device", "gzpx_encode_block",
    "gzpx_alloc_compressor", "gzpx_deflate_compress", "gzpx_deflate_compress_bound",
    "gzpx_free_compressor", "gzpx_compressor_set_compat", "gzpx_crc32",
    "gzpx_ctx_set_profiling", "gzpx_ctx_last_stage_ms", "gzpx_stage_name", "gzpx_ctx_stage_kernel", "gzpx_debug_tokens",
    "gzpx_debug_set_flags", "gzpx_debug_redo_count", "gzpx_strerror", "gzpx_device_name", "gzpx_version",
    "gzpx_compress_slab_submit", "gzpx_compress_slab_submit_device", "gzpx_compress_slab_wait",
    "gzpx_compress_slab_event", "gzpx_crc32_checked", "gzpx_last_status", "gzpx_crc32_combine", "gzpx_adler32",
    "gzpx_adler32_checked", "gzpx_adler32_combine",
    "gzpx_par_create", "gzpx_par_write", "gzpx_par_write_chunked", "gzpx_par_flush", "gzpx_par_finish", "gzpx_par_destroy",
    "gzpx_par_last_error", "gzpx_par_create_pinned", "gzpx_par_reserve", "gzpx_par_commit", "gzpx_par_index", "gzpx_gzi_size",
    "gzpx_gzi_write", "gzpx_dctx_create", "gzpx_dctx_destroy", "gzpx_scan_blocks",
    "gzpx_decompress_blocks", "gzpx_decompress_blocks_device", "gzpx_decompress_blocks_submit",
    "gzpx_decompress_blocks_wait", "gzpx_alloc_decompressor", "gzpx_deflate_decompress",
    "gzpx_free_decompressor", "gzpx_pard_create", "gzpx_pard_read", "gzpx_pard_fill_buf", "gzpx_pard_consume", "gzpx_pard_destroy",
    "gzpx_pard_last_error", "gzpx_host_alloc", "gzpx_host_free", "gzpx_dctx_last_inflate_ms",
    "gzpx_debug_inflate", "gzpx_dctx_last_inflate_stage_ms", "gzpx_dctx_set_route", "gzpx_dctx_last_redo_count", "gzpx_synth_fastq_device", "gzpx_synth_ascii_device",
    "gzpx_ctx_active_compat", "gzpx_build_id", "gzpx_multi_create", "gzpx_multi_destroy", "gzpx_multi_devices", "gzpx_multi_compress_slab",
    "gzpx_multi_shard", "gzpx_multi_compress_slab_device",
]


class GzpxConfig(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int), ("format", ctypes.c_int), ("level", ctypes.c_int),
                ("compat", ctypes.c_int), ("buffer_size", ctypes.c_size_t),
                ("max_slab_bytes", ctypes.c_size_t)]


class GzpxParConfig(ctypes.Structure):
    _fields_ = [("format", ctypes.c_int), ("level", ctypes.c_int), ("compat", ctypes.c_int),
                ("device", ctypes.c_int), ("buffer_size", ctypes.c_size_t),
                ("num_threads", ctypes.c_size_t), ("batch_blocks", ctypes.c_size_t)]


class GzpxCheckInfo(ctypes.Structure):
    _fields_ = [("block", ctypes.c_size_t), ("found", ctypes.c_uint32), ("expected", ctypes.c_uint32)]


READ_FN = ctypes.CFUNCTYPE(ctypes.c_long, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t)
WRITE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t)


class GzpxError(RuntimeError):
    def __init__(self, code, msg, block=None):
        super().__init__("gzpx error %d: %s%s" % (code, msg, "" if block is None else " (block %d)" % block))
        self.code = code
        self.block = block


class GzpxLib:
    """All entry points of include/gzpx.h with argtypes set."""

    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise ImportError(
                "gzp_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
        self.path = path
        # PyTorch-ROCm bundles its own HIP runtime (same SONAME as /opt/rocm's).  Two copies in one
        # process fight over the device, so when torch is importable let it load first; libgzpx.so
        # then binds to the runtime that is already resident.
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch-less hosts use the system ROCm runtime
            pass
        L = self.L = ctypes.CDLL(path)
        vp, sz, i32, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32
        psz = ctypes.POINTER(ctypes.c_size_t)
        L.gzpx_config_default.restype = None
        L.gzpx_config_default.argtypes = [ctypes.POINTER(GzpxConfig), i32]
        L.gzpx_ctx_create.restype = i32
        L.gzpx_ctx_create.argtypes = [ctypes.POINTER(GzpxConfig), ctypes.POINTER(vp)]
        L.gzpx_ctx_destroy.restype = None
        L.gzpx_ctx_destroy.argtypes = [vp]
        L.gzpx_slab_bound.restype = sz
        L.gzpx_slab_bound.argtypes = [vp, sz]
        L.gzpx_build_id.restype = ctypes.c_char_p
        L.gzpx_build_id.argtypes = []
        L.gzpx_ctx_active_compat.restype = i32
        L.gzpx_ctx_active_compat.argtypes = [vp]
        L.gzpx_compress_slab.restype = i32
        L.gzpx_compress_slab.argtypes = [vp, vp, sz, i32, vp, sz, psz, vp, sz, psz]
        L.gzpx_compress_slab_device.restype = i32
        L.gzpx_compress_slab_device.argtypes = [vp, vp, sz, i32, vp, sz, psz, vp, sz, psz, vp]
        L.gzpx_encode_block.restype = i32
        L.gzpx_encode_block.argtypes = [vp, vp, sz, i32, vp, sz, psz]
        L.gzpx_alloc_compressor.restype = vp
        L.gzpx_alloc_compressor.argtypes = [i32]
        L.gzpx_deflate_compress.restype = sz
        L.gzpx_deflate_compress.argtypes = [vp, vp, sz, vp, sz]
        L.gzpx_deflate_compress_bound.restype = sz
        L.gzpx_deflate_compress_bound.argtypes = [vp, sz]
        L.gzpx_free_compressor.restype = None
        L.gzpx_free_compressor.argtypes = [vp]
        L.gzpx_compressor_set_compat.restype = i32
        L.gzpx_compressor_set_compat.argtypes = [vp, i32]
        L.gzpx_crc32.restype = u32
        L.gzpx_crc32.argtypes = [u32, vp, sz]
        L.gzpx_ctx_set_profiling.restype = i32
        L.gzpx_ctx_set_profiling.argtypes = [vp, i32]
        L.gzpx_ctx_last_stage_ms.restype = i32
        L.gzpx_ctx_last_stage_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
        L.gzpx_stage_name.restype = ctypes.c_char_p
        L.gzpx_stage_name.argtypes = [i32]
        L.gzpx_ctx_stage_kernel.restype = ctypes.c_char_p
        L.gzpx_ctx_stage_kernel.argtypes = [vp, i32]
        L.gzpx_debug_tokens.restype = i32
        L.gzpx_debug_tokens.argtypes = [vp, sz, vp, sz, psz, vp, psz]
        L.gzpx_debug_set_flags.restype = i32
        L.gzpx_debug_set_flags.argtypes = [vp, u32]
        L.gzpx_debug_redo_count.restype = i32
        L.gzpx_debug_redo_count.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32)]
        pu64 = ctypes.POINTER(ctypes.c_uint64)
        L.gzpx_compress_slab_submit.restype = i32
        L.gzpx_compress_slab_submit.argtypes = [vp, vp, sz, i32, vp, sz, pu64]
        L.gzpx_compress_slab_submit_device.restype = i32
        L.gzpx_compress_slab_submit_device.argtypes = [vp, vp, sz, i32, vp, sz, vp, pu64]
        L.gzpx_compress_slab_wait.restype = i32
        L.gzpx_compress_slab_wait.argtypes = [vp, ctypes.c_uint64, psz, vp, sz, psz]
        L.gzpx_compress_slab_event.restype = i32
        L.gzpx_compress_slab_event.argtypes = [vp, ctypes.c_uint64, ctypes.POINTER(vp)]
        L.gzpx_crc32_checked.restype = i32
        L.gzpx_crc32_checked.argtypes = [u32, vp, sz, ctypes.POINTER(u32)]
        L.gzpx_last_status.restype = i32
        L.gzpx_last_status.argtypes = []
        L.gzpx_strerror.restype = ctypes.c_char_p
        L.gzpx_strerror.argtypes = [i32]
        L.gzpx_device_name.restype = ctypes.c_char_p
        L.gzpx_device_name.argtypes = [vp]
        L.gzpx_version.restype = ctypes.c_char_p
        L.gzpx_version.argtypes = []
        L.gzpx_par_create.restype = i32
        L.gzpx_par_create.argtypes = [ctypes.POINTER(GzpxParConfig), WRITE_FN, vp, ctypes.POINTER(vp)]
        L.gzpx_par_create_pinned.restype = i32
        L.gzpx_par_create_pinned.argtypes = [ctypes.POINTER(GzpxParConfig), sz, WRITE_FN, vp, ctypes.POINTER(vp)]
        L.gzpx_par_write.restype = i32
        L.gzpx_par_write.argtypes = [vp, vp, sz]
        L.gzpx_par_write_chunked.restype = i32
        L.gzpx_par_write_chunked.argtypes = [vp, vp, sz, sz]
        L.gzpx_par_reserve.restype = i32
        L.gzpx_par_reserve.argtypes = [vp, ctypes.POINTER(vp), psz]
        L.gzpx_par_commit.restype = i32
        L.gzpx_par_commit.argtypes = [vp, sz]
        L.gzpx_par_index.restype = i32
        L.gzpx_par_index.argtypes = [vp, vp, sz, psz]
        L.gzpx_gzi_size.restype = sz
        L.gzpx_gzi_size.argtypes = [sz]
        L.gzpx_gzi_write.restype = i32
        L.gzpx_gzi_write.argtypes = [vp, sz, vp, sz, psz]
        L.gzpx_par_flush.restype = i32
        L.gzpx_par_flush.argtypes = [vp]
        L.gzpx_par_finish.restype = i32
        L.gzpx_par_finish.argtypes = [vp]
        L.gzpx_par_destroy.restype = None
        L.gzpx_par_destroy.argtypes = [vp]
        L.gzpx_par_last_error.restype = ctypes.c_char_p
        L.gzpx_par_last_error.argtypes = [vp]
        pinfo = ctypes.POINTER(GzpxCheckInfo)
        L.gzpx_dctx_create.restype = i32
        L.gzpx_dctx_create.argtypes = [i32, i32, ctypes.POINTER(vp)]
        L.gzpx_dctx_destroy.restype = None
        L.gzpx_dctx_destroy.argtypes = [vp]
        L.gzpx_scan_blocks.restype = i32
        L.gzpx_scan_blocks.argtypes = [i32, vp, sz, vp, vp, sz, psz, psz]
        L.gzpx_decompress_blocks.restype = i32
        L.gzpx_decompress_blocks.argtypes = [vp, vp, sz, vp, vp, sz, vp, sz, psz, pinfo]
        L.gzpx_decompress_blocks_submit.restype = i32
        L.gzpx_decompress_blocks_submit.argtypes = [vp, vp, sz, vp, vp, sz, vp, sz, ctypes.POINTER(ctypes.c_uint64)]
        L.gzpx_decompress_blocks_wait.restype = i32
        L.gzpx_decompress_blocks_wait.argtypes = [vp, ctypes.c_uint64, psz, pinfo]
        L.gzpx_decompress_blocks_device.restype = i32
        L.gzpx_decompress_blocks_device.argtypes = [vp, vp, sz, vp, vp, sz, vp, sz, psz, pinfo, vp]
        L.gzpx_alloc_decompressor.restype = vp
        L.gzpx_alloc_decompressor.argtypes = []
        L.gzpx_deflate_decompress.restype = i32
        L.gzpx_deflate_decompress.argtypes = [vp, vp, sz, vp, sz, psz]
        L.gzpx_free_decompressor.restype = None
        L.gzpx_free_decompressor.argtypes = [vp]
        L.gzpx_host_alloc.restype = vp
        L.gzpx_host_alloc.argtypes = [sz]
        L.gzpx_host_free.restype = None
        L.gzpx_host_free.argtypes = [vp]
        L.gzpx_dctx_last_inflate_ms.restype = i32
        L.gzpx_dctx_last_inflate_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
        L.gzpx_debug_inflate.restype = i32
        L.gzpx_debug_inflate.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_uint64)]
        L.gzpx_crc32_combine.restype = ctypes.c_uint32
        L.gzpx_crc32_combine.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64]
        L.gzpx_adler32_combine.restype = ctypes.c_uint32
        L.gzpx_adler32_combine.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64]
        L.gzpx_adler32.restype = ctypes.c_uint32
        L.gzpx_adler32.argtypes = [ctypes.c_uint32, vp, ctypes.c_size_t]
        L.gzpx_adler32_checked.restype = i32
        L.gzpx_adler32_checked.argtypes = [ctypes.c_uint32, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32)]
        L.gzpx_dctx_last_inflate_stage_ms.restype = i32
        L.gzpx_dctx_last_inflate_stage_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
        L.gzpx_dctx_set_route.restype = i32
        L.gzpx_dctx_set_route.argtypes = [vp, i32]
        L.gzpx_dctx_last_redo_count.restype = i32
        L.gzpx_dctx_last_redo_count.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32)]
        L.gzpx_synth_fastq_device.restype = i32
        L.gzpx_synth_fastq_device.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, vp]
        L.gzpx_multi_create.restype = i32
        L.gzpx_multi_create.argtypes = [ctypes.POINTER(GzpxConfig), ctypes.POINTER(ctypes.c_int), sz, ctypes.POINTER(vp)]
        L.gzpx_multi_destroy.restype = None
        L.gzpx_multi_destroy.argtypes = [vp]
        L.gzpx_multi_devices.restype = sz
        L.gzpx_multi_devices.argtypes = [vp]
        L.gzpx_multi_compress_slab.restype = i32
        L.gzpx_multi_compress_slab.argtypes = [vp, vp, sz, i32, vp, sz, psz, vp, sz, psz]
        L.gzpx_multi_shard.restype = i32
        L.gzpx_multi_shard.argtypes = [vp, sz, sz, psz, psz]
        L.gzpx_multi_compress_slab_device.restype = i32
        L.gzpx_multi_compress_slab_device.argtypes = [vp, ctypes.POINTER(vp), sz, i32, sz, vp, sz, psz, vp, sz, psz]
        L.gzpx_synth_ascii_device.restype = i32
        L.gzpx_synth_ascii_device.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, vp]
        L.gzpx_pard_create.restype = i32
        L.gzpx_pard_create.argtypes = [i32, i32, sz, READ_FN, vp, ctypes.POINTER(vp)]
        L.gzpx_pard_read.restype = i32
        L.gzpx_pard_read.argtypes = [vp, vp, sz, psz]
        L.gzpx_pard_fill_buf.restype = i32
        L.gzpx_pard_fill_buf.argtypes = [vp, ctypes.POINTER(vp), psz]
        L.gzpx_pard_consume.restype = i32
        L.gzpx_pard_consume.argtypes = [vp, sz]
        L.gzpx_pard_destroy.restype = None
        L.gzpx_pard_destroy.argtypes = [vp]
        L.gzpx_pard_last_error.restype = ctypes.c_char_p
        L.gzpx_pard_last_error.argtypes = [vp]

    def build_id(self):
        """gzpx_build_id(): which sources this library was built from (gzp_amd/build.py: source_id())."""
        return self.L.gzpx_build_id().decode()

    def strerror(self, code):
        return self.L.gzpx_strerror(code).decode()

    def check(self, code, block=None):
        if code != OK:
            raise GzpxError(code, self.strerror(code), block)


_default = None


def load():
    """The product library (HIP build).  Raises ImportError when it has not been built."""
    global _default
    if _default is None:
        _default = GzpxLib(LIB_PATH)
    return _default


def _u8(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


class Context:
    """gzpx_ctx: one device, one format/level/buffer_size -- the GPU-side `create_compressor`."""

    def __init__(self, format=FORMAT_BGZF, level=1, buffer_size=None, compat=COMPAT_1_24, device=0,
                 max_slab_bytes=1 << 30, lib=None):
        self.lib = lib or load()
        cfg = GzpxConfig()
        self.lib.L.gzpx_config_default(ctypes.byref(cfg), format)
        cfg.device = device
        cfg.level = level
        cfg.compat = compat
        if buffer_size is not None:
            cfg.buffer_size = buffer_size
        cfg.max_slab_bytes = max_slab_bytes
        self.cfg = cfg
        self.buffer_size = cfg.buffer_size
        self.format = format
        h = ctypes.c_void_p()
        self.lib.check(self.lib.L.gzpx_ctx_create(ctypes.byref(cfg), ctypes.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.L.gzpx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def device_name(self):
        return self.lib.L.gzpx_device_name(self.h).decode()

    def slab_bound(self, n):
        return int(self.lib.L.gzpx_slab_bound(self.h, n))

    def active_compat(self):
        """The GZPX_COMPAT_* rules the context really runs (levels 10-12: always the 1.10 ones)."""
        return int(self.lib.L.gzpx_ctx_active_compat(self.h))

    def n_blocks(self, n):
        return 1 if n == 0 else -(-n // self.buffer_size)

    @staticmethod
    def _mode(is_last):
        if is_last is True:
            return SLAB_LAST
        if is_last is False:
            return SLAB_FULL_BLOCKS
        return int(is_last)

    def compress_slab(self, data, is_last=True, return_block_sizes=False):
        """Host buffer in, framed bytes out (gzpx_compress_slab).  is_last: True = SLAB_LAST,
        False = SLAB_FULL_BLOCKS, or an explicit SLAB_* mode."""
        a = _u8(data)
        cap = self.slab_bound(a.size)
        out = np.empty(cap, dtype=np.uint8)
        nb_max = self.n_blocks(a.size)
        sizes = np.zeros(nb_max, dtype=np.uint32)
        out_len = ctypes.c_size_t(0)
        nb = ctypes.c_size_t(0)
        rc = self.lib.L.gzpx_compress_slab(self.h, a.ctypes.data, a.size, self._mode(is_last),
                                           out.ctypes.data, cap, ctypes.byref(out_len),
                                           sizes.ctypes.data, nb_max, ctypes.byref(nb))
        self.lib.check(rc, nb.value if rc == ERR_BLOCK_SIZE_EXCEEDED else None)
        res = out[:out_len.value].tobytes()
        if return_block_sizes:
            return res, sizes[:nb.value].copy()
        return res

    def compress_slab_device(self, d_in_ptr, in_len, d_out_ptr, out_cap, is_last=True, stream=None,
                             block_sizes=None):
        """Device pointers in/out (gzpx_compress_slab_device).  Returns (out_len, n_blocks)."""
        out_len = ctypes.c_size_t(0)
        nb = ctypes.c_size_t(0)
        bs_ptr, bs_n = (None, 0)
        if block_sizes is not None:
            bs_ptr, bs_n = block_sizes.ctypes.data, block_sizes.size
        rc = self.lib.L.gzpx_compress_slab_device(self.h, d_in_ptr, in_len, self._mode(is_last), d_out_ptr,
                                                  out_cap, ctypes.byref(out_len), bs_ptr, bs_n,
                                                  ctypes.byref(nb), stream)
        self.lib.check(rc, nb.value if rc == ERR_BLOCK_SIZE_EXCEEDED else None)
        return out_len.value, nb.value

    def encode_block(self, data, is_last=False):
        a = _u8(data)
        cap = self.slab_bound(a.size)
        out = np.empty(cap, dtype=np.uint8)
        out_len = ctypes.c_size_t(0)
        self.lib.check(self.lib.L.gzpx_encode_block(self.h, a.ctypes.data, a.size, int(is_last),
                                                    out.ctypes.data, cap, ctypes.byref(out_len)))
        return out[:out_len.value].tobytes()

    def set_profiling(self, on=True):
        self.lib.check(self.lib.L.gzpx_ctx_set_profiling(self.h, int(on)))

    def last_stage_ms(self):
        ms = (ctypes.c_float * N_STAGES)()
        self.lib.check(self.lib.L.gzpx_ctx_last_stage_ms(self.h, ms))
        out = {}
        for i in range(N_STAGES):
            name = self.lib.L.gzpx_ctx_stage_kernel(self.h, i).decode()
            if name != "-":
                out[name] = float(ms[i])
        return out

    def debug_set_flags(self, flags):
        self.lib.check(self.lib.L.gzpx_debug_set_flags(self.h, flags))

    def debug_redo_count(self):
        """Level 1: blocks of the last batch that k_mparse handed back to the dense kernels.  Levels 2-4 (round 5: the same
        scratch list serves them): blocks that k_parse_hc listed for k_match_hc_stale -- a sub-block with another min_len
        behind a start that k_match_hc_sparse compacted."""
        c = ctypes.c_uint32(0)
        self.lib.check(self.lib.L.gzpx_debug_redo_count(self.h, ctypes.byref(c)))
        return c.value

    def submit(self, in_ptr, in_len, out_ptr, out_cap, mode=SLAB_LAST):
        """gzpx_compress_slab_submit on raw host pointers (page-locked for DMA overlap); returns the
        ticket, or None when every slot is in flight."""
        t = ctypes.c_uint64(0)
        rc = self.lib.L.gzpx_compress_slab_submit(self.h, in_ptr, in_len, int(mode), out_ptr, out_cap,
                                                  ctypes.byref(t))
        if rc == ERR_BUSY:
            return None
        self.lib.check(rc)
        return t.value

    def submit_device(self, d_in_ptr, in_len, d_out_ptr, out_cap, mode=SLAB_LAST, after_stream=None):
        t = ctypes.c_uint64(0)
        rc = self.lib.L.gzpx_compress_slab_submit_device(self.h, d_in_ptr, in_len, int(mode), d_out_ptr,
                                                         out_cap, after_stream, ctypes.byref(t))
        if rc == ERR_BUSY:
            return None
        self.lib.check(rc)
        return t.value

    def wait(self, ticket, block_sizes=None):
        """gzpx_compress_slab_wait: (out_len, n_blocks)."""
        out_len = ctypes.c_size_t(0)
        nb = ctypes.c_size_t(0)
        bs_ptr, bs_n = (None, 0)
        if block_sizes is not None:
            bs_ptr, bs_n = block_sizes.ctypes.data, block_sizes.size
        rc = self.lib.L.gzpx_compress_slab_wait(self.h, ticket, ctypes.byref(out_len), bs_ptr, bs_n,
                                                ctypes.byref(nb))
        self.lib.check(rc, nb.value if rc == ERR_BLOCK_SIZE_EXCEEDED else None)
        return out_len.value, nb.value

    def debug_tokens(self, block):
        toks = np.empty(max(65536, int(self.buffer_size)), dtype=np.uint32)
        first = np.zeros(int(self.buffer_size) // 5000 + 4, dtype=np.uint32)
        n = ctypes.c_size_t(0)
        ns = ctypes.c_size_t(0)
        self.lib.check(self.lib.L.gzpx_debug_tokens(self.h, block, toks.ctypes.data, toks.size,
                                                    ctypes.byref(n), first.ctypes.data,
                                                    ctypes.byref(ns)))
        return toks[:n.value].copy(), first[:ns.value].copy()


class MultiContext:
    """gzpx_multi: one slab sharded over several devices, written out in order."""

    def __init__(self, devices, format=FORMAT_BGZF, level=1, buffer_size=None, compat=COMPAT_1_24,
                 max_slab_bytes=1 << 30, lib=None):
        self.lib = lib or load()
        cfg = GzpxConfig()
        self.lib.L.gzpx_config_default(ctypes.byref(cfg), format)
        cfg.level = level
        cfg.compat = compat
        if buffer_size is not None:
            cfg.buffer_size = buffer_size
        cfg.max_slab_bytes = max_slab_bytes
        self.buffer_size = cfg.buffer_size
        devs = (ctypes.c_int * len(devices))(*devices)
        h = ctypes.c_void_p()
        self.lib.check(self.lib.L.gzpx_multi_create(ctypes.byref(cfg), devs, len(devices), ctypes.byref(h)))
        self.h = h
        self.devices = list(devices)

    def compress_slab(self, data, mode=SLAB_LAST, return_block_sizes=False):
        a = _u8(data)
        nb_max = 1 if a.size == 0 else -(-a.size // self.buffer_size)
        cap = nb_max * (self.buffer_size + max(128, self.buffer_size // 10) + 28) + 128
        out = np.empty(cap, dtype=np.uint8)
        sizes = np.zeros(nb_max, dtype=np.uint32)
        out_len = ctypes.c_size_t(0)
        nb = ctypes.c_size_t(0)
        rc = self.lib.L.gzpx_multi_compress_slab(self.h, a.ctypes.data, a.size, int(mode), out.ctypes.data, cap,
                                                 ctypes.byref(out_len), sizes.ctypes.data, nb_max, ctypes.byref(nb))
        self.lib.check(rc, nb.value if rc == ERR_BLOCK_SIZE_EXCEEDED else None)
        res = out[:out_len.value].tobytes()
        return (res, sizes[:nb.value].copy()) if return_block_sizes else res

    def shard(self, in_len, g):
        """(offset, length) of the block range device g takes of a slab of in_len bytes."""
        off, n = ctypes.c_size_t(0), ctypes.c_size_t(0)
        self.lib.check(self.lib.L.gzpx_multi_shard(self.h, in_len, g, ctypes.byref(off), ctypes.byref(n)))
        return off.value, n.value

    def compress_slab_device(self, d_in_ptrs, in_len, d_out_ptr, out_cap, mode=SLAB_LAST, root=0, sync=True, wait_events=None):
        """Every range already on its own device (d_in_ptrs[g]); the shards are gathered device to device
        into d_out_ptr on devices[root].  Returns (out_len, block_sizes).

        Caller contract of gzpx_multi_compress_slab_device (include/gzpx.h): the ranges are COMPLETE and d_out is IDLE
        on entry -- the call takes no stream of the caller's to wait behind (it submits with GZPX_STREAM_NONE; since
        round 4 nothing is ordered behind the legacy default stream either).  `sync=True` (the default) makes that true
        for PyTorch callers, whose producers run asynchronously on torch's current stream: every device of the context
        is synchronised first.  Pass sync=False only when the inputs were produced synchronously.

        `wait_events` (round 6): the producers' own events (objects with .synchronize(): torch.cuda.Event, or anything
        else a runtime hands out) -- waited for INSTEAD of whole devices, so that what the caller has queued for slab
        k + 1 keeps running while slab k is compressed.  The full-device wait assumes that this context's device numbers
        are torch's ordinals (true when both see the same HIP_VISIBLE_DEVICES) and does nothing for producers of a
        runtime that is not torch: those callers pass their events, or synchronise themselves and say sync=False."""
        if wait_events is not None:
            for ev in wait_events:
                ev.synchronize()
            sync = False
        if sync:
            import sys
            torch = sys.modules.get("torch")  # (only callers that use torch have asynchronous producers to wait for)
            if torch is not None and torch.cuda.is_available():
                for d in sorted(set(self.devices)):
                    torch.cuda.synchronize(d)
        ptrs = (ctypes.c_void_p * len(d_in_ptrs))(*[ctypes.c_void_p(int(p) if p else None) for p in d_in_ptrs])
        nb_max = 1 if in_len == 0 else -(-in_len // self.buffer_size)
        sizes = np.zeros(nb_max, dtype=np.uint32)
        out_len = ctypes.c_size_t(0)
        nb = ctypes.c_size_t(0)
        rc = self.lib.L.gzpx_multi_compress_slab_device(self.h, ptrs, in_len, int(mode), root, d_out_ptr, out_cap,
                                                        ctypes.byref(out_len), sizes.ctypes.data, nb_max, ctypes.byref(nb))
        self.lib.check(rc, nb.value if rc == ERR_BLOCK_SIZE_EXCEEDED else None)
        return out_len.value, sizes[:nb.value].copy()

    def slab_bound(self, n):
        nb_max = 1 if n == 0 else -(-n // self.buffer_size)
        return nb_max * (self.buffer_size + max(128, self.buffer_size // 10) + 28) + 128

    def close(self):
        if getattr(self, "h", None):
            self.lib.L.gzpx_multi_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Compressor:
    """libdeflater::Compressor shape: alloc / deflate_compress / bound / free."""

    def __init__(self, level=1, compat=COMPAT_1_24, lib=None):
        self.lib = lib or load()
        self.h = self.lib.L.gzpx_alloc_compressor(level)
        if not self.h:
            raise GzpxError(ERR_COMPRESSION_LEVEL, "gzpx_alloc_compressor failed")
        self.lib.check(self.lib.L.gzpx_compressor_set_compat(self.h, compat))

    def deflate_compress(self, data, cap=None):
        a = _u8(data)
        if cap is None:
            cap = int(self.lib.L.gzpx_deflate_compress_bound(self.h, a.size))
        out = np.empty(max(cap, 1), dtype=np.uint8)
        n = self.lib.L.gzpx_deflate_compress(self.h, a.ctypes.data, a.size, out.ctypes.data, cap)
        if n == 0:
            raise GzpxError(ERR_INSUFFICIENT_SPACE, "gzpx_deflate_compress returned 0")
        return out[:n].tobytes()

    def close(self):
        if getattr(self, "h", None):
            self.lib.L.gzpx_free_compressor(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def synth_fastq_device(d_out_ptr, stream_offset, n, seed=20250927, stream=None, lib=None):
    """Fill device memory with bytes [stream_offset, stream_offset + n) of the synthetic FASTQ stream
    (BASELINE configs[3]); asynchronous on `stream`."""
    lib = lib or load()
    lib.check(lib.L.gzpx_synth_fastq_device(d_out_ptr, stream_offset, n, seed, stream))


def synth_ascii_device(d_out_ptr, stream_offset, n, seed=8, stream=None, lib=None):
    """Fill device memory with bytes [stream_offset, +n) of synth.ascii_random's stream (BASELINE configs[2])."""
    lib = lib or load()
    lib.check(lib.L.gzpx_synth_ascii_device(d_out_ptr, stream_offset, n, seed, stream))


def crc32(data, crc=0, lib=None):
    lib = lib or load()
    a = _u8(data)
    out = ctypes.c_uint32(0)
    lib.check(lib.L.gzpx_crc32_checked(crc, a.ctypes.data, a.size, ctypes.byref(out)))
    return int(out.value)


def crc32_combine(crc1, crc2, len2, lib=None):
    """Crc32::combine (src/check.rs:160-163): the CRC-32 of A || B from crc(A), crc(B), len(B)."""
    lib = lib or load()
    return int(lib.L.gzpx_crc32_combine(crc1, crc2, len2))


def adler32(data, adler=1, lib=None):
    """Adler32::update (src/check.rs:112-119) on the device."""
    lib = lib or load()
    a = _u8(data)
    out = ctypes.c_uint32(0)
    lib.check(lib.L.gzpx_adler32_checked(adler, a.ctypes.data, a.size, ctypes.byref(out)))
    return int(out.value)


def adler32_combine(adler1, adler2, len2, lib=None):
    """Adler32::combine (src/check.rs:121-127)."""
    lib = lib or load()
    return int(lib.L.gzpx_adler32_combine(adler1, adler2, len2))


class DContext:
    """gzpx_dctx: the GPU side of ParDecompress<Bgzf/Mgzip> (create_decompressor + decode_block +
    the per-block CRC check, for every block of a slab at once)."""

    def __init__(self, format=FORMAT_BGZF, device=0, lib=None):
        self.lib = lib or load()
        self.format = format
        h = ctypes.c_void_p()
        self.lib.check(self.lib.L.gzpx_dctx_create(device, format, ctypes.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.L.gzpx_dctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def scan_blocks(self, data):
        """(offsets uint64[], sizes uint32[], consumed) of the complete blocks in `data`."""
        a = _u8(data)
        nb = ctypes.c_size_t(0)
        used = ctypes.c_size_t(0)
        self.lib.check(self.lib.L.gzpx_scan_blocks(self.format, a.ctypes.data, a.size, None, None, 0,
                                                   ctypes.byref(nb), ctypes.byref(used)))
        offs = np.zeros(max(nb.value, 1), dtype=np.uint64)
        sizes = np.zeros(max(nb.value, 1), dtype=np.uint32)
        self.lib.check(self.lib.L.gzpx_scan_blocks(self.format, a.ctypes.data, a.size, offs.ctypes.data,
                                                   sizes.ctypes.data, offs.size, ctypes.byref(nb),
                                                   ctypes.byref(used)))
        return offs[:nb.value], sizes[:nb.value], used.value

    def _raise(self, rc, info):
        if rc == ERR_INVALID_CHECK:
            raise GzpxError(rc, "InvalidCheck { found: %d, expected: %d }" % (info.found, info.expected), info.block)
        raise GzpxError(rc, self.lib.strerror(rc), info.block)

    def decompress(self, data):
        """Host bytes of whole blocks in, inflated bytes out."""
        a = _u8(data)
        offs, sizes, used = self.scan_blocks(a)
        if used != a.size:
            raise GzpxError(ERR_INVALID_ARG, "trailing partial block (%d bytes)" % (a.size - used))
        isz = 0
        if offs.size:  # ISIZE fields of all blocks (little-endian u32 at the end of every member)
            ends = offs.astype(np.int64) + sizes.astype(np.int64)
            fields = a[(ends[:, None] - 4) + np.arange(4)].astype(np.uint64)
            isz = int((fields[:, 0] | (fields[:, 1] << np.uint64(8)) | (fields[:, 2] << np.uint64(16)) |
                       (fields[:, 3] << np.uint64(24))).sum())
        out = np.empty(max(isz, 1), dtype=np.uint8)
        out_len = ctypes.c_size_t(0)
        info = GzpxCheckInfo()
        rc = self.lib.L.gzpx_decompress_blocks(self.h, a.ctypes.data, a.size, offs.ctypes.data,
                                               sizes.ctypes.data, offs.size, out.ctypes.data, isz,
                                               ctypes.byref(out_len), ctypes.byref(info))
        if rc != OK:
            self._raise(rc, info)
        return out[:out_len.value].tobytes()

    def last_inflate_ms(self):
        ms = ctypes.c_float(0)
        self.lib.check(self.lib.L.gzpx_dctx_last_inflate_ms(self.h, ctypes.byref(ms)))
        return ms.value

    def last_inflate_stage_ms(self):
        """(k_inflate_seg ms, k_lzcopy + hand-backs ms) of the last launch on the decode / copy route."""
        ms = (ctypes.c_float * 2)()
        self.lib.check(self.lib.L.gzpx_dctx_last_inflate_stage_ms(self.h, ms))
        return ms[0], ms[1]

    def set_route(self, route):
        """INFLATE_SEG (default): k_inflate_seg + k_lzcopy, hand-backs to k_inflate; INFLATE_WAVE: k_inflate for every member."""
        self.lib.check(self.lib.L.gzpx_dctx_set_route(self.h, int(route)))

    def last_redo_count(self):
        """Members of the last launch that the decode / copy pair handed to k_inflate."""
        n = ctypes.c_uint32(0)
        self.lib.check(self.lib.L.gzpx_dctx_last_redo_count(self.h, ctypes.byref(n)))
        return n.value

    def debug_inflate(self, enable):
        """Switch the instrumented inflate kernels on/off; returns the counters of the last launch."""
        c = (ctypes.c_uint64 * 8)()
        self.lib.check(self.lib.L.gzpx_debug_inflate(self.h, int(enable), c))
        return list(c)

    def decompress_device(self, d_in_ptr, in_len, offsets, sizes, d_out_ptr, out_cap, stream=None):
        out_len = ctypes.c_size_t(0)
        info = GzpxCheckInfo()
        rc = self.lib.L.gzpx_decompress_blocks_device(self.h, d_in_ptr, in_len, offsets.ctypes.data,
                                                      sizes.ctypes.data, offsets.size, d_out_ptr, out_cap,
                                                      ctypes.byref(out_len), ctypes.byref(info), stream)
        if rc != OK:
            self._raise(rc, info)
        return out_len.value


class Decompressor:
    """libdeflater::Decompressor shape: deflate_decompress(raw, out_size)."""

    def __init__(self, lib=None):
        self.lib = lib or load()
        self.h = self.lib.L.gzpx_alloc_decompressor()

    def deflate_decompress(self, data, out_size):
        a = _u8(data)
        out = np.empty(max(out_size, 1), dtype=np.uint8)
        actual = ctypes.c_size_t(0)
        rc = self.lib.L.gzpx_deflate_decompress(self.h, a.ctypes.data, a.size, out.ctypes.data, out_size,
                                                ctypes.byref(actual))
        self.lib.check(rc)
        return out[:actual.value].tobytes()

    def close(self):
        if getattr(self, "h", None):
            self.lib.L.gzpx_free_decompressor(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""Python face of the ParCompress twin (gzp_amd/csrc/gzpx_par.{hpp,cpp}) -- same names and
behaviour as the reference's builder/writer API for the block formats:

    ParCompressBuilder / ParCompress   src/par/compress.rs:33-469
    ZBuilder                           src/lib.rs:181-275
    Compression                        flate2::Compression (src/lib.rs:81)
    GzpError                           src/lib.rs:114-163

    w = ParCompressBuilder(Bgzf).compression_level(Compression.fast()).from_writer(open(p, "wb"))
    w.write_all(data); w.finish()

All compression happens in the HIP library; this module only forwards bytes.
"""
import ctypes
import os

import numpy as np

from . import _native
from ._native import GzpxError as GzpError  # noqa: F401  (the error type users catch)

BUFSIZE = 64 * (1 << 10) * 2  # src/lib.rs:105
DICT_SIZE = 32768             # src/lib.rs:108


class Compression:
    """flate2::Compression"""

    def __init__(self, level=6):
        self._level = int(level)

    @staticmethod
    def new(level):
        return Compression(level)

    @staticmethod
    def none():
        return Compression(0)

    @staticmethod
    def fast():
        return Compression(1)

    @staticmethod
    def best():
        return Compression(9)

    def level(self):
        return self._level


class Bgzf:
    DEFAULT_BUFSIZE = 65280  # src/deflate.rs:583
    FORMAT = _native.FORMAT_BGZF


class Mgzip:
    DEFAULT_BUFSIZE = BUFSIZE  # src/lib.rs:330
    FORMAT = _native.FORMAT_MGZIP


class ParCompress:
    """`Write` + `ZWriter` (src/par/compress.rs:221-469, src/lib.rs:166-170)."""

    def __init__(self, cfg, writer, lib=None, pin=None):
        self._lib = lib or _native.load()
        self._writer = writer
        self._io_error = None

        def _cb(user, data, n):
            try:
                self._writer.write(ctypes.string_at(data, n))
                return 0
            except Exception as e:  # the wrapped writer failed: GzpError::Io
                self._io_error = e
                return 1

        self._cb = _native.WRITE_FN(_cb)  # keep alive
        h = ctypes.c_void_p()
        if pin is None:
            rc = self._lib.L.gzpx_par_create(ctypes.byref(cfg), self._cb, None, ctypes.byref(h))
        else:
            rc = self._lib.L.gzpx_par_create_pinned(ctypes.byref(cfg), int(pin), self._cb, None, ctypes.byref(h))
        self._lib.check(rc)
        self._h = h
        self._finished = False

    def _check(self, rc):
        if rc != _native.OK:
            msg = self._lib.L.gzpx_par_last_error(self._h).decode() or self._lib.strerror(rc)
            err = _native.GzpxError(rc, msg)
            if rc == _native.ERR_IO and self._io_error is not None:
                raise err from self._io_error
            raise err

    def write(self, buf):
        a = _native._u8(buf)
        self._check(self._lib.L.gzpx_par_write(self._h, a.ctypes.data, a.size))
        return a.size

    write_all = write

    def write_chunked(self, buf, chunk=65536):
        """The reference benchmark's write shape (benches/bench.rs:36-45), looped on the native side."""
        a = _native._u8(buf)
        self._check(self._lib.L.gzpx_par_write_chunked(self._h, a.ctypes.data, a.size, int(chunk)))
        return a.size

    def reserve(self):
        """Room inside the page-locked slab being filled, as a writable numpy view (gzpx_par_reserve);
        fill a prefix of it and commit(n)."""
        ptr = ctypes.c_void_p()
        cap = ctypes.c_size_t(0)
        self._check(self._lib.L.gzpx_par_reserve(self._h, ctypes.byref(ptr), ctypes.byref(cap)))
        buf = (ctypes.c_uint8 * cap.value).from_address(ptr.value)
        return np.frombuffer(buf, dtype=np.uint8)

    def commit(self, n):
        self._check(self._lib.L.gzpx_par_commit(self._h, int(n)))

    def index(self):
        """(compressed_offset, uncompressed_offset) of every block written so far, as an (n, 2) uint64
        array in stream order (README.md:161); complete after finish()."""
        n = ctypes.c_size_t(0)
        self._check(self._lib.L.gzpx_par_index(self._h, None, 0, ctypes.byref(n)))
        out = np.zeros((n.value, 2), dtype=np.uint64)
        self._check(self._lib.L.gzpx_par_index(self._h, out.ctypes.data, n.value, ctypes.byref(n)))
        return out[:n.value]

    def gzi(self):
        """The index in htslib's .gzi layout (bytes)."""
        idx = np.ascontiguousarray(self.index())
        cap = int(self._lib.L.gzpx_gzi_size(idx.shape[0]))
        out = np.zeros(cap, dtype=np.uint8)
        got = ctypes.c_size_t(0)
        self._lib.check(self._lib.L.gzpx_gzi_write(idx.ctypes.data, idx.shape[0], out.ctypes.data, cap,
                                                   ctypes.byref(got)))
        return out[:got.value].tobytes()

    def flush(self):
        self._check(self._lib.L.gzpx_par_flush(self._h))

    def finish(self):
        """Flush, append the trailer (BGZF EOF), join the pipeline; returns the wrapped writer."""
        if not self._finished:
            self._finished = True
            self._check(self._lib.L.gzpx_par_finish(self._h))
        return self._writer

    def close(self):
        if getattr(self, "_h", None):
            self._lib.L.gzpx_par_destroy(self._h)  # Drop: finishes if finish() was not called
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is None:
            self.finish()
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ParCompressBuilder:
    """ParCompressBuilder<F> (src/par/compress.rs:33-204)."""

    def __init__(self, fmt=Bgzf, lib=None):
        self._lib = lib
        self._fmt = fmt
        self._buffer_size = fmt.DEFAULT_BUFSIZE
        self._num_threads = os.cpu_count() or 1
        self._level = Compression(3)
        self._pin = None
        self._device = 0
        self._compat = _native.COMPAT_1_24
        self._batch_blocks = 1024

    def buffer_size(self, n):
        if n < DICT_SIZE:
            raise _native.GzpxError(_native.ERR_BUFFER_SIZE,
                                    "Invalid buffer size %d, must be >= %d" % (n, DICT_SIZE))
        self._buffer_size = n
        return self

    def num_threads(self, n):
        if n == 0:
            raise _native.GzpxError(_native.ERR_NUM_THREADS, "Invalid number of threads 0")
        self._num_threads = n
        return self

    def compression_level(self, c):
        self._level = c if isinstance(c, Compression) else Compression(c)
        return self

    def pin_threads(self, p):
        self._pin = p  # first core of the twin's threads (src/par/compress.rs:99-107)
        return self

    # GPU-side knobs (no counterpart in the reference)
    def device(self, d):
        self._device = d
        return self

    def compat(self, c):
        self._compat = c
        return self

    def batch_blocks(self, b):
        self._batch_blocks = b
        return self

    def from_writer(self, writer):
        cfg = _native.GzpxParConfig(self._fmt.FORMAT, self._level.level(), self._compat, self._device,
                                    self._buffer_size, self._num_threads, self._batch_blocks)
        return ParCompress(cfg, writer, self._lib, self._pin)

    # from_borrowed_writer (src/par/compress.rs:162-194): the caller keeps the writer; in Python
    # (and through the C ABI's callback + user pointer) every writer is borrowed, so this is the
    # same constructor -- finish() before the writer goes away, as the reference demands.
    from_borrowed_writer = from_writer


class ZBuilder:
    """ZBuilder<F, W> (src/lib.rs:181-275).  num_threads <= 1 selects SyncZ in the reference; the
    GPU path has no single-threaded variant, so both cases build a ParCompress with one lane."""

    def __init__(self, fmt=Bgzf, lib=None):
        self._b = ParCompressBuilder(fmt, lib)
        self._threads = os.cpu_count() or 1

    def buffer_size(self, n):
        self._b.buffer_size(n)
        return self

    def num_threads(self, n):
        self._threads = n
        return self

    def compression_level(self, c):
        self._b.compression_level(c)
        return self

    def pin_threads(self, p):
        self._b.pin_threads(p)
        return self

    def from_writer(self, writer):
        self._b.num_threads(max(1, self._threads))
        return self._b.from_writer(writer)


class ParDecompress:
    """`Read` over a BGZF / Mgzip stream (src/par/decompress.rs:112-352)."""

    def __init__(self, fmt, reader, device=0, batch_bytes=0, lib=None):
        self._lib = lib or _native.load()
        self._reader = reader
        self._io_error = None

        def _cb(user, buf, cap):
            try:
                data = self._reader.read(cap)
                n = len(data)
                ctypes.memmove(buf, data, n)
                return n
            except Exception as e:  # the wrapped reader failed: GzpError::Io
                self._io_error = e
                return -1

        self._cb = _native.READ_FN(_cb)
        h = ctypes.c_void_p()
        self._lib.check(self._lib.L.gzpx_pard_create(fmt.FORMAT, device, batch_bytes, self._cb, None,
                                                     ctypes.byref(h)))
        self._h = h

    def _fail(self, rc):
        msg = self._lib.L.gzpx_pard_last_error(self._h).decode() or self._lib.strerror(rc)
        err = _native.GzpxError(rc, msg)
        if self._io_error is not None:
            raise err from self._io_error
        raise err

    def readinto(self, b):
        """io.RawIOBase.readinto: up to len(b) inflated bytes straight into the caller's buffer (no allocation, one copy
        out of the slab); 0 at the end of the stream."""
        mv = memoryview(b).cast("B")
        if len(mv) == 0:
            return 0
        got = ctypes.c_size_t(0)
        dst = (ctypes.c_uint8 * len(mv)).from_buffer(mv)
        rc = self._lib.L.gzpx_pard_read(self._h, dst, len(mv), ctypes.byref(got))
        if rc != _native.OK:
            self._fail(rc)
        return got.value

    def fill_buf(self):
        """std::io::BufRead::fill_buf: a memoryview of the inflated bytes where they lie (the current slab's page-locked
        buffer) -- no copy; empty at the end of the stream.  Valid until the fill_buf / read after the consume() of all
        of it."""
        ptr, n = ctypes.c_void_p(), ctypes.c_size_t(0)
        rc = self._lib.L.gzpx_pard_fill_buf(self._h, ctypes.byref(ptr), ctypes.byref(n))
        if rc != _native.OK:
            self._fail(rc)
        if not n.value:
            return memoryview(b"")
        return memoryview((ctypes.c_uint8 * n.value).from_address(ptr.value)).cast("B")

    def consume(self, n):
        rc = self._lib.L.gzpx_pard_consume(self._h, n)
        if rc != _native.OK:
            self._fail(rc)

    def read(self, n=-1):
        """`Read::read` / read_to_end (n < 0).  One allocation for the result and one copy out of the slabs (until
        round 6: a scratch array per call, a bytes object per piece and a join -- 2 GiB/s of page faults)."""
        if n == 0:
            return b""
        if n > 0:
            out = bytearray(n)
            got, mv = 0, memoryview(out)
            while got < n:
                k = self.readinto(mv[got:])
                if k == 0:
                    break
                got += k
            del mv
            return bytes(out[:got]) if got < n else bytes(out)  # (readinto / fill_buf are the ways without this copy)
        parts = []
        while True:
            v = self.fill_buf()
            if len(v) == 0:
                break
            parts.append(bytes(v))
            self.consume(len(v))
        return b"".join(parts)

    def finish(self):
        return self._reader

    def close(self):
        if getattr(self, "_h", None):
            self._lib.L.gzpx_pard_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ParDecompressBuilder:
    """ParDecompressBuilder<F> (src/par/decompress.rs:17-109)."""

    def __init__(self, fmt=Bgzf, lib=None):
        self._fmt = fmt
        self._lib = lib
        self._device = 0
        self._batch = 0
        self._threads = os.cpu_count() or 1

    def num_threads(self, n):
        if n == 0:
            raise _native.GzpxError(_native.ERR_NUM_THREADS, "Invalid number of threads 0")
        self._threads = n
        return self

    def pin_threads(self, p):
        return self

    def device(self, d):
        self._device = d
        return self

    def batch_bytes(self, b):
        self._batch = b
        return self

    def from_reader(self, reader):
        return ParDecompress(self._fmt, reader, self._device, self._batch, self._lib)

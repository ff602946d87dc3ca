"""ctypes loader for the CPU oracle (oracle/libgzpx_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the gzp_amd package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgzpx_oracle.so")

COMPAT_1_24 = 0
COMPAT_1_10 = 1
FMT_BGZF = 0
FMT_MGZIP = 1

_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("gzpx_oracle.c", "gzpx_oracle.h", "cpu_bench.c", "synth_fastq.c", "Makefile")]
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(f) for f in srcs)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libgzpx_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        u8p = ctypes.c_void_p
        L.gzpx_oracle_crc32.restype = ctypes.c_uint32
        L.gzpx_oracle_crc32.argtypes = [ctypes.c_uint32, u8p, ctypes.c_size_t]
        L.gzpx_oracle_deflate_bound.restype = ctypes.c_size_t
        L.gzpx_oracle_deflate_bound.argtypes = [ctypes.c_size_t]
        L.gzpx_oracle_deflate_compress.restype = ctypes.c_size_t
        L.gzpx_oracle_deflate_compress.argtypes = [ctypes.c_int, ctypes.c_int, u8p, ctypes.c_size_t,
                                                   u8p, ctypes.c_size_t]
        L.gzpx_oracle_encode_block.restype = ctypes.c_size_t
        L.gzpx_oracle_encode_block.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, u8p,
                                               ctypes.c_size_t, ctypes.c_int, u8p, ctypes.c_size_t,
                                               ctypes.POINTER(ctypes.c_int)]
        L.gzpx_oracle_compress_stream.restype = ctypes.c_size_t
        L.gzpx_oracle_compress_stream.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_size_t, u8p, ctypes.c_size_t, u8p,
                                                  ctypes.c_size_t, u8p, ctypes.c_size_t,
                                                  ctypes.POINTER(ctypes.c_size_t),
                                                  ctypes.POINTER(ctypes.c_int)]
        L.gzpx_oracle_l1_tokens.restype = ctypes.c_size_t
        L.gzpx_oracle_l1_tokens.argtypes = [u8p, ctypes.c_size_t, u8p, ctypes.c_size_t, u8p,
                                            ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        L.gzpx_oracle_make_huffman_code.restype = None
        L.gzpx_oracle_make_huffman_code.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_int, u8p,
                                                    u8p, u8p]
        dp, u64p, ip = (ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64),
                        ctypes.POINTER(ctypes.c_int))
        L.gzpx_cpu_bench_compress.restype = ctypes.c_int
        L.gzpx_cpu_bench_compress.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, u8p,
                                              ctypes.c_size_t, ctypes.c_int, ctypes.c_double, dp, u64p, ip]
        L.gzpx_cpu_bench_compress_ref.restype = ctypes.c_int
        L.gzpx_cpu_bench_compress_ref.argtypes = [ctypes.c_int, ctypes.c_size_t, u8p, ctypes.c_size_t, ctypes.c_int,
                                                  ctypes.c_double, dp, u64p, ip]
        L.gzpx_cpu_bench_inflate.restype = ctypes.c_int
        L.gzpx_cpu_bench_inflate.argtypes = [u8p, u8p, u8p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int,
                                             ctypes.c_double, dp, u64p, ip]
        L.gzpx_oracle_fastq.restype = None
        L.gzpx_oracle_fastq.argtypes = [ctypes.c_uint64, ctypes.c_uint64, u8p, ctypes.c_size_t]
        L.gzpx_oracle_ascii.restype = None
        L.gzpx_oracle_ascii.argtypes = [ctypes.c_uint64, ctypes.c_uint64, u8p, ctypes.c_size_t]
        _lib = L
    return _lib


def ascii_stream(offset, n, seed=8):
    """Bytes [offset, offset + n) of synth.ascii_random's stream, natively (full-size fixtures)."""
    from gzp_amd.synth import big_zeros
    out = big_zeros(n)
    lib().gzpx_oracle_ascii(seed, offset, _ptr(out), n)
    return out


def fastq_stream(offset, n, seed=20250927):
    """Bytes [offset, offset + n) of the synthetic FASTQ stream of BASELINE configs[3]
    (oracle/synth_fastq.c): the CPU statement of gzpx_synth_fastq_device."""
    from gzp_amd.synth import big_zeros
    out = big_zeros(n)  # (pre-populated: first-touch page faults are slow in the build sandbox)
    lib().gzpx_oracle_fastq(seed, offset, _ptr(out), n)
    return out


def cpu_bench_compress(slab, fmt=FMT_BGZF, level=1, compat=COMPAT_1_24, block=65280, threads=1, wall_s=6.0):
    """Native (pthreads) ParCompress-style timing of the oracle: (bytes, seconds, threads)."""
    a = _as_u8(slab)
    el, nb, th = ctypes.c_double(0), ctypes.c_uint64(0), ctypes.c_int(0)
    rc = lib().gzpx_cpu_bench_compress(fmt, level, compat, block, _ptr(a), a.size, threads, wall_s,
                                       ctypes.byref(el), ctypes.byref(nb), ctypes.byref(th))
    if rc != 0:
        raise RuntimeError("gzpx_cpu_bench_compress failed (%d)" % rc)
    return nb.value, el.value, th.value


def cpu_bench_parcompress_ref(slab, level=1, block=65280, chunk=65536, threads=1, wall_s=6.0):
    """ParCompress<Bgzf> as gzp runs it (caller thread cutting 64 KiB write_all calls into blocks, N workers behind a
    bounded queue, an in-order writer thread into an in-memory sink -- oracle/cpu_bench.c) over the image's libdeflate
    binary.  Returns (bytes, seconds, passes, last pass's stream as a uint8 array), or None without a libdeflate."""
    a = _as_u8(slab)
    cap = a.size + a.size // 8 + (1 << 20)
    from gzp_amd.synth import big_zeros
    sink = big_zeros(cap)
    el, nb, passes, sl = ctypes.c_double(0), ctypes.c_uint64(0), ctypes.c_int(0), ctypes.c_size_t(0)
    fn = lib().gzpx_cpu_bench_parcompress_ref
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                   ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t),
                   ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int)]
    rc = fn(level, block, chunk, _ptr(a), a.size, threads, wall_s, _ptr(sink), cap, ctypes.byref(sl), ctypes.byref(el),
            ctypes.byref(nb), ctypes.byref(passes))
    if rc == -2:
        return None
    if rc != 0:
        raise RuntimeError("gzpx_cpu_bench_parcompress_ref failed (%d)" % rc)
    return nb.value, el.value, passes.value, sink[:sl.value]


def cpu_bench_compress_ref(slab, level=1, block=65280, threads=1, wall_s=6.0):
    """The same loop with the image's libdeflate binary doing the work; None if the box has none."""
    a = _as_u8(slab)
    el, nb, th = ctypes.c_double(0), ctypes.c_uint64(0), ctypes.c_int(0)
    rc = lib().gzpx_cpu_bench_compress_ref(level, block, _ptr(a), a.size, threads, wall_s, ctypes.byref(el),
                                           ctypes.byref(nb), ctypes.byref(th))
    if rc == -2:
        return None
    if rc != 0:
        raise RuntimeError("gzpx_cpu_bench_compress_ref failed (%d)" % rc)
    return nb.value, el.value, th.value


def cpu_bench_inflate(comp, offs, sizes, hdr_len=18, threads=1, wall_s=6.0):
    """Native timing of the image's libdeflate inflate + CRC32 over the blocks of `comp`; returns
    (bytes, seconds, threads) or None when the box has no libdeflate binary."""
    a = _as_u8(comp)
    o = np.ascontiguousarray(offs, dtype=np.uint64)
    s = np.ascontiguousarray(sizes, dtype=np.uint32)
    el, nb, th = ctypes.c_double(0), ctypes.c_uint64(0), ctypes.c_int(0)
    rc = lib().gzpx_cpu_bench_inflate(_ptr(a), o.ctypes.data, s.ctypes.data, o.size, hdr_len, threads, wall_s,
                                      ctypes.byref(el), ctypes.byref(nb), ctypes.byref(th))
    if rc == -2:
        return None
    if rc != 0:
        raise RuntimeError("gzpx_cpu_bench_inflate failed (%d)" % rc)
    return nb.value, el.value, th.value


def _as_u8(data):
    a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    return np.ascontiguousarray(a, dtype=np.uint8)


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def crc32(data, crc=0):
    a = _as_u8(data)
    return int(lib().gzpx_oracle_crc32(crc, _ptr(a), a.size))


def deflate_compress(data, level=1, compat=COMPAT_1_24, cap=None):
    a = _as_u8(data)
    if cap is None:
        cap = int(lib().gzpx_oracle_deflate_bound(a.size)) + 64
    out = np.empty(cap, dtype=np.uint8)
    n = lib().gzpx_oracle_deflate_compress(level, compat, _ptr(a), a.size, _ptr(out), cap)
    if n == 0:
        raise RuntimeError("oracle deflate_compress: does not fit / unsupported level")
    return out[:n].tobytes()


def encode_block(data, fmt=FMT_BGZF, level=1, compat=COMPAT_1_24, is_last=False):
    a = _as_u8(data)
    cap = 20 + a.size + max(128, a.size // 10) + 8 + 28 + 16
    out = np.empty(cap, dtype=np.uint8)
    err = ctypes.c_int(0)
    n = lib().gzpx_oracle_encode_block(fmt, level, compat, _ptr(a), a.size, int(is_last),
                                       _ptr(out), cap, ctypes.byref(err))
    if n == 0:
        raise RuntimeError("oracle encode_block error %d" % err.value)
    return out[:n].tobytes()


def compress_stream(data, fmt=FMT_BGZF, level=1, compat=COMPAT_1_24, buffer_size=65280,
                    return_block_sizes=False):
    a = _as_u8(data)
    nb_max = a.size // buffer_size + 2
    cap = a.size + nb_max * (20 + 8 + 8 + max(128, buffer_size // 10)) + 64
    out = np.empty(cap, dtype=np.uint8)
    sizes = np.zeros(nb_max, dtype=np.uint32)
    nb = ctypes.c_size_t(0)
    err = ctypes.c_int(0)
    n = lib().gzpx_oracle_compress_stream(fmt, level, compat, buffer_size, _ptr(a), a.size,
                                          _ptr(out), cap, _ptr(sizes), nb_max, ctypes.byref(nb),
                                          ctypes.byref(err))
    if n == 0:
        raise RuntimeError("oracle compress_stream error %d" % err.value)
    if return_block_sizes:
        return out[:n].tobytes(), sizes[:nb.value].copy()
    return out[:n].tobytes()


def l1_tokens(data):
    """(tokens uint32[], sub_block_first_token uint32[]) of the level-1 parse."""
    a = _as_u8(data)
    toks = np.empty(max(a.size, 1), dtype=np.uint32)
    first = np.empty(a.size // 8192 + 4, dtype=np.uint32)
    nsub = ctypes.c_size_t(0)
    n = lib().gzpx_oracle_l1_tokens(_ptr(a), a.size, _ptr(toks), toks.size, _ptr(first), first.size,
                                    ctypes.byref(nsub))
    return toks[:n].copy(), first[:nsub.value].copy()


def make_huffman_code(freqs, max_len, compat=COMPAT_1_24):
    f = np.ascontiguousarray(freqs, dtype=np.uint32)
    lens = np.zeros(f.size, dtype=np.uint8)
    cws = np.zeros(f.size, dtype=np.uint32)
    lib().gzpx_oracle_make_huffman_code(f.size, max_len, compat, _ptr(f), _ptr(lens), _ptr(cws))
    return lens, cws

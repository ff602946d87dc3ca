"""Crc32::combine, Adler32::update / combine (src/check.rs:85-164) behind the C ABI, against Python's zlib module: the
arithmetic helpers straight, the device kernel through the emulator.  No GPU."""
import zlib

import numpy as np

from gzp_amd import _native, synth


def _pieces():
    rng = np.random.default_rng(7)
    out = [b"", b"a", b"hello hello hello", bytes(65521), bytes([255]) * 70000]
    for n in (1, 255, 256, 257, 65535, 65536, 65537, 200001):
        out.append(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
    out.append(synth.make("text", 300000, 3).tobytes())
    return out


def test_combines_against_zlib(emu_lib):
    ps = _pieces()
    for a in ps:
        for b in ps[:9]:
            assert _native.crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b), lib=emu_lib) == zlib.crc32(a + b)
            assert _native.adler32_combine(zlib.adler32(a), zlib.adler32(b), len(b), lib=emu_lib) == zlib.adler32(a + b)
    # lengths far beyond a buffer: the algebra against itself (A || B || C both ways) and against zlib where it can run
    big = 5_000_000_123
    c1, c2, c3 = 0x12345678, 0x9ABCDEF0, zlib.crc32(b"tail")
    left = _native.crc32_combine(_native.crc32_combine(c1, c2, big, lib=emu_lib), c3, 4, lib=emu_lib)
    right = _native.crc32_combine(c1, _native.crc32_combine(c2, c3, 4, lib=emu_lib), big + 4, lib=emu_lib)
    assert left == right


def test_adler32_kernel_through_the_emulator(emu_lib):
    for p in _pieces():
        assert _native.adler32(p, lib=emu_lib) == zlib.adler32(p)
    a, b = _pieces()[-1], _pieces()[-2]
    mid = _native.adler32(a, lib=emu_lib)
    assert _native.adler32(b, mid, lib=emu_lib) == zlib.adler32(a + b)  # Adler32::update continues a running check

"""Adler32::update on the MI355X (gzpx_adler32: k_adler32) and the combines, against Python's zlib module."""
import zlib

import numpy as np
import pytest

from gzp_amd import _native, synth

pytestmark = pytest.mark.gpu


def test_adler32_and_combines(hip_lib):
    rng = np.random.default_rng(11)
    bufs = [b"", b"x", bytes([255]) * 1_000_003, rng.integers(0, 256, 150_000_001, dtype=np.uint8).tobytes(),
            synth.text_slab(40 << 20, seed=3).tobytes()]
    sums = []
    for b in bufs:
        got = _native.adler32(b, lib=hip_lib)
        assert got == zlib.adler32(b), len(b)
        sums.append(got)
    run_a, run_c, total = 1, 0, b""
    for b, s in zip(bufs, sums):
        run_a = _native.adler32(b, run_a, lib=hip_lib)  # update ...
        total += b
        assert run_a == zlib.adler32(total)
    acc_a, acc_c, n = zlib.adler32(b""), zlib.crc32(b""), 0
    for b in bufs:  # ... and combine, the way ParCompress<Gzip / Zlib>'s writer folds its blocks' checks
        acc_a = _native.adler32_combine(acc_a, zlib.adler32(b), len(b), lib=hip_lib)
        acc_c = _native.crc32_combine(acc_c, _native.crc32(b, lib=hip_lib) if b else 0, len(b), lib=hip_lib)
    assert acc_a == zlib.adler32(total) and acc_c == zlib.crc32(total)

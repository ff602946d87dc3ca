"""The CPU oracle against the committed known-answer vectors (libdeflate 1.10 binary output,
tests/golden/make_golden.py) and against independent inflaters.  No GPU."""
import gzip
import hashlib
import zlib

import numpy as np
import pytest

from gzp_amd import synth


def _input(e):
    if "input_hex" in e:
        return np.frombuffer(bytes.fromhex(e["input_hex"]), dtype=np.uint8)
    return synth.make(e["class"], e["n"], e["seed"])


def test_inputs_are_reproducible(golden):
    for e in golden["raw_deflate"][::7]:
        a = synth.make(e["class"], e["n"], e["seed"])
        assert hashlib.sha256(a.tobytes()).hexdigest() == e["input_sha256"]


def test_raw_deflate_matches_libdeflate_vectors(oracle, golden):
    for e in golden["raw_deflate"] + golden["raw_deflate_literal_inputs"]:
        a = _input(e)
        out = oracle.deflate_compress(a, e["level"], oracle.COMPAT_1_10)
        assert len(out) == e["size"], e
        assert hashlib.sha256(out).hexdigest() == e["sha256"], e
        if "hex" in e:
            assert out.hex() == e["hex"]
        assert zlib.decompress(out, -15) == a.tobytes()


def test_streams_match_vectors(oracle, golden):
    for e in golden["streams"]:
        a = _input(e)
        fmt = oracle.FMT_BGZF if e["fmt"] == "bgzf" else oracle.FMT_MGZIP
        out, sizes = oracle.compress_stream(a, fmt, e["level"], oracle.COMPAT_1_10, e["buffer_size"], True)
        assert hashlib.sha256(out).hexdigest() == e["sha256"], e
        assert list(sizes) == e["block_sizes"]
        assert gzip.decompress(out) == a.tobytes()


def test_bgzf_eof_and_empty_stream(oracle, golden):
    eof = bytes.fromhex(golden["bgzf_eof_hex"])
    out = oracle.compress_stream(np.zeros(0, np.uint8), oracle.FMT_BGZF, 1, oracle.COMPAT_1_10, 65280)
    # an empty stream is one empty stored block + EOF
    assert out == bytes.fromhex("1f8b08040000000004ff0600424302001e00010000ffff0000000000000000") + eof
    assert out.endswith(eof) and len(eof) == 28


def test_compat_delta_is_only_the_empty_offset_code(oracle):
    # all-literal dynamic block: the two libdeflate generations differ in HDIST / offset lens only
    a = synth.low_entropy_binary(3000, 9)
    o10 = oracle.deflate_compress(a, 1, oracle.COMPAT_1_10)
    o24 = oracle.deflate_compress(a, 1, oracle.COMPAT_1_24)
    assert zlib.decompress(o10, -15) == a.tobytes()
    assert zlib.decompress(o24, -15) == a.tobytes()
    # text with matches: identical
    t = synth.english_like(65280, 3)
    assert oracle.deflate_compress(t, 1, oracle.COMPAT_1_10) == oracle.deflate_compress(t, 1, oracle.COMPAT_1_24)


def test_crc32_matches_zlib(oracle):
    for n in [0, 1, 255, 256, 257, 65280, 100001]:
        a = synth.uniform_random(n, n + 1)
        assert oracle.crc32(a) == zlib.crc32(a.tobytes())


def test_huffman_code_is_prefix_free_and_length_limited(oracle):
    rng = np.random.default_rng(5)
    for _ in range(50):
        f = (rng.geometric(0.02, 288) * (rng.random(288) < 0.7)).astype(np.uint32)
        if f.sum() == 0:
            continue
        lens, cws = oracle.make_huffman_code(f, 14)
        assert lens.max() <= 14
        used = lens[lens > 0].astype(np.int64)
        if (f > 0).sum() >= 2:
            assert np.sum(2.0 ** -used) == pytest.approx(1.0)  # complete code


def test_levels_2_to_4_match_libdeflate_vectors(oracle, golden_hc):
    for e in golden_hc["raw_deflate"]:
        a = synth.make(e["class"], e["n"], e["seed"])
        out = oracle.deflate_compress(a, e["level"], oracle.COMPAT_1_10)
        assert hashlib.sha256(out).hexdigest() == e["sha256"], e
        assert zlib.decompress(out, -15) == a.tobytes()
    for e in golden_hc["streams"]:
        a = synth.make(e["class"], e["n"], e["seed"])
        fmt = oracle.FMT_BGZF if e["fmt"] == "bgzf" else oracle.FMT_MGZIP
        out, sizes = oracle.compress_stream(a, fmt, e["level"], oracle.COMPAT_1_10, e["buffer_size"], True)
        assert hashlib.sha256(out).hexdigest() == e["sha256"], e
        assert list(sizes) == e["block_sizes"]


def test_levels_5_to_9_match_libdeflate_vectors(oracle, golden_lazy):
    """The lazy / lazy2 parsers (deflate_compress_lazy_generic) against the v1.10 binary's output."""
    for e in golden_lazy["raw_deflate"]:
        a = synth.make(e["class"], e["n"], e["seed"])
        out = oracle.deflate_compress(a, e["level"], oracle.COMPAT_1_10)
        assert hashlib.sha256(out).hexdigest() == e["sha256"], e
        if e["n"] <= 70000:
            assert zlib.decompress(out, -15) == a.tobytes()
    for e in golden_lazy["streams"]:
        a = synth.make(e["class"], e["n"], e["seed"])
        fmt = oracle.FMT_BGZF if e["fmt"] == "bgzf" else oracle.FMT_MGZIP
        out, sizes = oracle.compress_stream(a, fmt, e["level"], oracle.COMPAT_1_10, e["buffer_size"], True)
        assert hashlib.sha256(out).hexdigest() == e["sha256"], e
        assert list(sizes) == e["block_sizes"]


def test_levels_10_to_12_match_libdeflate_vectors(oracle, golden_near_optimal):
    """The near-optimal parser (deflate_compress_near_optimal: bt_matchfinder, match cache, block splitting with
    rewind, iterated minimum-cost path) against the v1.10 binary's output."""
    for e in golden_near_optimal["raw_deflate"]:
        a = synth.make(e["class"], e["n"], e["seed"])
        out = oracle.deflate_compress(a, e["level"], oracle.COMPAT_1_10)
        assert hashlib.sha256(out).hexdigest() == e["sha256"], e
        if e["n"] <= 70000:
            assert zlib.decompress(out, -15) == a.tobytes()
    for e in golden_near_optimal["streams"]:
        a = synth.make(e["class"], e["n"], e["seed"])
        fmt = oracle.FMT_BGZF if e["fmt"] == "bgzf" else oracle.FMT_MGZIP
        out, sizes = oracle.compress_stream(a, fmt, e["level"], oracle.COMPAT_1_10, e["buffer_size"], True)
        assert hashlib.sha256(out).hexdigest() == e["sha256"], e
        assert list(sizes) == e["block_sizes"]


def test_default_litlen_cost_tables_are_the_binarys(oracle, golden_near_optimal):
    """libdeflate's default_litlen_costs[] (three rows of 257 literal costs + a length-symbol cost), as they sit in
    the v1.10 binary's read-only data, are int(-log2((1 - p) / max(j, 1)) * 16) and int(-log2(p / 29) * 16)."""
    import ctypes
    tables = golden_near_optimal["default_litlen_costs"]
    assert tables is not None and len(tables) == 3
    lit = ((ctypes.c_uint8 * 257) * 3)()
    ln = (ctypes.c_uint8 * 3)()
    oracle.lib().gzpx_oracle_default_litlen_costs(lit, ln)
    for k in range(3):
        assert list(lit[k]) == tables[k][:257]
        assert ln[k] == tables[k][257]

"""Kernel LOGIC on the CPU: the product sources (gzp_amd/csrc) compiled against the SIMT
emulator in tests/emu and compared with the oracle and the golden vectors.  This is a check of
indexing/arithmetic, not a product path (see tests/emu/hip/hip_runtime.h).  No GPU."""
import gzip
import hashlib

import numpy as np
import pytest

from gzp_amd import _native, synth


@pytest.fixture(scope="module")
def ctx10(emu_lib):
    c = _native.Context(level=1, compat=_native.COMPAT_1_10, lib=emu_lib, max_slab_bytes=6 * 65280)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx24(emu_lib):
    c = _native.Context(level=1, compat=_native.COMPAT_1_24, lib=emu_lib, max_slab_bytes=6 * 65280)
    yield c
    c.close()


def test_golden_streams_bgzf(ctx10, golden):
    for e in golden["streams"]:
        if e["fmt"] != "bgzf" or e["buffer_size"] != 65280:
            continue
        a = (np.frombuffer(bytes.fromhex(e["input_hex"]), dtype=np.uint8) if "input_hex" in e
             else synth.make(e["class"], e["n"], e["seed"]))
        out, sizes = ctx10.compress_slab(a, True, return_block_sizes=True)
        assert hashlib.sha256(out).hexdigest() == e["sha256"], e
        assert list(sizes) == e["block_sizes"]


@pytest.mark.parametrize("cls", sorted(synth.CLASSES))
def test_classes_vs_oracle(ctx10, ctx24, oracle, cls):
    for n in [0, 1, 51, 52, 53, 100, 1000, 5000, 32767, 32768, 32769, 40000, 65279, 65280, 65281]:
        a = synth.make(cls, n, 100 + n)
        for ctx, compat in ((ctx10, oracle.COMPAT_1_10), (ctx24, oracle.COMPAT_1_24)):
            got = ctx.compress_slab(a, True)
            want = oracle.compress_stream(a, oracle.FMT_BGZF, 1, compat, 65280)
            assert got == want, (cls, n, compat)


@pytest.mark.parametrize("flags", [2, 4])
def test_level1_dense_and_handed_back_paths(emu_lib, oracle, flags):
    """Level 1 has three routes to the same tokens: k_mparse (match on demand, the default), the dense
    k_match / k_parse pair over every block (debug bit 1: what blocks above 64 KiB take), and k_mparse
    handing blocks back to the dense pair through the redo list (debug bit 2 forces it for every block)."""
    with _native.Context(level=1, compat=_native.COMPAT_1_10, lib=emu_lib, max_slab_bytes=3 * 65280) as c:
        c.debug_set_flags(flags)
        for cls in sorted(synth.CLASSES):
            for n in (100, 5000, 65280, 2 * 65280 + 4321):
                a = synth.make(cls, n, 7 + n)
                assert c.compress_slab(a, True) == oracle.compress_stream(a, oracle.FMT_BGZF, 1, oracle.COMPAT_1_10,
                                                                          65280), (cls, n, flags)
                if flags == 4:
                    assert c.debug_redo_count() == (n + 65279) // 65280


def test_level1_on_demand_keeps_text_and_hands_back_long_runs(emu_lib, oracle):
    """k_mparse settles on text-like data (no block goes to the dense kernels); long runs shift the
    phase of the walk segments by one per round and are handed back -- with the same stream either way."""
    with _native.Context(level=1, compat=_native.COMPAT_1_10, lib=emu_lib, max_slab_bytes=3 * 65280) as c:
        for cls, expect_redo in (("text", False), ("fastq", False), ("dna", False), ("random", False),
                                 ("zeros", True), ("period2", True)):
            a = synth.make(cls, 3 * 65280, 5)
            assert c.compress_slab(a, True) == oracle.compress_stream(a, oracle.FMT_BGZF, 1, oracle.COMPAT_1_10, 65280)
            assert (c.debug_redo_count() > 0) == expect_redo, (cls, c.debug_redo_count())


def test_tokens_match_oracle(ctx10, oracle):
    a = synth.repeated_phrases(65280, 3)
    ctx10.compress_slab(a, True)
    toks, first = ctx10.debug_tokens(0)
    et, ef = oracle.l1_tokens(a)
    assert np.array_equal(toks, et)
    assert list(first) == list(ef)


def test_multi_block_and_not_last(ctx10, oracle):
    a = synth.make("fastq", 3 * 65280, 8)
    got = ctx10.compress_slab(a, is_last=False)
    want = b"".join(oracle.encode_block(a[i * 65280:(i + 1) * 65280], oracle.FMT_BGZF, 1,
                                        oracle.COMPAT_1_10, False) for i in range(3))
    assert got == want
    with pytest.raises(_native.GzpxError):
        ctx10.compress_slab(a[:1000], is_last=False)  # not a multiple of buffer_size


def test_other_buffer_sizes_and_mgzip(emu_lib, oracle):
    a = synth.make("mixed", 150001, 4)
    for fmt, ofmt in ((_native.FORMAT_BGZF, oracle.FMT_BGZF), (_native.FORMAT_MGZIP, oracle.FMT_MGZIP)):
        for bs in (32768, 40001, 65536):
            if fmt == _native.FORMAT_BGZF and bs == 65536:
                continue
            with _native.Context(format=fmt, level=1, buffer_size=bs, compat=_native.COMPAT_1_10,
                                 lib=emu_lib, max_slab_bytes=a.size) as c:
                got = c.compress_slab(a, True)
            want = oracle.compress_stream(a, ofmt, 1, oracle.COMPAT_1_10, bs)
            assert got == want, (fmt, bs)
            assert gzip.decompress(got) == a.tobytes()


def test_block_size_exceeded_is_reported(emu_lib):
    # BGZF with buffer_size 65536 and incompressible data: payload 65541 >= 65536
    a = synth.uniform_random(65536, 1)
    with _native.Context(format=_native.FORMAT_BGZF, level=1, buffer_size=65536, lib=emu_lib,
                         max_slab_bytes=a.size) as c:
        with pytest.raises(_native.GzpxError) as ei:
            c.compress_slab(a, True)
    assert ei.value.code == _native.ERR_BLOCK_SIZE_EXCEEDED and ei.value.block == 0


def test_libdeflate_shaped_abi(emu_lib, oracle):
    comp = _native.Compressor(1, _native.COMPAT_1_10, lib=emu_lib)
    for cls, n in [("text", 65536), ("text", 10), ("random", 4000), ("zeros", 0)]:
        a = synth.make(cls, n, 21)
        assert comp.deflate_compress(a) == oracle.deflate_compress(a, 1, oracle.COMPAT_1_10)
    with pytest.raises(_native.GzpxError):
        comp.deflate_compress(synth.uniform_random(5000, 2), cap=100)  # does not fit -> 0
    comp.close()
    a = synth.uniform_random(200001, 3)
    import zlib
    assert _native.crc32(a, lib=emu_lib) == zlib.crc32(a.tobytes())
    assert _native.crc32(a[1000:], crc=zlib.crc32(a[:1000].tobytes()), lib=emu_lib) == zlib.crc32(a.tobytes())


def test_encode_block_flush_semantics(ctx10, oracle):
    a = synth.english_like(1234, 5)
    assert ctx10.encode_block(a, is_last=False) == oracle.encode_block(a, oracle.FMT_BGZF, 1, oracle.COMPAT_1_10, False)
    assert ctx10.encode_block(a, is_last=True) == oracle.encode_block(a, oracle.FMT_BGZF, 1, oracle.COMPAT_1_10, True)
    e = np.zeros(0, np.uint8)
    assert ctx10.encode_block(e, is_last=False) == oracle.encode_block(e, oracle.FMT_BGZF, 1, oracle.COMPAT_1_10, False)


def test_builder_validation(emu_lib):
    with pytest.raises(_native.GzpxError) as ei:
        _native.Context(buffer_size=1000, lib=emu_lib)
    assert ei.value.code == _native.ERR_BUFFER_SIZE
    with pytest.raises(_native.GzpxError) as ei:
        _native.Context(level=13, lib=emu_lib)
    assert ei.value.code == _native.ERR_COMPRESSION_LEVEL
    with pytest.raises(_native.GzpxError) as ei:  # valid in gzp, not built (blocks above 64 MiB): never a CPU fallback
        _native.Context(format=_native.FORMAT_MGZIP, buffer_size=(64 << 20) + 1, lib=emu_lib)
    assert ei.value.code == _native.ERR_UNSUPPORTED
    _native.Context(level=12, lib=emu_lib, max_slab_bytes=65280).close()  # CompressionLvl accepts 0..12 (src/deflate.rs:596-599)


def test_order_independent_candidate_kernel(emu_lib, oracle):
    """k_candidates' order-independent fallback (for a failed LDS-order check) forced on every block."""
    with _native.Context(level=1, compat=_native.COMPAT_1_10, lib=emu_lib, max_slab_bytes=3 * 65280) as c:
        c.debug_set_flags(1)
        for cls in ("text", "zeros", "period2", "repeats", "random"):
            a = synth.make(cls, 2 * 65280 + 99, 17)
            assert c.compress_slab(a, True) == oracle.compress_stream(a, oracle.FMT_BGZF, 1, oracle.COMPAT_1_10, 65280), cls


def test_golden_streams_large_blocks(emu_lib, golden):
    """Mgzip with buffer sizes above one 64 KiB tile (incl. the 1 MiB blocks of BASELINE config 3)."""
    seen = 0
    for e in golden["streams"]:
        if e["buffer_size"] <= 65536:
            continue
        a = synth.make(e["class"], e["n"], e["seed"])
        fmt = _native.FORMAT_BGZF if e["fmt"] == "bgzf" else _native.FORMAT_MGZIP
        with _native.Context(format=fmt, level=1, buffer_size=e["buffer_size"], compat=_native.COMPAT_1_10,
                             lib=emu_lib, max_slab_bytes=a.size) as c:
            out, sizes = c.compress_slab(a, True, return_block_sizes=True)
        assert hashlib.sha256(out).hexdigest() == e["sha256"], e
        assert list(sizes) == e["block_sizes"]
        seen += 1
    assert seen >= 2


def test_golden_raw_deflate_large_inputs(emu_lib, golden):
    comp = _native.Compressor(1, _native.COMPAT_1_10, lib=emu_lib)
    seen = 0
    for e in golden["raw_deflate"]:
        if e["n"] <= 65536:
            continue
        a = synth.make(e["class"], e["n"], e["seed"])
        assert hashlib.sha256(comp.deflate_compress(a)).hexdigest() == e["sha256"], e
        seen += 1
    comp.close()
    assert seen >= 8


@pytest.mark.parametrize("cls", ["text", "repeats", "zeros", "random", "fastq", "runs"])
def test_large_mgzip_blocks_vs_oracle(emu_lib, oracle, cls):
    for bs, n in [(131072, 131072), (131072, 300001), (100000, 250000), (1 << 20, (1 << 20) + 77)]:
        a = synth.make(cls, n, 40 + n % 97)
        with _native.Context(format=_native.FORMAT_MGZIP, level=1, buffer_size=bs, compat=_native.COMPAT_1_10,
                             lib=emu_lib, max_slab_bytes=a.size) as c:
            got = c.compress_slab(a, True)
        assert got == oracle.compress_stream(a, oracle.FMT_MGZIP, 1, oracle.COMPAT_1_10, bs), (cls, bs, n)


def test_regression_soft_limit_boundary_on_a_tile_edge(emu_lib, oracle):
    # found by tools/gpu_fuzz.py: a sub-block that starts on the first position of a 64 KiB parse
    # tile ends (65535-byte soft limit) on the tile's last position -- a second boundary in one tile
    # that is not caused by the 8192-match rule
    a = synth.make("mixed", 256005, 514286759)
    with _native.Context(format=_native.FORMAT_MGZIP, level=1, buffer_size=285614, lib=emu_lib,
                         max_slab_bytes=a.size) as c:
        got = c.compress_slab(a, True)
    assert got == oracle.compress_stream(a, oracle.FMT_MGZIP, 1, oracle.COMPAT_1_24, 285614)

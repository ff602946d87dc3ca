"""Levels 2-4 (greedy hc_matchfinder; gzp's default level is 3) and 5-9 (lazy / lazy2) on the real HIP
library."""
import gzip
import hashlib
import io

import numpy as np
import pytest

from gzp_amd import _native, par, synth

pytestmark = pytest.mark.gpu


def hetero(n, seed):
    rng = np.random.default_rng(seed)
    names = ["dna", "random", "text", "zeros", "lowent", "fastq", "ascii", "runs"]
    parts, size = [], 0
    while size < n:
        ln = int(rng.integers(3000, 60000))
        parts.append(synth.make(names[rng.integers(len(names))], ln, int(rng.integers(1 << 30))))
        size += ln
    return np.ascontiguousarray(np.concatenate(parts)[:n])


def test_golden_vectors_levels(hip_lib, golden_hc):
    comps = {L: _native.Compressor(L, _native.COMPAT_1_10, lib=hip_lib) for L in (0, 2, 3, 4)}
    for e in golden_hc["raw_deflate"]:
        a = synth.make(e["class"], e["n"], e["seed"])
        assert hashlib.sha256(comps[e["level"]].deflate_compress(a)).hexdigest() == e["sha256"], e
    for c in comps.values():
        c.close()
    for e in golden_hc["streams"]:
        a = synth.make(e["class"], e["n"], e["seed"])
        fmt = _native.FORMAT_BGZF if e["fmt"] == "bgzf" else _native.FORMAT_MGZIP
        with _native.Context(format=fmt, level=e["level"], buffer_size=e["buffer_size"],
                             compat=_native.COMPAT_1_10, lib=hip_lib, max_slab_bytes=max(a.size, 1)) as c:
            out, sizes = c.compress_slab(a, True, return_block_sizes=True)
        assert hashlib.sha256(out).hexdigest() == e["sha256"], e
        assert list(sizes) == e["block_sizes"]


def test_greedy_levels_by_every_route(hip_lib, oracle):
    from test_emu_levels import _route_cases, _three_routes
    _three_routes(hip_lib, oracle, _route_cases(scale=4))


@pytest.mark.parametrize("level", [2, 3, 4])
def test_heterogeneous_blocks_vs_oracle(hip_lib, oracle, level):
    for fmt, ofmt, bs, n in [(_native.FORMAT_BGZF, 0, 65280, 40 * 65280 + 99),
                             (_native.FORMAT_MGZIP, 1, 1 << 20, 5 * (1 << 20) + 4321)]:
        a = hetero(n, 10 * level + bs % 7)
        with _native.Context(format=fmt, level=level, buffer_size=bs, compat=_native.COMPAT_1_24, lib=hip_lib,
                             max_slab_bytes=n) as c:
            got = c.compress_slab(a, True)
        assert got == oracle.compress_stream(a, ofmt, level, oracle.COMPAT_1_24, bs), (level, fmt, bs)
        assert gzip.decompress(got) == a.tobytes()


def test_config3_shape_mgzip_1mib_level3(hip_lib, oracle):
    """BASELINE config 3 in small: Mgzip, 1 MiB blocks, level 3, ASCII noise (0x20 + u8 % 95)."""
    a = synth.ascii_random(16 * (1 << 20) + 5, 99)
    with _native.Context(format=_native.FORMAT_MGZIP, level=3, buffer_size=1 << 20, lib=hip_lib,
                         max_slab_bytes=a.size) as c:
        got = c.compress_slab(a, True)
    assert got == oracle.compress_stream(a, oracle.FMT_MGZIP, 3, oracle.COMPAT_1_24, 1 << 20)


def test_builder_default_level(hip_lib, oracle):
    a = synth.text_slab(20 * 65280 + 5, 2_000_000, 3)
    sink = io.BytesIO()
    w = par.ParCompressBuilder(par.Bgzf, lib=hip_lib).batch_blocks(4).from_writer(sink)  # level 3
    w.write_all(a)
    w.finish()
    w.close()
    assert sink.getvalue() == oracle.compress_stream(a, oracle.FMT_BGZF, 3, oracle.COMPAT_1_24, 65280)


# ---- levels 5-9

def test_golden_vectors_lazy_levels(hip_lib, golden_lazy):
    comps = {L: _native.Compressor(L, _native.COMPAT_1_10, lib=hip_lib) for L in (5, 6, 7, 8, 9)}
    for e in golden_lazy["raw_deflate"]:
        a = synth.make(e["class"], e["n"], e["seed"])
        assert hashlib.sha256(comps[e["level"]].deflate_compress(a)).hexdigest() == e["sha256"], e
    for c in comps.values():
        c.close()
    for e in golden_lazy["streams"]:
        a = synth.make(e["class"], e["n"], e["seed"])
        fmt = _native.FORMAT_BGZF if e["fmt"] == "bgzf" else _native.FORMAT_MGZIP
        with _native.Context(format=fmt, level=e["level"], buffer_size=e["buffer_size"],
                             compat=_native.COMPAT_1_10, lib=hip_lib, max_slab_bytes=max(a.size, 1)) as c:
            out, sizes = c.compress_slab(a, True, return_block_sizes=True)
        assert hashlib.sha256(out).hexdigest() == e["sha256"], e
        assert list(sizes) == e["block_sizes"]


@pytest.mark.parametrize("level", [5, 6, 7, 8, 9])
def test_heterogeneous_blocks_vs_oracle_lazy(hip_lib, oracle, level):
    for fmt, ofmt, bs, n in [(_native.FORMAT_BGZF, 0, 65280, 24 * 65280 + 99),
                             (_native.FORMAT_MGZIP, 1, 1 << 20, 2 * (1 << 20) + 4321)]:
        a = hetero(n, 10 * level + bs % 7)
        with _native.Context(format=fmt, level=level, buffer_size=bs, compat=_native.COMPAT_1_24, lib=hip_lib,
                             max_slab_bytes=n) as c:
            got = c.compress_slab(a, True)
        assert got == oracle.compress_stream(a, ofmt, level, oracle.COMPAT_1_24, bs), (level, fmt, bs)
        assert gzip.decompress(got) == a.tobytes()


def test_builder_best_level(hip_lib, oracle):
    """Compression::best() = 9 through the twin: XFL 2 in every header (src/bgzf.rs:278-284)."""
    a = synth.text_slab(6 * 65280 + 5, 2_000_000, 4)
    sink = io.BytesIO()
    w = par.ParCompressBuilder(par.Bgzf, lib=hip_lib).compression_level(par.Compression.best()).from_writer(sink)
    w.write_all(a)
    w.finish()
    w.close()
    got = sink.getvalue()
    assert got[8] == 2
    assert got == oracle.compress_stream(a, oracle.FMT_BGZF, 9, oracle.COMPAT_1_24, 65280)


@pytest.mark.parametrize("level", [1, 3, 6])
def test_largest_block_16mib(hip_lib, oracle, level):
    """buffer_size = 16 MiB: one full block and a ragged second one,
    heterogeneous content so that sub-blocks split and min_len changes along the block."""
    bs = 1 << 24
    a = hetero(bs + 1_234_567, 77 + level)
    with _native.Context(format=_native.FORMAT_MGZIP, level=level, buffer_size=bs, compat=_native.COMPAT_1_24,
                         lib=hip_lib, max_slab_bytes=a.size) as c:
        got, sizes = c.compress_slab(a, True, return_block_sizes=True)
    want, wsizes = oracle.compress_stream(a, oracle.FMT_MGZIP, level, oracle.COMPAT_1_24, bs, True)
    assert list(sizes) == list(wsizes)
    assert got == want


# ---- levels 10-12

def test_golden_vectors_near_optimal_levels(hip_lib, golden_near_optimal):
    comps = {L: _native.Compressor(L, _native.COMPAT_1_10, lib=hip_lib) for L in (10, 11, 12)}
    for e in golden_near_optimal["raw_deflate"]:
        # (one whole-buffer call = one block = ONE lane of k_near_optimal: the large vectors take ten seconds and more
        # each, so the GPU run keeps three of them -- the soft block limit, a window slide, a rewind -- and the oracle
        # test covers all)
        if e["n"] > 140000 and (e["level"], e["class"], e["n"]) not in ((11, "dna", 305001), (12, "text", 304999), (12, "lowent", 200000)):
            continue
        a = synth.make(e["class"], e["n"], e["seed"])
        assert hashlib.sha256(comps[e["level"]].deflate_compress(a)).hexdigest() == e["sha256"], e
    for c in comps.values():
        c.close()
    for e in golden_near_optimal["streams"]:
        a = synth.make(e["class"], e["n"], e["seed"])
        fmt = _native.FORMAT_BGZF if e["fmt"] == "bgzf" else _native.FORMAT_MGZIP
        with _native.Context(format=fmt, level=e["level"], buffer_size=e["buffer_size"],
                             compat=_native.COMPAT_1_10, lib=hip_lib, max_slab_bytes=max(a.size, 1)) as c:
            out, sizes = c.compress_slab(a, True, return_block_sizes=True)
        assert hashlib.sha256(out).hexdigest() == e["sha256"], e
        assert list(sizes) == e["block_sizes"]


@pytest.mark.parametrize("level", [10, 11, 12])
def test_heterogeneous_blocks_vs_oracle_near_optimal(hip_lib, oracle, level):
    for fmt, ofmt, bs, n in [(_native.FORMAT_BGZF, 0, 65280, 24 * 65280 + 99),
                             (_native.FORMAT_MGZIP, 1, 330001, 2 * 330001 + 4321)]:
        a = hetero(n, 10 * level + bs % 7)
        # (asked for the 1.24 rules on purpose: at these levels the context runs 1.10's throughout -- no hybrid stream)
        with _native.Context(format=fmt, level=level, buffer_size=bs, compat=_native.COMPAT_1_24, lib=hip_lib,
                             max_slab_bytes=n) as c:
            assert c.active_compat() == _native.COMPAT_1_10
            got = c.compress_slab(a, True)
        assert got == oracle.compress_stream(a, ofmt, level, oracle.COMPAT_1_10, bs), (level, fmt, bs)
        assert gzip.decompress(got) == a.tobytes()


@pytest.mark.parametrize("level", [1, 3])
def test_mgzip_blocks_of_32_mib(hip_lib, oracle, level):
    """src/mgzip.rs:187-218 puts no limit on a block's size (until round 6 this library stopped at 16 MiB): 32 MiB
    Mgzip blocks, level 1 and gzp's default level, against the oracle's stream; inflated again by both routes."""
    bs = 32 << 20
    a = np.concatenate([synth.make("text", 20 << 20, 3), synth.make("mixed", 30 << 20, 4), synth.make("fastq", (20 << 20) + 12345, 5)])
    want = oracle.compress_stream(a, oracle.FMT_MGZIP, level, oracle.COMPAT_1_24, bs)
    with _native.Context(format=_native.FORMAT_MGZIP, level=level, buffer_size=bs, lib=hip_lib) as c:
        got = c.compress_slab(a, True)
    assert bytes(got) == bytes(want)
    for route in (_native.INFLATE_SEG, _native.INFLATE_WAVE):
        with _native.DContext(format=_native.FORMAT_MGZIP, lib=hip_lib) as d:
            d.set_route(route)
            assert d.decompress(got) == a.tobytes()

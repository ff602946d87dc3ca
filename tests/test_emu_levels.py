"""Levels 2-4 (greedy hc_matchfinder; 3 is gzp's default level) through the emulated kernels:
golden vectors of the libdeflate 1.10 binary + oracle comparisons.  No GPU."""
import hashlib

import numpy as np
import pytest

from gzp_amd import _native, synth


def hetero(n, seed):
    """Unlike segments back to back: makes should_end_block split sub-blocks and the split-off
    part need another min_len (a second match/parse round)."""
    rng = np.random.default_rng(seed)
    names = ["dna", "random", "text", "zeros", "lowent", "fastq", "ascii", "runs"]
    parts, size = [], 0
    while size < n:
        ln = int(rng.integers(3000, 60000))
        parts.append(synth.make(names[rng.integers(len(names))], ln, int(rng.integers(1 << 30))))
        size += ln
    return np.ascontiguousarray(np.concatenate(parts)[:n])


def _three_routes(lib, oracle, cases):
    """Levels 2-4 by every route to k_match_hc's arrays (round 5): the default (k_match_hc_sparse at levels 3-4, which
    searches noise and orphan-candidate blocks the dense way itself; the dense kernel at level 2), the dense kernel for
    every block (Config.debug bit 4: rounds 2-4's path) and the sparse kernel forced on every block at every greedy level,
    noise included (bit 5) -- one stream, the oracle's."""
    stale = {}  # blocks k_parse_hc listed for k_match_hc_stale (Scratch.redo's count at these levels), per case and route
    for name, a, level, fmt, ofmt, bs in cases:
        want = oracle.compress_stream(a, ofmt, level, oracle.COMPAT_1_24, bs)
        for flags in (0, 16, 32):
            with _native.Context(format=fmt, level=level, buffer_size=bs, compat=_native.COMPAT_1_24, lib=lib,
                                 max_slab_bytes=max(a.size, 1)) as c:
                c.debug_set_flags(flags)
                assert c.compress_slab(a, True) == want, (name, level, flags)
                stale[(name, flags)] = c.debug_redo_count()
    return stale


def _route_cases(scale=1):
    import numpy as np
    B, M = (_native.FORMAT_BGZF, 0), (_native.FORMAT_MGZIP, 1)
    orphan = synth.make("repeats", 163416, 228638812)[:70000 * scale].copy()  # starts " oeh": hash4 bucket 0 (k_hc_orphan)
    out = [("text", synth.make("text", (2 * 65280 + 777) * scale, 31), 3) + B + (65280,),
           ("text-l4", synth.make("text", 140000 * scale, 32), 4) + B + (65280,),
           ("text-l2", synth.make("text", 70000 * scale, 33), 2) + B + (65280,),
           ("noise", synth.make("ascii", 150000 * scale, 34), 3) + M + (131072,),            # the census sends it the dense way
           ("hetero", hetero(260000 * scale, 35), 3) + M + (300001,),                         # min_len changes: kHcArraysStale
           ("hetero-l4", hetero(200000 * scale, 36), 4) + B + (65280,),
           ("orphan", orphan, 3) + M + (65536,),
           ("dna", synth.make("dna", 100000 * scale, 37), 4) + B + (65280,),                  # open chains everywhere: the sample sends it the dense way
           ("fastq", synth.make("fastq", 90000 * scale, 41), 3) + B + (65280,),
           ("lowent", synth.make("lowent", 70000 * scale, 42), 3) + M + (65536,),
           ("runs", synth.make("runs", 80000 * scale, 43), 4) + B + (65280,),                  # long matches: segments overshot, dense from that tile on
           ("period2", synth.make("period2", 70000, 44), 3) + B + (65280,),
           ("text-then-runs", np.concatenate([synth.make("text", 30000, 45), synth.make("runs", 40000, 46)]), 3) + B + (65280,),
           ("text-then-dna", np.concatenate([synth.make("text", 28000, 47), synth.make("dna", 60000, 48)]), 3) + M + (131072,),
           ("short", synth.make("text", 3000, 38), 3) + B + (65280,),
           ("tile-edge", synth.make("text", 13056 * 2 + 1, 39), 3) + B + (65280,),            # k_match_hc_sparse's tile is 13,056 positions
           ("zeros", synth.make("zeros", 40000, 40), 3) + B + (65280,)]
    return out


def test_greedy_levels_by_every_route(emu_lib, oracle):
    stale = _three_routes(emu_lib, oracle, _route_cases())
    # the path through k_match_hc_stale (a sub-block with another min_len behind a compacted start) is really taken ...
    assert stale[("hetero", 0)] + stale[("hetero", 32)] + stale[("hetero-l4", 0)] + stale[("hetero-l4", 32)] >= 3, stale
    assert not any(v for (_, flags), v in stale.items() if flags == 16), stale  # ... and never behind the dense kernel


def test_blocks_of_unlike_segments_take_the_stale_path(emu_lib, oracle):
    """A short slice of tests/fuzz_classes.stale_path_slice through the emulated kernels (the MI355X runs the long one,
    tests/test_gpu_fuzz_slice.py; tools/emu_fuzz_stale.py is the open-ended hunt)."""
    import fuzz_classes as fc
    blocks, stale = fc.stale_path_slice(emu_lib, oracle, np.random.default_rng(20260928), 10, block_sizes=(65536, 131072),
                                        max_n=200_000)
    assert blocks >= 10 and stale >= 3, (blocks, stale)


def test_golden_raw_deflate_levels(emu_lib, golden_hc):
    comps = {L: _native.Compressor(L, _native.COMPAT_1_10, lib=emu_lib) for L in (0, 2, 3, 4)}
    for e in golden_hc["raw_deflate"]:
        if e["n"] >= 400000 and (e["level"] != 3 or e["class"] != "text"):
            continue  # the big ones run on the GPU (tests/test_gpu_levels.py); one stays here
        a = synth.make(e["class"], e["n"], e["seed"])
        assert hashlib.sha256(comps[e["level"]].deflate_compress(a)).hexdigest() == e["sha256"], e
    for c in comps.values():
        c.close()


def test_golden_streams_levels(emu_lib, golden_hc):
    for e in golden_hc["streams"]:
        a = synth.make(e["class"], e["n"], e["seed"])
        fmt = _native.FORMAT_BGZF if e["fmt"] == "bgzf" else _native.FORMAT_MGZIP
        with _native.Context(format=fmt, level=e["level"], buffer_size=e["buffer_size"],
                             compat=_native.COMPAT_1_10, lib=emu_lib, max_slab_bytes=max(a.size, 1)) as c:
            out, sizes = c.compress_slab(a, True, return_block_sizes=True)
        assert hashlib.sha256(out).hexdigest() == e["sha256"], e
        assert list(sizes) == e["block_sizes"]


@pytest.mark.parametrize("level", [2, 3, 4])
def test_heterogeneous_blocks_vs_oracle(emu_lib, oracle, level):
    cases = [(_native.FORMAT_BGZF, 0, 65280, 4 * 65280 + 99), (_native.FORMAT_MGZIP, 1, 300001, 700000)]
    if level == 3:
        cases.append((_native.FORMAT_MGZIP, 1, 1 << 20, (1 << 20) + 4321))
    for fmt, ofmt, bs, n in cases:
        a = hetero(n, 10 * level + bs % 7)
        for compat in ((_native.COMPAT_1_10, _native.COMPAT_1_24) if level == 3 else (_native.COMPAT_1_10,)):
            with _native.Context(format=fmt, level=level, buffer_size=bs, compat=compat, lib=emu_lib,
                                 max_slab_bytes=n) as c:
                got = c.compress_slab(a, True)
            assert got == oracle.compress_stream(a, ofmt, level, compat, bs), (level, fmt, bs, compat)


def test_default_level_of_the_builder_is_3(emu_lib, oracle):
    import io
    from gzp_amd import par
    a = synth.make("text", 150000, 8)
    sink = io.BytesIO()
    w = par.ParCompressBuilder(par.Bgzf, lib=emu_lib).compat(_native.COMPAT_1_10).from_writer(sink)  # level 3
    w.write_all(a)
    w.finish()
    w.close()
    assert sink.getvalue() == oracle.compress_stream(a, oracle.FMT_BGZF, 3, oracle.COMPAT_1_10, 65280)


# ---- levels 5-9: the lazy / lazy2 parsers (k_match_hc's depth variants + k_parse_lazy)

def test_golden_raw_deflate_lazy_levels(emu_lib, golden_lazy):
    comps = {L: _native.Compressor(L, _native.COMPAT_1_10, lib=emu_lib) for L in (5, 6, 7, 8, 9)}
    for e in golden_lazy["raw_deflate"]:
        if e["n"] > 140000 or (e["n"] > 11000 and e["class"] not in ("text", "fastq", "mixed", "repeats", "dna")):
            continue  # the rest runs on the GPU (tests/test_gpu_levels.py)
        a = synth.make(e["class"], e["n"], e["seed"])
        assert hashlib.sha256(comps[e["level"]].deflate_compress(a)).hexdigest() == e["sha256"], e
    for c in comps.values():
        c.close()


def test_golden_streams_lazy_levels(emu_lib, golden_lazy):
    for e in golden_lazy["streams"]:
        if e["n"] > 320000:
            continue
        a = synth.make(e["class"], e["n"], e["seed"])
        fmt = _native.FORMAT_BGZF if e["fmt"] == "bgzf" else _native.FORMAT_MGZIP
        with _native.Context(format=fmt, level=e["level"], buffer_size=e["buffer_size"],
                             compat=_native.COMPAT_1_10, lib=emu_lib, max_slab_bytes=max(a.size, 1)) as c:
            out, sizes = c.compress_slab(a, True, return_block_sizes=True)
        assert hashlib.sha256(out).hexdigest() == e["sha256"], e
        assert list(sizes) == e["block_sizes"]


@pytest.mark.parametrize("level", [5, 6, 8])
def test_heterogeneous_blocks_vs_oracle_lazy(emu_lib, oracle, level):
    """Sub-block splits, min_len recalculation (every 10000+ bytes) and >300000-byte blocks."""
    for fmt, ofmt, bs, n in [(_native.FORMAT_BGZF, 0, 65280, 2 * 65280 + 99), (_native.FORMAT_MGZIP, 1, 330001, 400000)]:
        a = hetero(n, 10 * level + bs % 7)
        with _native.Context(format=fmt, level=level, buffer_size=bs, compat=_native.COMPAT_1_24, lib=emu_lib,
                             max_slab_bytes=n) as c:
            got = c.compress_slab(a, True)
        assert got == oracle.compress_stream(a, ofmt, level, _native.COMPAT_1_24, bs), (level, fmt, bs)


def test_regression_50000_sequences_end_a_sub_block_between_literals(emu_lib, oracle):
    # found by the GPU soak: the token after the 50000th match of a sub-block starts the next one
    # also when it is a literal (DNA at level 2: > 50000 matches in 600000 bytes)
    a = synth.make("dna", 600000, 3)
    with _native.Context(format=_native.FORMAT_MGZIP, level=2, buffer_size=1 << 20, compat=_native.COMPAT_1_10,
                         lib=emu_lib, max_slab_bytes=a.size) as c:
        got = c.compress_slab(a, True)
    assert got == oracle.compress_stream(a, oracle.FMT_MGZIP, 2, oracle.COMPAT_1_10, 1 << 20)


# ---- levels 10-12: the near-optimal parser (k_near_optimal, one lane per block)

def test_golden_raw_deflate_near_optimal_levels(emu_lib, golden_near_optimal):
    comps = {L: _native.Compressor(L, _native.COMPAT_1_10, lib=emu_lib) for L in (10, 11, 12)}
    for e in golden_near_optimal["raw_deflate"]:
        if e["n"] > 140000 or (e["n"] > 11000 and e["class"] not in ("text", "fastq", "mixed", "repeats", "dna")):
            continue  # the rest runs on the GPU (tests/test_gpu_levels.py)
        a = synth.make(e["class"], e["n"], e["seed"])
        assert hashlib.sha256(comps[e["level"]].deflate_compress(a)).hexdigest() == e["sha256"], e
    for c in comps.values():
        c.close()


def test_golden_streams_near_optimal_levels(emu_lib, golden_near_optimal):
    for e in golden_near_optimal["streams"]:
        if e["n"] > 320000:
            continue
        a = synth.make(e["class"], e["n"], e["seed"])
        fmt = _native.FORMAT_BGZF if e["fmt"] == "bgzf" else _native.FORMAT_MGZIP
        with _native.Context(format=fmt, level=e["level"], buffer_size=e["buffer_size"],
                             compat=_native.COMPAT_1_10, lib=emu_lib, max_slab_bytes=max(a.size, 1)) as c:
            out, sizes = c.compress_slab(a, True, return_block_sizes=True)
        assert hashlib.sha256(out).hexdigest() == e["sha256"], e
        assert list(sizes) == e["block_sizes"]


@pytest.mark.parametrize("level", [10, 12])
def test_heterogeneous_blocks_vs_oracle_near_optimal(emu_lib, oracle, level):
    """DEFLATE block splits with the rewind to the previous check, and blocks above the 300000-byte soft limit."""
    for fmt, ofmt, bs, n in [(_native.FORMAT_BGZF, 0, 65280, 2 * 65280 + 99), (_native.FORMAT_MGZIP, 1, 330001, 400000)]:
        a = hetero(n, 10 * level + bs % 7)
        # whatever the caller asks for, levels 10-12 run libdeflate 1.10's rules throughout (the parser is 1.10's):
        # the stream is the 1.10 binary's, never a hybrid of two versions
        for compat in (_native.COMPAT_1_10, _native.COMPAT_1_24):
            with _native.Context(format=fmt, level=level, buffer_size=bs, compat=compat, lib=emu_lib, max_slab_bytes=n) as c:
                assert c.active_compat() == _native.COMPAT_1_10
                got = c.compress_slab(a, True)
            assert got == oracle.compress_stream(a, ofmt, level, oracle.COMPAT_1_10, bs), (level, fmt, bs, compat)
    with _native.Context(format=_native.FORMAT_BGZF, level=9, compat=_native.COMPAT_1_24, lib=emu_lib, max_slab_bytes=100) as c:
        assert c.active_compat() == _native.COMPAT_1_24  # levels 0-9: the caller's choice

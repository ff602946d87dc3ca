"""A seeded slice of the randomised GPU parity soak (tools/gpu_fuzz.py) under -m gpu: random
(class, size, level 0-9, format, block size, compat) through the real library against the oracle,
byte for byte, each result also inflated on the GPU; plus the regression cases the soak has found."""
import os
import sys

import pytest

from gzp_amd import _native, synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import gpu_fuzz  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [20250927, 20250928, 20250929])
def test_seeded_fuzz_slice(hip_lib, oracle, seed):
    cases, bad = gpu_fuzz.fuzz(hip_lib, oracle, seed=seed, max_cases=900, max_n=600_000, verbose=False)
    assert cases == 900 and not bad, bad[:5]


def _raw_vs_oracle(hip_lib, oracle, cases, levels=range(0, 10)):
    """(array, level, compat) cases through the libdeflate-shaped ABI against the oracle, byte for byte."""
    comps = {}
    try:
        for it, (a, level, compat) in enumerate(cases):
            if (level, compat) not in comps:
                comps[(level, compat)] = _native.Compressor(level, compat, lib=hip_lib)
            got = comps[(level, compat)].deflate_compress(a)
            assert got == oracle.deflate_compress(a, level, compat), (it, level, compat, a.size, bytes(a[:8]))
    finally:
        for c in comps.values():
            c.close()


def test_first_bytes_matter_for_every_matchfinder(hip_lib, oracle):
    """Round 5: buffers that start in bucket 0 of the hc hash4 / hc hash3 / level-1 table -- the class whose
    absence let round 2's position-0 bug live until a new soak seed met it -- levels 0-9, both compat rules.  (The
    pre-fix k_match_hc fails tests/test_gpu_orphan.py's slice of the hash4 starts 146 times in 300; this slice covers
    the other two tables and the level-1 kernels the same way.)"""
    import numpy as np
    import fuzz_classes as fc
    rng = np.random.default_rng(20260928)
    _raw_vs_oracle(hip_lib, oracle, ((fc.first_bytes(rng), int(rng.integers(0, 10)), int(rng.integers(0, 2)))
                                     for _ in range(1200)))


def test_copies_at_the_distance_and_length_thresholds(hip_lib, oracle):
    """Copies 32,765 ... 32,770 / 4,096 / 4,097 / 8,192 / 8,193 bytes back and 3 / 4 / nice_len +- 1 / 257 ... 259 bytes
    long, some ending with the buffer: the window edge, the lazy parsers' distance rules, every level's nice_match_length."""
    import numpy as np
    import fuzz_classes as fc
    rng = np.random.default_rng(20260929)
    _raw_vs_oracle(hip_lib, oracle, ((fc.thresholds(rng), int(rng.integers(0, 10)), int(rng.integers(0, 2)))
                                     for _ in range(500)))


def test_sub_blocks_that_fill_up_exactly(hip_lib, oracle):
    """Thousands of short matches, the buffer cut 0 ... 5 bytes behind (and just in front of) the 8,192nd match of level 1
    (its sequence store) and the 50,000th of levels 2-9 (SEQ_STORE_LENGTH): the token behind a full sub-block, trailing
    literals and the end of the buffer in every order."""
    import numpy as np
    import fuzz_classes as fc
    rng = np.random.default_rng(20260930)
    cases = []
    for _ in range(8):
        cases += [(a, 1, int(rng.integers(0, 2))) for a in fc.full_sub_block_cuts(rng, 1, oracle)]
    for lv in (2, 3, 5, 6, 8, 9):
        cases += [(a, lv, int(rng.integers(0, 2))) for a in fc.full_sub_block_cuts(rng, lv, oracle, deltas=(-1, 0, 1, 2, 4, 300))]
    _raw_vs_oracle(hip_lib, oracle, cases)


def test_blocks_of_unlike_segments_take_the_stale_path(hip_lib, oracle):
    """Round 5: streams of unlike segments back to back (text, DNA, noise, runs ...), levels 2-4 by the default and the
    forced-sparse route, blocks of 64 KiB ... 1 MiB: a split-off sub-block with another min_len behind a compacted start
    sends the block through k_match_hc_stale (its pieces dealt out to every CU) -- the whole stream against the oracle."""
    import numpy as np
    import fuzz_classes as fc
    blocks, stale = fc.stale_path_slice(hip_lib, oracle, np.random.default_rng(20260929), 250)
    assert blocks >= 250 and stale >= 60, (blocks, stale)


def test_regression_soft_limit_boundary_on_a_tile_edge(hip_lib, oracle):
    # found by the soak in round 1: a sub-block that starts on the first position of a 64 KiB parse
    # tile ends (65535-byte soft limit) on the tile's last position
    a = synth.make("mixed", 256005, 514286759)
    with _native.Context(format=_native.FORMAT_MGZIP, level=1, buffer_size=285614, lib=hip_lib,
                         max_slab_bytes=a.size) as c:
        got = c.compress_slab(a, True)
    assert got == oracle.compress_stream(a, oracle.FMT_MGZIP, 1, oracle.COMPAT_1_24, 285614)

"""Buffers whose first bytes hash to bucket 0 of a matchfinder table -- hc hash4, hc hash3, the level-1 table -- through
the emulated kernels, levels 0-9: a seeded slice of tools/emu_fuzz.py (libdeflate files position 0 under bucket 0 of
every table, so these starts are where the first position behaves unlike any other).  No GPU."""
import os
import sys

import numpy as np

from gzp_amd import _native

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


def test_bucket0_starts_vs_oracle(emu_lib, oracle):
    import emu_fuzz
    h4, h15, h3 = emu_fuzz.starts()
    assert h4 and h15 and h3
    rng = np.random.default_rng(4)
    comps = {}
    for it in range(250):
        a = emu_fuzz.case(rng, h4, h15, h3)
        level, compat = int(rng.integers(0, 10)), int(rng.integers(0, 2))
        if (level, compat) not in comps:
            comps[(level, compat)] = _native.Compressor(level, compat, lib=emu_lib)
        assert comps[(level, compat)].deflate_compress(a) == oracle.deflate_compress(a, level, compat), (it, level, compat, a.size)
    for c in comps.values():
        c.close()


def test_thresholds_and_full_sub_blocks_vs_oracle(emu_lib, oracle):
    """The other two committed classes of tests/fuzz_classes.py through the emulated kernels (a short slice: the MI355X
    runs the long one, tests/test_gpu_fuzz_slice.py)."""
    import fuzz_classes as fc
    rng = np.random.default_rng(5)
    cases = [(fc.thresholds(rng), int(rng.integers(0, 10)), int(rng.integers(0, 2))) for _ in range(40)]
    cases += [(a, 1, int(rng.integers(0, 2))) for a in fc.full_sub_block_cuts(rng, 1, oracle)]  # the 8,192nd match + k bytes
    cases += [(a, 3, 1) for a in fc.full_sub_block_cuts(rng, 3, oracle, deltas=(0, 2))]          # the 50,000th at level 3
    comps = {}
    for it, (a, level, compat) in enumerate(cases):
        if (level, compat) not in comps:
            comps[(level, compat)] = _native.Compressor(level, compat, lib=emu_lib)
        assert comps[(level, compat)].deflate_compress(a) == oracle.deflate_compress(a, level, compat), (it, level, compat, a.size)
    for c in comps.values():
        c.close()

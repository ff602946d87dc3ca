"""Position 0 and the hash3 gate (tests/test_emu_orphan.py) on the MI355X, through libgzpx.so: the soak's case as a whole
Mgzip stream and a longer slice of the targeted cases, levels 2-9."""
import numpy as np
import pytest

from gzp_amd import _native, synth
from test_emu_orphan import STARTS, _case

pytestmark = pytest.mark.gpu


def test_the_soak_case_as_a_stream(hip_lib, oracle):
    a = synth.make("repeats", 163416, 228638812)
    for level in (3, 7, 9):
        with _native.Context(format=_native.FORMAT_MGZIP, level=level, buffer_size=65536, compat=_native.COMPAT_1_24,
                             lib=hip_lib, max_slab_bytes=a.size) as c:
            assert c.compress_slab(a, True) == oracle.compress_stream(a, 1, level, _native.COMPAT_1_24, 65536), level


def test_orphan_matches_vs_oracle(hip_lib, oracle):
    rng = np.random.default_rng(7)
    comps = {}
    for it in range(400):
        start = STARTS[rng.integers(len(STARTS))]
        wide = bool(rng.random() < 0.4)
        a = _case(rng, start, wide, int(rng.integers(200, 40000)))
        level, compat = int(rng.integers(2, 10)), int(rng.integers(0, 2))
        if (level, compat) not in comps:
            comps[(level, compat)] = _native.Compressor(level, compat, lib=hip_lib)
        assert comps[(level, compat)].deflate_compress(a) == oracle.deflate_compress(a, level, compat), (it, start, wide, level, compat)
    for c in comps.values():
        c.close()

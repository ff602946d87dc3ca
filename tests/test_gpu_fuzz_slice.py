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


def test_regression_soft_limit_boundary_on_a_tile_edge(hip_lib, oracle):
    # found by the soak in round 1: a sub-block that starts on the first position of a 64 KiB parse
    # tile ends (65535-byte soft limit) on the tile's last position
    a = synth.make("mixed", 256005, 514286759)
    with _native.Context(format=_native.FORMAT_MGZIP, level=1, buffer_size=285614, lib=hip_lib,
                         max_slab_bytes=a.size) as c:
        got = c.compress_slab(a, True)
    assert got == oracle.compress_stream(a, oracle.FMT_MGZIP, 1, oracle.COMPAT_1_24, 285614)

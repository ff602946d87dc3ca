"""The CPU leg of bench.py (oracle/cpu_bench.c): native worker threads over the oracle port, and over
the image's libdeflate binary where there is one.  No GPU."""
import numpy as np

from gzp_amd import synth


def _scan(comp):
    offs, sizes, p = [], [], 0
    while p < comp.size:
        bsize = int(comp[p + 16]) | (int(comp[p + 17]) << 8)
        offs.append(p)
        sizes.append(bsize + 1)
        p += bsize + 1
    return np.array(offs, dtype=np.uint64), np.array(sizes, dtype=np.uint32)


def test_native_cpu_baselines(oracle):
    a = synth.text_slab(6 * 65280 + 123, seed=3)
    nbytes, dt, used = oracle.cpu_bench_compress(a, threads=2, wall_s=0.2)
    assert used == 2 and nbytes >= a.size and 0.15 < dt < 5.0
    r = oracle.cpu_bench_compress_ref(a, 1, 65280, threads=2, wall_s=0.2)
    if r is not None:  # the image ships libdeflate.so.0; a box without it falls back to the port
        assert r[2] == 2 and r[0] >= a.size
    comp = np.frombuffer(oracle.compress_stream(a, oracle.FMT_BGZF, 1, oracle.COMPAT_1_24, 65280), dtype=np.uint8)
    offs, sizes = _scan(comp)
    r = oracle.cpu_bench_inflate(comp, offs, sizes, 18, threads=2, wall_s=0.2)
    if r is not None:
        assert r[2] == 2 and r[0] >= a.size  # every worker inflated (and CRC-checked) its blocks at least once


def test_available_cores_is_sane():
    import bench
    n, note = bench.available_cores()
    assert n >= 1 and isinstance(note, str)


def test_traffic_figures_are_only_taken_from_this_build_of_the_library():
    """profiles/pmc_traffic.json records the gzpx_build_id it was collected with; bench.py uses its counters only for a
    library of that very build (a kernel change must not leave a stale `roofline.traffic` in the line)."""
    import json
    import os

    import bench
    from gzp_amd import build
    with open(os.path.join(bench.ROOT, "profiles", "pmc_traffic.json")) as f:
        doc = json.load(f)
    have = doc["build_id"]
    assert isinstance(have, str) and len(have) == 16
    assert bench.pmc_traffic("pipeline", have) == doc["hbm_bytes_per_launch"]["pipeline"]
    assert bench.pmc_traffic("pipeline", "0123456789abcdef") is None
    assert "refused" in bench.pmc_traffic_source("0123456789abcdef")
    assert have in bench.pmc_traffic_source(have)
    # the committed counters belong to the committed sources
    assert have == build.source_id(), "kernel sources changed: re-collect profiles/pmc_traffic.json (tools/pmc_traffic_all.sh + tools/summarize_profiles.py)"

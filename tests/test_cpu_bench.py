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


def test_parcompress_shaped_cpu_baseline_writes_gzps_stream(oracle):
    """oracle/cpu_bench.c's ParCompress<Bgzf> twin over the libdeflate binary (bench.py's `cpu_baseline_parcompress`):
    64 KiB write_all calls, the strict `>` cut (src/par/compress.rs:415), flush_last(true) with at least one -- maybe
    empty -- last block + EOF (:332-362), N workers, in-order writer.  Its stream is the oracle's (compat 1.10: the
    binary's version) for every edge of the cut rule, whatever the number of workers."""
    for n in (0, 1, 65279, 65280, 65281, 65536, 2 * 65280, 2 * 65280 + 1234, 9 * 65280 + 77):
        a = synth.text_slab(n, seed=5) if n else np.zeros(0, dtype=np.uint8)
        want = oracle.compress_stream(a, oracle.FMT_BGZF, 1, oracle.COMPAT_1_10, 65280)
        for threads in (1, 3):
            r = oracle.cpu_bench_parcompress_ref(a, 1, 65280, 65536, threads=threads, wall_s=0.0)
            if r is None:
                return  # no libdeflate binary on this box
            nbytes, dt, passes, stream = r
            assert passes == 1 and nbytes == n and stream.tobytes() == want, (n, threads)
    # other write sizes cut the same stream (the cut rule looks at the buffered length only)
    a = synth.text_slab(5 * 65280 + 4321, seed=6)
    want = oracle.compress_stream(a, oracle.FMT_BGZF, 1, oracle.COMPAT_1_10, 65280)
    for chunk in (1000, 65280, 200000):
        assert oracle.cpu_bench_parcompress_ref(a, 1, 65280, chunk, threads=2, wall_s=0.0)[3].tobytes() == want, chunk


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

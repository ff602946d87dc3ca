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

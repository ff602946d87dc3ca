"""Fuzz the CPU oracle (compat=1.10) against the libdeflate binary of this image.

Container-only tool (the GPU box is not assumed to have libdeflate); the pinned results
travel as tests/golden/ fixtures written by tests/golden/make_golden.py.
"""
import ctypes, sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import oracle
from gzp_amd import synth

LD = ctypes.CDLL(os.environ.get("LIBDEFLATE_SO", "/lib/x86_64-linux-gnu/libdeflate.so.0"))
LD.libdeflate_alloc_compressor.restype = ctypes.c_void_p
LD.libdeflate_alloc_compressor.argtypes = [ctypes.c_int]
LD.libdeflate_deflate_compress.restype = ctypes.c_size_t
LD.libdeflate_deflate_compress.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                           ctypes.c_void_p, ctypes.c_size_t]
LD.libdeflate_crc32.restype = ctypes.c_uint32
LD.libdeflate_crc32.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t]
_comp = {}

def ld_compress(a, level):
    if level not in _comp:
        _comp[level] = LD.libdeflate_alloc_compressor(level)
    out = np.empty(a.size + a.size // 8 + 1024, dtype=np.uint8)
    n = LD.libdeflate_deflate_compress(_comp[level], a.ctypes.data, a.size, out.ctypes.data, out.size)
    assert n > 0
    return out[:n].tobytes()

def main():
    level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    rng = np.random.default_rng(12345)
    sizes = [0, 1, 2, 4, 5, 51, 52, 53, 100, 300, 511, 512, 513, 1000, 4096, 5000, 32767, 32768,
             32769, 32773, 40000, 65279, 65280, 65536, 70534, 70535, 70536, 100000, 131072, 300000]
    bad = 0
    t0 = time.time()
    n_cases = 0
    for cls in synth.CLASSES:
        for n in sizes:
            a = synth.make(cls, n, seed=n + 17)
            ref = ld_compress(a, level)
            got = oracle.deflate_compress(a, level, oracle.COMPAT_1_10)
            n_cases += 1
            if ref != got:
                bad += 1
                print("MISMATCH class=%s n=%d ref=%d got=%d" % (cls, n, len(ref), len(got)))
            assert LD.libdeflate_crc32(0, a.ctypes.data, a.size) == oracle.crc32(a)
    names = list(synth.CLASSES)
    for it in range(iters):
        cls = names[rng.integers(len(names))]
        n = int(rng.integers(0, 140000)) if it % 3 else int(rng.integers(0, 3000))
        seed = int(rng.integers(1 << 30))
        a = synth.make(cls, n, seed=seed)
        # splice in slices of itself to create long-distance repeats
        if n > 2000 and it % 2:
            k = int(rng.integers(1, 6))
            a = a.copy()
            for _ in range(k):
                s = int(rng.integers(0, n - 600)); d = int(rng.integers(0, n - 600)); ln = int(rng.integers(4, 600))
                a[d:d + ln] = a[s:s + ln]
        ref = ld_compress(a, level)
        got = oracle.deflate_compress(a, level, oracle.COMPAT_1_10)
        n_cases += 1
        if ref != got:
            bad += 1
            print("MISMATCH fuzz class=%s n=%d seed=%d ref=%d got=%d" % (cls, n, seed, len(ref), len(got)))
    print("cases=%d mismatches=%d  %.1fs" % (n_cases, bad, time.time() - t0))
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())

"""Randomised parity soak: random (class, size, level, format, block size, compat) -> the library vs
the oracle, byte for byte, plus GPU inflate of the result against the input.
usage: gpu_fuzz.py [seconds] [seed]          (tests/test_gpu_fuzz_slice.py runs a seeded slice)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

from gzp_amd import _native, synth


def fuzz(lib, oracle, seed=12345, secs=None, max_cases=None, max_n=1_500_000, verbose=True, max_level=9, min_level=0,
         debug_flags=0):
    """Returns (cases, failures); failures is a list of case tuples."""
    rng = np.random.default_rng(seed)
    classes = sorted(synth.CLASSES)
    t_end = time.time() + secs if secs else None
    cases = 0
    bad = []
    ctxs = {}
    dctx = {}
    while (t_end is None or time.time() < t_end) and (max_cases is None or cases < max_cases):
        cls = classes[rng.integers(len(classes))]
        level = int(rng.integers(min_level, max_level + 1))
        fmt = int(rng.integers(0, 2))
        if fmt == 0:
            bs = int(rng.choice([65280, 65280, 32768 + int(rng.integers(0, 32000)), 40000]))
        else:
            bs = int(rng.choice([131072, 65536, 200000, 1 << 20, 32768 + int(rng.integers(0, 400000))]))
        r = rng.random()
        n = int(rng.integers(0, 300)) if r < 0.15 else int(rng.integers(0, 4 * bs + 5000)) if r < 0.9 else int(
            rng.integers(0, 1_500_000))
        n = min(n, max_n)
        compat = int(rng.integers(0, 2))
        a = synth.make(cls, n, int(rng.integers(1, 1 << 30)))
        if cls == "random" and fmt == 0 and bs > 65280:
            continue  # incompressible data in a BGZF block that large is BlockSizeExceeded by design
        key = (fmt, level, bs, compat)
        if key not in ctxs:
            if len(ctxs) > 24:
                for c in ctxs.values():
                    c.close()
                ctxs.clear()
            ctxs[key] = _native.Context(format=fmt, level=level, buffer_size=bs, compat=compat, lib=lib,
                                        max_slab_bytes=2_000_000)
            if debug_flags:  # (e.g. 32: k_match_hc_sparse at every greedy level and for every block; 16: never)
                ctxs[key].debug_set_flags(debug_flags)
        try:
            got = ctxs[key].compress_slab(a, True)
        except _native.GzpxError as e:
            if e.code == _native.ERR_BLOCK_SIZE_EXCEEDED and fmt == 0:
                continue
            raise
        want = oracle.compress_stream(a, fmt, level, compat, bs)
        cases += 1
        case = (cls, n, level, fmt, bs, compat)
        if got != want:
            bad.append(("MISMATCH",) + case)
            if verbose:
                print("MISMATCH", *case, flush=True)
            continue
        if fmt not in dctx:
            dctx[fmt] = _native.DContext(format=fmt, lib=lib)
        if dctx[fmt].decompress(got) != a.tobytes():
            bad.append(("INFLATE MISMATCH",) + case)
            if verbose:
                print("INFLATE MISMATCH", *case, flush=True)
    for c in list(ctxs.values()) + list(dctx.values()):
        c.close()
    return cases, bad


if __name__ == "__main__":
    from oracle import oracle
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
    cases, bad = fuzz(_native.load(), oracle, seed, secs=secs)
    print("gpu_fuzz: %d cases, %d failures" % (cases, len(bad)))
    sys.exit(1 if bad else 0)

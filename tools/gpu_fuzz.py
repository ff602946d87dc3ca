"""Randomised parity soak on the GPU: random (class, size, level, format, block size, compat) ->
libgzpx.so vs the oracle, byte for byte, plus GPU inflate of the result against the input.
usage: gpu_fuzz.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

from gzp_amd import _native, synth
from oracle import oracle

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
lib = _native.load()
classes = sorted(synth.CLASSES)
t_end = time.time() + secs
cases = bad = 0
ctxs = {}
while time.time() < t_end:
    cls = classes[rng.integers(len(classes))]
    level = int(rng.integers(0, 5))
    fmt = int(rng.integers(0, 2))
    if fmt == 0:
        bs = int(rng.choice([65280, 65280, 32768 + int(rng.integers(0, 32000)), 40000]))
    else:
        bs = int(rng.choice([131072, 65536, 200000, 1 << 20, 32768 + int(rng.integers(0, 400000))]))
    r = rng.random()
    n = int(rng.integers(0, 300)) if r < 0.15 else int(rng.integers(0, 4 * bs + 5000)) if r < 0.9 else int(
        rng.integers(0, 1_500_000))
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
    try:
        got = ctxs[key].compress_slab(a, True)
    except _native.GzpxError as e:
        if e.code == _native.ERR_BLOCK_SIZE_EXCEEDED and fmt == 0:
            continue
        raise
    want = oracle.compress_stream(a, fmt, level, compat, bs)
    cases += 1
    if got != want:
        bad += 1
        print("MISMATCH", cls, n, level, fmt, bs, compat, flush=True)
        continue
    with _native.DContext(format=fmt, lib=lib) as d:
        if d.decompress(got) != a.tobytes():
            bad += 1
            print("INFLATE MISMATCH", cls, n, level, fmt, bs, compat, flush=True)
print("gpu_fuzz: %d cases, %d failures" % (cases, bad))
sys.exit(1 if bad else 0)

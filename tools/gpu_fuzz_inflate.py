"""Randomised soak of the GPU inflate path with FOREIGN streams: BGZF members made by zlib at random
levels / strategies / memLevels (stored, fixed and dynamic blocks, many sub-blocks, long distances),
concatenated and inflated by libgzpx.so, compared with the input.  usage: gpu_fuzz_inflate.py [seconds] [seed]"""
import os
import struct
import sys
import time
import zlib

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

from gzp_amd import _native, synth

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
# (GZPX_LIB=<path>: another build of the library, e.g. tests/emu/libgzpx_emu.so -- the same soak without a GPU)
lib = _native.GzpxLib(os.environ["GZPX_LIB"]) if os.environ.get("GZPX_LIB") else _native.load()
classes = sorted(synth.CLASSES)
strategies = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]


def member(chunk):
    co = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, -15, int(rng.integers(1, 10)),
                          strategies[rng.integers(len(strategies))])
    parts = []
    pos = 0
    while pos < len(chunk):  # random full flushes: more DEFLATE sub-blocks, empty stored blocks
        step = int(rng.integers(1, 20000))
        parts.append(co.compress(chunk[pos:pos + step]))
        if rng.random() < 0.2:
            parts.append(co.flush(zlib.Z_FULL_FLUSH if rng.random() < 0.5 else zlib.Z_SYNC_FLUSH))
        pos += step
    parts.append(co.flush())
    payload = b"".join(parts)
    if len(payload) + 26 > 65536:
        return None
    hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, ord("B"), ord("C"), 2, len(payload) + 25)
    return hdr + payload + struct.pack("<II", zlib.crc32(chunk), len(chunk))


d = _native.DContext(lib=lib)
t_end = time.time() + secs
cases = bad = 0
while time.time() < t_end:
    data = []
    stream = []
    for _ in range(int(rng.integers(1, 40))):
        cls = classes[rng.integers(len(classes))]
        n = int(rng.integers(0, 60000)) if rng.random() < 0.9 else int(rng.integers(0, 200))
        chunk = synth.make(cls, n, int(rng.integers(1, 1 << 30))).tobytes()
        m = member(chunk)
        if m is None:
            continue
        data.append(chunk)
        stream.append(m)
    if not stream:
        continue
    cases += 1
    try:
        ok = d.decompress(b"".join(stream)) == b"".join(data)
    except _native.GzpxError as e:
        ok = False
        print("ERROR", e, flush=True)
    if not ok:
        bad += 1
        print("INFLATE MISMATCH in a stream of", len(stream), "members", flush=True)
print("gpu_fuzz_inflate: %d streams, %d failures" % (cases, bad))
sys.exit(1 if bad else 0)

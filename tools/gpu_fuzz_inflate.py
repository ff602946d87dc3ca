"""Randomised soak of the GPU inflate path with FOREIGN streams: BGZF members made by zlib at random
levels / strategies / memLevels (stored, fixed and dynamic blocks, many sub-blocks, long distances),
concatenated and inflated by libgzpx.so, compared with the input.  usage: gpu_fuzz_inflate.py [seconds] [seed] [bgzf|mgzip]
(mgzip: members of up to 3 MiB in Mgzip framing -- the eight-waves-per-member form of k_inflate_seg)"""
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
MGZIP = len(sys.argv) > 3 and sys.argv[3] == "mgzip"
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
    if MGZIP:
        hdr = struct.pack("<BBBBIBBHBBHI", 31, 139, 8, 4, 0, 0, 255, 8, ord("I"), ord("G"), 4, len(payload) + 28)
        return hdr + payload + struct.pack("<II", zlib.crc32(chunk), len(chunk))
    if len(payload) + 26 > 65536:
        return None
    hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, ord("B"), ord("C"), 2, len(payload) + 25)
    return hdr + payload + struct.pack("<II", zlib.crc32(chunk), len(chunk))


FMT = _native.FORMAT_MGZIP if MGZIP else _native.FORMAT_BGZF
HDR = 20 if MGZIP else 18
d = _native.DContext(format=FMT, lib=lib)
dw = _native.DContext(format=FMT, lib=lib)  # round 6: the other route (k_inflate for every member) as the judge of damaged streams
dw.set_route(_native.INFLATE_WAVE)
t_end = time.time() + secs
cases = bad = damaged = handed_back = 0


def outcome(ctx, blob):
    try:
        return ("ok", ctx.decompress(blob))
    except _native.GzpxError as e:
        return ("err", e.code, e.block)


while time.time() < t_end:
    data = []
    stream = []
    for _ in range(int(rng.integers(1, 8 if MGZIP else 40))):
        cls = classes[rng.integers(len(classes))]
        n = int(rng.integers(0, 60000)) if rng.random() < 0.9 else int(rng.integers(0, 200))
        if MGZIP:
            n = int(rng.integers(100000, 3 << 20)) if rng.random() < 0.85 else int(rng.integers(0, 70000))
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
    handed_back += d.last_redo_count()
    # the same stream with a few payload bytes damaged: the decode / copy pair (which hands what it cannot take to
    # k_inflate) and k_inflate alone must agree on the bytes or on (error class, first failing member)
    blob = bytearray(b"".join(stream))
    k = int(rng.integers(len(stream)))
    lo = sum(len(m) for m in stream[:k]) + HDR
    hi = lo + len(stream[k]) - HDR - 8
    if hi > lo:
        for _ in range(int(rng.integers(1, 4))):
            blob[int(rng.integers(lo, hi))] ^= int(rng.integers(1, 256))
        damaged += 1
        a, b = outcome(d, bytes(blob)), outcome(dw, bytes(blob))
        if a != b:
            bad += 1
            print("ROUTES DISAGREE on a damaged stream:", a[:1] + a[1:3] if a[0] == "err" else "ok", b[:1] + b[1:3] if b[0] == "err" else "ok", flush=True)
print("gpu_fuzz_inflate: %d streams (%d members handed back to k_inflate), %d damaged copies, %d failures" % (cases, handed_back, damaged, bad))
sys.exit(1 if bad else 0)

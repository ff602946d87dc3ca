"""CPU-only randomised parity hunt through the emulated kernels (tests/emu): small buffers whose FIRST bytes are chosen
to hit the matchfinders' special cases -- libdeflate files position 0 under bucket 0 of every table, so starts whose
hash4 / hash3 / level-1 hash is 0 are where "position 0" behaves unlike any other -- mixed with ordinary synthetic
classes and with copies at the distance / length thresholds of the matchfinders and parsers (case_edges), every level 0-9 (10-12 with --near-optimal), both compat rules, raw DEFLATE against the oracle.
usage: emu_fuzz.py [seconds] [seed] [--near-optimal]      (the round-4 position-0 / hash3-gate bug is the reason it exists)"""
import itertools
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
import numpy as np

import build_emu
from gzp_amd import _native, synth
from oracle import oracle

ALPHA = b" abcdefghijklmnopqrstuvwxyz"


def _mul(v):
    return (v * 0x1E35A7BD) & 0xFFFFFFFF


def starts():
    s4 = [bytes(c) for c in itertools.product(ALPHA, repeat=4)]
    h4 = [c for c in s4 if _mul(int.from_bytes(c, "little")) >> 16 == 0]          # hc hash4 bucket 0
    h15 = [c for c in s4 if _mul(int.from_bytes(c, "little")) >> 17 == 0]         # level-1 bucket 0
    h3 = [bytes(c) for c in itertools.product(bytes(range(32, 127)), repeat=3)  # (no lower-case triple hashes to 0)
          if _mul(int.from_bytes(bytes(c), "little")) >> 17 == 0]
    return h4, h15, h3


def case(rng, h4, h15, h3):
    wide = rng.random() < 0.35
    sym = np.frombuffer(bytes(range(32, 127)) if wide else ALPHA, np.uint8)
    n = int(rng.integers(6, 300)) if rng.random() < 0.1 else int(rng.integers(300, 24000))
    kind = rng.random()
    if kind < 0.25:
        names = sorted(synth.CLASSES)
        body = synth.make(names[rng.integers(len(names))], n, int(rng.integers(1, 1 << 30))).copy()
    else:
        body = sym[rng.integers(0, len(sym), n)].copy()
        pool = [sym[rng.integers(0, len(sym), int(rng.integers(3, 40)))] for _ in range(10)]
        i = 8
        while i < n - 48:
            if rng.random() < 0.55:
                ph = pool[rng.integers(len(pool))]
                body[i:i + len(ph)] = ph
                i += len(ph)
            i += int(rng.integers(1, 12))
    pick = rng.random()
    head = h4[rng.integers(len(h4))] if pick < 0.4 else h15[rng.integers(len(h15))] if pick < 0.6 else \
        h3[rng.integers(len(h3))] + bytes([int(sym[rng.integers(len(sym))])]) if pick < 0.8 else bytes(body[:4])
    if n >= 64:
        body[:4] = np.frombuffer(head, np.uint8)
        for _ in range(int(rng.integers(0, 4))):  # the start again, later, with some of what follows it
            at = int(rng.integers(8, n - 40))
            k = int(rng.integers(3, 24))
            body[at:at + k] = body[:k]
    return np.ascontiguousarray(body)


def case_edges(rng):
    """Copies at distances around the window size and the parsers' distance rules (32,767 +- 3, 4,096 / 4,097, 8,192 / 8,193)
    and of lengths around every level's nice_match_length (and 258), some of them ending with the buffer."""
    sym = np.frombuffer(ALPHA if rng.random() < 0.6 else bytes(range(32, 127)), np.uint8)
    far = rng.random() < 0.4
    n = int(rng.integers(33000, 100000)) if far else int(rng.integers(300, 30000))
    a = sym[rng.integers(0, len(sym), n)].copy()
    lens = [3, 4, 5, 8, 9, 10, 11, 13, 14, 15, 29, 30, 31, 64, 65, 66, 129, 130, 131, 257, 258, 259, 300, 600]
    for _ in range(int(rng.integers(1, 6)) if far else int(rng.integers(2, 40))):
        ln = min(int(rng.choice(lens)), n // 3)
        dist = int(rng.choice([32765, 32766, 32767, 32768, 32769, 32770, 16384, 4096, 4097, 8192, 8193])) if far else \
            int(rng.integers(1, min(n - ln, 9000)))
        if n - ln <= dist:
            continue
        at = n - ln if rng.random() < 0.1 else int(rng.integers(dist, n - ln + 1))
        for i in range(ln):  # (byte by byte: a copy may overlap its source)
            a[at + i] = a[at + i - dist]
    return np.ascontiguousarray(a)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    secs = float(args[0]) if args else 120.0
    seed = int(args[1]) if len(args) > 1 else 1
    levels = list(range(0, 13 if "--near-optimal" in sys.argv else 10))
    lib = _native.GzpxLib(build_emu.build())
    h4, h15, h3 = starts()
    rng = np.random.default_rng(seed)
    comps, t_end, cases, bad = {}, time.time() + secs, 0, 0
    while time.time() < t_end:
        a = case_edges(rng) if rng.random() < 0.25 else case(rng, h4, h15, h3)
        level, compat = int(levels[rng.integers(len(levels))]), int(rng.integers(0, 2))
        if (level, compat) not in comps:
            comps[(level, compat)] = _native.Compressor(level, compat, lib=lib)
        got = comps[(level, compat)].deflate_compress(a)
        want = oracle.deflate_compress(a, level, 1 if level >= 10 else compat)
        cases += 1
        if got != want:
            bad += 1
            name = "/tmp/emu_fuzz_fail_%d_%d.bin" % (seed, cases)
            a.tofile(name)
            print("MISMATCH case %d: n %d level %d compat %d, %d vs %d bytes, input saved to %s" % (cases, a.size, level, compat, len(got), len(want), name), flush=True)
    print("emu_fuzz seed %d: %d cases, %d failures" % (seed, cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

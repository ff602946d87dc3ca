"""Position 0 and the hash3 gate (levels 2-9), through the emulated kernels.  No GPU.

libdeflate files a buffer's first position under bucket 0 of both hc_matchfinder tables.  When the buffer's first four
bytes hash to hash4 bucket 0, the first later position with those bytes has position 0 in its hash4 chain but an empty
hash3 bucket: a search started from best_len < 4 gives up there, one started from best_len >= 4 (min_len >= 5, a lazy
lookahead behind a match of >= 5) finds the match.  The round-4 GPU soak found the case (seed 20260928: 'repeats',
level 7); this is that input and a seeded slice of the targeted fuzz that reproduced it 146 times in 300."""
import itertools

import numpy as np

from gzp_amd import _native, synth

ALPHA = b" abcdefghijklmnopqrstuvwxyz"


def _h4(b):
    return ((int.from_bytes(b, "little") * 0x1E35A7BD) & 0xFFFFFFFF) >> 16


STARTS = [bytes(c) for c in itertools.product(ALPHA, repeat=4) if _h4(bytes(c)) == 0]


def _case(rng, start, wide, n):
    sym = np.frombuffer(bytes(range(32, 127)) if wide else ALPHA, np.uint8)  # wide: >= 80 byte values, min_len 3
    body = sym[rng.integers(0, len(sym), n)].copy()
    pool = [sym[rng.integers(0, len(sym), int(rng.integers(3, 12)))] for _ in range(12)]
    i = 8
    while i < n - 40:
        if rng.random() < 0.5:
            ph = pool[rng.integers(len(pool))]
            body[i:i + len(ph)] = ph
            i += len(ph)
        i += int(rng.integers(1, 9))
    cont = sym[rng.integers(0, len(sym), 12)]
    head = np.frombuffer(start, np.uint8)
    body[:4] = head
    body[4:4 + len(cont)] = cont
    for _ in range(int(rng.integers(1, 4))):  # recurrences of the start, the first one the orphan
        at = int(rng.integers(20, n - 40))
        k = int(rng.integers(0, 13))
        body[at:at + 4] = head
        body[at + 4:at + 4 + k] = cont[:k]
        if rng.random() < 0.5:  # something matchable right in front: the orphan is then a lazy lookahead
            ph = pool[rng.integers(len(pool))]
            body[at - len(ph):at] = ph
    return np.ascontiguousarray(body)


def test_there_are_starts_that_hash_to_bucket_0():
    assert b" oeh" in STARTS and len(STARTS) >= 3


def test_the_soak_case(emu_lib, oracle):
    a = synth.make("repeats", 163416, 228638812)[:65536].copy()
    assert bytes(a[:4]) == b" oeh"
    for level in (3, 7):
        c = _native.Compressor(level, _native.COMPAT_1_10, lib=emu_lib)
        got = c.deflate_compress(a)
        c.close()
        assert got == oracle.deflate_compress(a, level, _native.COMPAT_1_10), level


def test_orphan_matches_vs_oracle(emu_lib, oracle):
    rng = np.random.default_rng(1)
    comps = {}
    for it in range(120):
        start = STARTS[rng.integers(len(STARTS))]
        wide = bool(rng.random() < 0.4)
        a = _case(rng, start, wide, int(rng.integers(200, 9000)))
        level, compat = int(rng.integers(2, 10)), int(rng.integers(0, 2))
        if (level, compat) not in comps:
            comps[(level, compat)] = _native.Compressor(level, compat, lib=emu_lib)
        assert comps[(level, compat)].deflate_compress(a) == oracle.deflate_compress(a, level, compat), (it, start, wide, level, compat)
    for c in comps.values():
        c.close()

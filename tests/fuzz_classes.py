"""Input classes whose FIRST bytes or exact COUNTS matter (round 5; the lesson of round 4's position-0 bug: the committed
suite's classes were the gap, not the hours of soak).  Generators only -- used by tests/test_emu_bucket0.py (emulator,
CPU) and tests/test_gpu_fuzz_slice.py (the MI355X):

  first_bytes(rng)        tools/emu_fuzz.case: buffers that start with a string hashing to bucket 0 of the hc hash4, hc
                          hash3 or level-1 table (libdeflate files position 0 under bucket 0 of every table), the start
                          recurring later with and without what followed it
  thresholds(rng)         tools/emu_fuzz.case_edges: copies at distances 32,765 ... 32,770 / 4,096 / 4,097 / 8,192 / 8,193
                          and of lengths around 3 / 4 / every nice_match_length / 258, some ending with the buffer
  full_sub_block_cuts(rng, level, oracle)
                          word salad over a tiny vocabulary: thousands of 3-5 byte matches, the buffer cut 0 ... 5 bytes
                          behind (and just in front of) the point where the oracle's parse fills a DEFLATE sub-block --
                          the 8,192nd match of level 1 (deflate_compress_fastest's sequence store), the 50,000th of
                          levels 2-9 (SEQ_STORE_LENGTH) -- so that the token behind the last match, the trailing
                          literals and the end of the buffer meet in every order
  unlike_segments(rng, n) / stale_path_slice(lib, oracle, rng, cases)
                          (round 5) blocks of unlike segments back to back -- text, DNA, noise, runs ... 300 bytes to 70 KB
                          each: should_end_block splits them and the split-off sub-block needs another min_len, which
                          behind a start that k_match_hc_sparse compacted sends the block through k_match_hc_stale
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))

_starts = None


def first_bytes(rng):
    import emu_fuzz
    global _starts
    if _starts is None:
        _starts = emu_fuzz.starts()
    return emu_fuzz.case(rng, *_starts)


def thresholds(rng):
    import emu_fuzz
    return emu_fuzz.case_edges(rng)


def many_short_matches(rng, level, scale=1.0):
    """Word salad with about one short match per word; `scale` x the size at which one match per word would fill the
    sub-block's sequence store (matches often swallow two words, so scale 2-3 is what really fills it)."""
    limit = 8192 if level <= 1 else 50000
    nv = int(rng.integers(12, 60))
    wl = int(rng.integers(2, 5))  # word length; with the separator a match is wl + 1 bytes
    alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz", np.uint8)
    vocab = alpha[rng.integers(0, 26, (nv, wl))]
    n = int(limit * (wl + 1) * scale) + int(rng.integers(0, 6))
    words = vocab[rng.integers(0, nv, n // (wl + 1) + 2)]
    a = np.concatenate([words, np.full((words.shape[0], 1), 32, np.uint8)], axis=1).reshape(-1)[:n].copy()
    # a few literals that are no match anywhere, so that sequences carry literal runs of different lengths
    for at in rng.integers(0, n, int(rng.integers(0, 40))):
        a[at] = rng.integers(48, 58)
    return np.ascontiguousarray(a)


def _full_sub_block_end(a, level, oracle):
    """Byte position behind the first DEFLATE sub-block of the oracle's parse of `a` that holds a FULL sequence store
    (8,192 matches at level 1, 50,000 at levels 2-9); None if no sub-block fills up."""
    if level <= 1:
        toks, first = oracle.l1_tokens(a)
        if len(first) < 2:
            return None
        t = toks[:int(first[1])].astype(np.int64)
        return int(np.where(t >> 31 != 0, t & 0x1FF, 1).sum())
    import deflate_tokens as dt  # (pure Python: about a second per buffer; only a handful of hc cases use it)
    out, blocks = dt.tokens(oracle.deflate_compress(a, level, 1))
    matches = [0] * len(blocks)
    for t in out:
        if t[1] == "M":
            matches[t[0]] += 1
    for i, m in enumerate(matches[:-1]):
        if m >= 50000:
            return blocks[i + 1][1]  # (type, byte position, first token) per DEFLATE block
    return None


def _two_letter_words(rng, n_words, nv=3000):
    """Words of two printable characters + a space from a vocabulary too large for word PAIRS to recur inside the window:
    nearly every match is exactly three bytes long, 100,000 sequences in 300,000 bytes -- the only way to fill a
    50,000-sequence store before SOFT_MAX_BLOCK_LENGTH (300,000 bytes) ends the sub-block first."""
    alpha = np.frombuffer(bytes(range(33, 127)), np.uint8)
    vocab = alpha[rng.integers(0, len(alpha), (nv, 2))]
    words = vocab[rng.integers(0, nv, n_words)]
    return np.concatenate([words, np.full((n_words, 1), 32, np.uint8)], axis=1).reshape(-1).copy()


def full_sub_block_cuts(rng, level, oracle, deltas=(-3, -1, 0, 1, 2, 3, 4, 5, 9, 300)):
    """Buffers cut right where a sub-block fills up: the oracle's parse says where the sub-block with a full sequence
    store ends (the byte behind its 8,192nd / 50,000th match); the cuts end the buffer 0 ... 5 bytes (and a few more)
    behind that point and just in front of it -- 'the last match of a full sub-block + k trailing bytes' for every
    small k, and the buffer that stops just short of filling it."""
    for _ in range(8):
        a = many_short_matches(rng, level, scale=float(rng.uniform(2.2, 3.2))) if level <= 1 else \
            _two_letter_words(rng, int(rng.integers(100000, 125000)), nv=int(rng.integers(2500, 4000)))
        p = _full_sub_block_end(a, level, oracle)
        if p is not None:
            return [np.ascontiguousarray(a[:p + d]) for d in deltas if 0 < p + d <= a.size]
    raise AssertionError("no buffer filled a sub-block at level %d" % level)


_SEGMENT_CLASSES = ["dna", "random", "text", "zeros", "lowent", "fastq", "ascii", "runs", "repeats", "period2"]


def unlike_segments(rng, n):
    from gzp_amd import synth
    parts, size = [], 0
    short = rng.random() < 0.4
    while size < n:
        ln = int(rng.integers(300, 9000)) if short else int(rng.integers(3000, 70000))
        parts.append(synth.make(_SEGMENT_CLASSES[rng.integers(len(_SEGMENT_CLASSES))], ln, int(rng.integers(1 << 30))))
        size += ln
    return np.ascontiguousarray(np.concatenate(parts)[:n])


def stale_path_slice(lib, oracle, rng, cases, block_sizes=(65536, 131072, 300001, 1 << 20), max_n=1_300_000):
    """`cases` random streams of unlike segments through `lib` (the emulated or the real library) at levels 2-4 -- the
    default route, and k_match_hc_sparse forced on every block (Config.debug bit 5; always at level 2) -- BGZF and Mgzip,
    both compat rules, the whole stream against the oracle.  Returns (blocks, blocks that went through k_match_hc_stale)."""
    from gzp_amd import _native
    blocks = stale = 0
    for it in range(cases):
        bgzf = rng.random() < 0.5
        bs = 65280 if bgzf else int(rng.choice(list(block_sizes)))
        n = int(rng.integers(bs // 2, min(3 * bs, max_n)))
        a = unlike_segments(rng, n)
        level = int(rng.choice([3, 3, 4, 2]))
        flags = 32 if level == 2 or rng.random() < 0.4 else 0
        compat = _native.COMPAT_1_24 if rng.random() < 0.7 else _native.COMPAT_1_10
        fmt, ofmt = (_native.FORMAT_BGZF, oracle.FMT_BGZF) if bgzf else (_native.FORMAT_MGZIP, oracle.FMT_MGZIP)
        with _native.Context(format=fmt, level=level, buffer_size=bs, compat=compat, lib=lib, max_slab_bytes=n) as c:
            c.debug_set_flags(flags)
            got = c.compress_slab(a, True)
            stale += c.debug_redo_count()
        assert got == oracle.compress_stream(a, ofmt, level, compat, bs), (it, level, flags, bs, n, compat)
        blocks += (n + bs - 1) // bs
    return blocks, stale

"""How deep are the dependency chains of a block's matches?  (round 6, before k_lzcopy was built.)  depth(match) = 1 + the
deepest match its source bytes come from, per tile; the number of polling levels k_lzcopy cannot go below.  Pure Python."""
import os
import sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from gzp_amd import synth
import zlib


def raw_deflate(data, level):
    """A raw DEFLATE stream of `data` (Python's zlib: the statistics are a DEFLATE stream's, whoever made it)."""
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    return c.compress(bytes(data)) + c.flush()
from sim_inflate_sync import *
LB = LBASE
DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]

def toks(raw):
    big = int.from_bytes(raw + b'\0' * 8, 'little'); pos = 0; out = []; o = 0
    while True:
        final, lt, dt, pos = parse_header(big, pos)
        while True:
            sym, l = lt[(big >> pos) & 32767]; pos += l
            if sym < 256: o += 1; continue
            if sym == 256: break
            ln = LB[sym - 257] + ((big >> pos) & ((1 << LEXT[sym - 257]) - 1)); pos += LEXT[sym - 257]
            ds, dl = dt[(big >> pos) & 32767]; pos += dl
            d = DBASE[ds] + ((big >> pos) & ((1 << DEXT[ds]) - 1)); pos += DEXT[ds]
            out.append((o, ln, d)); o += ln
        if final: break
    return out, o

def depth_stats(ms, n, T):
    # depth per byte; match depth = 1 + max depth of source bytes (within same tile; window bytes depth 0)
    res = []
    for ts in range(0, n, T):
        te = min(n, ts + T)
        dep = np.zeros(te - ts, dtype=np.int32)
        md = []
        for (p, ln, d) in ms:
            if p < ts or p >= te: continue
            s = p - d
            # needed existing range [s, min(s+ln, p))
            e = min(s + ln, p)
            lo = max(s, ts)
            m = 0
            if e > lo: m = int(dep[lo - ts:e - ts].max())
            m += 1
            dep[p - ts:min(p + ln, te) - ts] = m
            md.append(m)
        md = np.array(md)
        res.append((len(md), md.max(), md.mean(), np.bincount(md)[:12]))
    return res

if __name__ == '__main__':
    cases = [('text l1', synth.text_slab(65280 * 3, 65280 * 3), 1, 65280),
             ('text l3', synth.text_slab(65280 * 2, 65280 * 2), 3, 65280),
             ('dna l1', synth.dna(65280 * 2), 1, 65280),
             ('fastq l1', synth.fastq_like(65280 * 2), 1, 65280),
             ('lowent l1', synth.low_entropy_binary(65280 * 2), 1, 65280),
             ('repeated l1', synth.repeated_phrases(65280 * 2), 1, 65280),
             ('runs l1', synth.byte_runs(65280 * 2), 1, 65280),
             ('text l6 zlib', synth.text_slab(65280 * 2, 65280 * 2), -6, 65280),
             ]
    import zlib
    for name, data, level, bs in cases:
        data = bytes(data)
        for off in range(0, len(data), bs):
            blk = data[off:off + bs]
            if level < 0:
                c = zlib.compressobj(-level, zlib.DEFLATED, -15); raw = c.compress(blk) + c.flush()
            else:
                raw = raw_deflate(blk, level)
            try:
                ms, n = toks(raw)
            except AssertionError:
                print(name, 'skip'); continue
            lens = np.array([m[1] for m in ms]); dists = np.array([m[2] for m in ms])
            print(name, off, 'matches', len(ms), 'match bytes %.0f%%' % (100.0 * lens.sum() / n), 'len mean %.1f p90 %d p99 %d max %d' % (lens.mean(), np.percentile(lens, 90), np.percentile(lens, 99), lens.max()),
                  'dist<256 %.0f%% <4096 %.0f%%' % (100 * (dists < 256).mean(), 100 * (dists < 4096).mean()))
            for T in (32768, 65536):
                for (nm, mx, mean, hist) in depth_stats(ms, n, T):
                    print('   T=%d: matches %d depth max %d mean %.1f hist %s' % (T, nm, mx, mean, hist.tolist()))

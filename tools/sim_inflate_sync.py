"""How fast does a DEFLATE stream re-synchronise?  (round 6, before k_inflate_seg was built.)  For every block of a few
input classes: decode from segment starts that are NOT codeword boundaries and count the bits / symbols until the decode
meets the true chain; failures = it did not within the segment.  Also the lanes' imbalance (symbols per equal-bit segment).
Pure Python + the CPU-side compressor of the test infrastructure; no GPU."""
import os
import sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from gzp_amd import synth
import zlib


def raw_deflate(data, level):
    """A raw DEFLATE stream of `data` (Python's zlib: the statistics are a DEFLATE stream's, whoever made it)."""
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    return c.compress(bytes(data)) + c.flush()

LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEXT = [0] * 8 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [0]
DEXT = [0, 0, 0, 0] + [i // 2 for i in range(2, 28)] + [0, 0]
ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]

def build(lens):
    # 15-bit direct table: entry = (sym, len) ; LSB-first
    tab = [None] * 32768
    code = 0
    for ln in range(1, 16):
        for sym, l in enumerate(lens):
            if l == ln:
                rev = int(format(code, '0%db' % ln)[::-1], 2)
                for k in range(rev, 32768, 1 << ln):
                    tab[k] = (sym, ln)
                code += 1
        code <<= 1
    return tab

def parse_header(big, pos):
    final = (big >> pos) & 1; typ = (big >> (pos + 1)) & 3; pos += 3
    assert typ == 2, typ
    hlit = ((big >> pos) & 31) + 257; hdist = ((big >> (pos + 5)) & 31) + 1; hclen = ((big >> (pos + 10)) & 15) + 4
    pos += 14
    cl = [0] * 19
    for i in range(hclen):
        cl[ORDER[i]] = (big >> pos) & 7; pos += 3
    pt = build(cl)
    lens = []
    while len(lens) < hlit + hdist:
        sym, l = pt[(big >> pos) & 32767]; pos += l
        if sym < 16: lens.append(sym)
        elif sym == 16:
            r = 3 + ((big >> pos) & 3); pos += 2; lens += [lens[-1]] * r
        elif sym == 17:
            r = 3 + ((big >> pos) & 7); pos += 3; lens += [0] * r
        else:
            r = 11 + ((big >> pos) & 127); pos += 7; lens += [0] * r
    return final, build(lens[:hlit]), build(lens[hlit:]), pos

def step(big, pos, lt, dt):
    """decode one symbol at pos; returns (newpos, kind) kind: 0 lit 1 match 2 eob 3 invalid"""
    e = lt[(big >> pos) & 32767]
    if e is None: return pos + 1, 3
    sym, l = e
    pos += l
    if sym < 256: return pos, 0
    if sym == 256: return pos, 2
    if sym > 285: return pos, 3
    pos += LEXT[sym - 257]
    e = dt[(big >> pos) & 32767]
    if e is None: return pos + 1, 3
    ds, dl = e
    pos += dl + DEXT[ds]
    if ds > 29: return pos, 3
    return pos, 1

def analyse(raw, S_list, name):
    big = int.from_bytes(raw, 'little')
    nbits = len(raw) * 8
    pos = 0
    res = {S: dict(fail=0, n=0, syncbits=[], syncsyms=[], maxsyms=[], meansyms=[]) for S in S_list}
    nblocks = 0
    while True:
        final, lt, dt, pos = parse_header(big, pos)
        nblocks += 1
        start = pos
        true = set(); tl = []
        while True:
            true.add(pos); tl.append(pos)
            pos, k = step(big, pos, lt, dt)
            if k == 2: break
            assert k != 3
        end = pos
        tl_arr = np.array(tl)
        for S in S_list:
            r = res[S]
            nseg = (end - start + S - 1) // S
            # symbol counts per segment on true path
            cnt = np.bincount((tl_arr - start) // S, minlength=nseg)
            # per span of 64 lanes
            for sp in range(0, nseg, 64):
                c = cnt[sp:sp + 64]
                r['maxsyms'].append(c.max()); r['meansyms'].append(c.sum() / 64.0)
            for i in range(1, nseg):
                s = start + i * S
                p = s; n = 0
                lim = min(s + S, end)
                while p not in true and p < lim:
                    p, k = step(big, p, lt, dt); n += 1
                    if k == 2 or k == 3:
                        # restart? a lane that hits eob/invalid just stops: counts as fail unless..
                        p = lim + 1; break
                r['n'] += 1
                if p in true and p <= lim:
                    r['syncbits'].append(p - s); r['syncsyms'].append(n)
                else:
                    r['fail'] += 1
        if final: break
    print(name, 'deflate blocks', nblocks, 'comp bytes', len(raw))
    for S in S_list:
        r = res[S]
        sb = np.array(r['syncbits']) if r['syncbits'] else np.array([0])
        ss = np.array(r['syncsyms']) if r['syncsyms'] else np.array([0])
        print('  S=%5d segs=%6d fail=%5d (%.3f%%) sync bits mean %.0f p90 %.0f p99 %.0f max %d ; syms mean %.1f p99 %.0f ; lane imbalance max/mean %.2f' % (
            S, r['n'], r['fail'], 100.0 * r['fail'] / max(1, r['n']), sb.mean(), np.percentile(sb, 90), np.percentile(sb, 99), sb.max(),
            ss.mean(), np.percentile(ss, 99), np.sum(r['maxsyms']) / max(1e-9, np.sum(r['meansyms']))))

if __name__ == '__main__':
    S_list = [512, 1024, 2048, 4096]
    cases = [('text l1', synth.text_slab(65280 * 6, 65280 * 6), 1, 65280),
             ('text l3', synth.text_slab(65280 * 4, 65280 * 4), 3, 65280),
             ('ascii noise l3 256K', synth.ascii_random(262144), 3, 262144),
             ('dna l1', synth.dna(65280 * 3), 1, 65280),
             ('fastq l1', synth.fastq_like(65280 * 3), 1, 65280),
             ('lowent l1', synth.low_entropy_binary(65280 * 3), 1, 65280),
             ('mixed l1', synth.mixed(65280 * 3), 1, 65280),
             ]
    for name, data, level, bs in cases:
        data = bytes(data) if not isinstance(data, bytes) else data
        for off in range(0, len(data), bs):
            raw = raw_deflate(data[off:off + bs], level)
            raw = bytes(raw)
            try:
                analyse(raw + b'\0' * 8, S_list, '%s @%d' % (name, off))
            except AssertionError as e:
                print(name, off, 'skip (non-dynamic block)', e)

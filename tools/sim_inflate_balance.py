"""Lane balance of k_inflate_seg: symbols per equal-bit segment for 64 / 128 / 256 / 512 segments per block, the fullest
lane over the mean when a lane takes every 64th segment (static) or the next free one (dynamic).  Pure Python."""
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
import sim_inflate_sync as ss

def true_positions(raw):
    big = int.from_bytes(raw + b'\0' * 8, 'little'); pos = 0
    final, lt, dt, pos = ss.parse_header(big, pos)
    start = pos; tl = []
    while True:
        tl.append(pos)
        pos, k = ss.step(big, pos, lt, dt)
        if k == 2: break
    return start, pos, np.array(tl)

if __name__ == '__main__':
    data = bytes(synth.text_slab(65280 * 8, 65280 * 8))
    for off in range(0, len(data), 65280):
        raw = raw_deflate(data[off:off + 65280], 1)
        start, end, tl = true_positions(raw)
        out = []
        for nseg in (64, 128, 256, 512):
            S = (((end - start + nseg - 1) // nseg) + 31) & ~31
            cnt = np.bincount((tl - start) // S, minlength=nseg)[:nseg]
            mean_lane = cnt.sum() / 64.0
            static = cnt.reshape(-1, 64).sum(axis=0).max() if nseg % 64 == 0 else 0   # lane i gets i, i+64, ...
            # dynamic: greedy list scheduling in order
            lanes = np.zeros(64)
            for c in cnt:
                j = lanes.argmin(); lanes[j] += c
            out.append('nseg %d: max/mean static %.2f dynamic %.2f' % (nseg, static / mean_lane, lanes.max() / mean_lane))
        print(off, len(tl), ' | '.join(out))

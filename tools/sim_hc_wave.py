"""What does a WAVE pay for hc_matchfinder_longest_match?  A CPU model of k_match_hc's search on one 65,280-byte block
of the bench text (no GPU): every position's chain walk is traced (nodes looked at, hits, extension lengths), then the
positions are put together the way the kernel does it -- 64 consecutive positions per wave, the wave in lockstep -- and
the things a wave executes are counted: rounds (= the longest walk of its 64 lanes), rounds in which SOME lane hits
(the whole wave then runs the hit path), iterations of the 16-byte extension loop.  Round 4 used it to decide what NOT
to build:

  * on text a lane looks at 5.7 nodes at level 3 and hits 1.07 times, but its wave runs 11.9 rounds and the hit path in
    6.8 of them -- the lanes use 48 % of the rounds (38 % at level 6, 21 % at level 9);
  * "lane sequences" -- every lane searching its sixteen positions of a tile at its own pace -- would cut the rounds to
    8.2 / 20.4 / 130 (levels 3 / 6 / 9), which is what the refill kernel was built for and lost on the GPU (the start-up
    code then runs with 24 lanes instead of 64);
  * only 26 % of the positions start a token (greedy, min_len 4), so a match-on-demand walk needs a third of the
    searches -- but at the price of one search per wave STEP, whatever the number of lanes in it.

    python tools/sim_hc_wave.py            (levels 3, 6, 9: a few seconds of pure Python)
    python tools/sim_hc_wave.py --parked   (also the model of parked-and-compacted walks)
"""
import collections
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gzp_amd import synth

N = 65280
data = bytes(synth.text_slab(N * 3, seed=20250927)[N:2 * N])
if os.environ.get("SIM_HC_DATA"):  # another input for the model: "<class>:<seed>" of gzp_amd.synth, e.g. text:6
    _cls, _seed = os.environ["SIM_HC_DATA"].split(":")
    data = bytes(synth.make(_cls, N * 3, int(_seed))[N:2 * N])


def le32(i):
    return int.from_bytes(data[i:i + 4].ljust(4, b"\0"), "little")


# hc_matchfinder's tables as k_candidates leaves them: distance to the previous position with the same hash
h4, h3 = {}, {}
d4, d3 = [0] * N, [0] * N
for p in range(N - 3):
    v = le32(p)
    k4 = ((v * 0x1E35A7BD) & 0xFFFFFFFF) >> 16
    k3 = (((v & 0xFFFFFF) * 0x1E35A7BD) & 0xFFFFFFFF) >> 17
    q = h4.get(k4)
    d4[p] = p - q if q is not None and p - q <= 32767 else 0
    h4[k4] = p
    q = h3.get(k3)
    d3[p] = p - q if q is not None and p - q <= 32767 else 0
    h3[k3] = p


def ext(a, c, start, maxl):
    n = start
    while n < maxl and data[a + n] == data[c + n]:
        n += 1
    return n


def trace(depth0, nice):
    """Per position: the list of rounds of its walk, ('m',) or ('h', extension bytes), and the match length."""
    traces, lens = [], []
    for p in range(N):
        rem = N - p
        tr, best = [], 2
        if rem >= 5 and d3[p]:
            maxl = min(258, rem)
            nl = min(nice, maxl)
            if data[p - d3[p]:p - d3[p] + 3] == data[p:p + 3]:
                best = 3
            tot = d4[p] if d4[p] else 0x8000
            off = used = 0
            while tot <= 32767 and used < depth0:
                c = p - tot
                nxt = d4[c] if d4[c] else 0x8000
                if data[c + off:c + off + 4] == data[p + off:p + off + 4] and (off == 0 or data[c:c + 4] == data[p:p + 4]):
                    start = 4 if off == 0 or best > 7 else best + 1
                    n = ext(p, c, start, maxl)
                    tr.append(("h", n - start))
                    if n > best:
                        best, off = n, n - 3
                        if n >= nl:
                            tot = 0x10000
                else:
                    tr.append(("m",))
                tot += nxt
                used += 1
        traces.append(tr)
        lens.append(best)
    return traces, lens


for level, depth0, nice in ((3, 12, 14), (6, 35, 65), (9, 600, 258)):
    traces, lens = trace(depth0, nice)
    nodes = np.array([len(t) for t in traces])
    hits = np.array([sum(1 for e in t if e[0] == "h") for t in traces])
    # the greedy parse (min_len 4 on this text): how many positions start a token
    pos = ntok = 0
    while pos < N:
        pos += lens[pos] if lens[pos] >= 4 else 1
        ntok += 1
    rounds = hit_rounds = ext16 = nw = 0
    for w0 in range(0, N, 64):
        tw = traces[w0:w0 + 64]
        r_w = max(len(t) for t in tw)
        rounds += r_w
        nw += 1
        for r in range(r_w):
            hs = [t[r] for t in tw if len(t) > r and t[r][0] == "h"]
            if hs:
                hit_rounds += 1
                ext16 += max([0] + [1 + (e[1] - 8) // 16 for e in hs if e[1] >= 8])  # (the first round compares 8 bytes)
    # lanes that search their sixteen positions of a 16 KiB tile at their own pace (ideal refill)
    seq = 0
    for t0 in range(0, N - 16384 + 1, 16384):
        for w in range(16):
            m = np.array([[nodes[t0 + 1024 * k + 64 * w + l] for l in range(64)] for k in range(16)])
            seq += m.sum(axis=0).max()
    print("level %d (depth %d, nice %d): %.1f nodes and %.2f hits per position; per wave and 64 positions: %.1f rounds "
          "(lanes used %.0f %%), the hit path in %.1f of them, %.1f sixteen-byte extension rounds; lane sequences: %.1f rounds; "
          "%.0f %% of the positions start a token"
          % (level, depth0, nice, nodes.mean(), hits.mean(), rounds / nw, 100.0 * nodes.sum() / (rounds * 64), hit_rounds / nw,
             ext16 / nw, seq / (3 * 16 * 16), 100.0 * ntok / N), flush=True)


def parked(nodes, q0, q, pop_cost=2.0, push_cost=1.0):
    """Walks parked and compacted (not built): every position gets q0 nodes of its walk, the searches that are not
    over are parked in a per-wave list and continued q nodes at a time, 64 to a wave; a pop is charged two rounds, a
    push one.  Returns (rounds of the plain lockstep wave, rounds of this scheme)."""
    base = new = 0.0
    for t0 in range(0, N - 16384 + 1, 16384):
        for w in range(16):
            lst = []
            for k in range(16):
                g = [int(nodes[t0 + 1024 * k + 64 * w + l]) for l in range(64)]
                base += max(g)
                new += min(max(g), q0) + push_cost
                lst += [x - q0 for x in g if x > q0]
                while len(lst) >= 64:
                    cur, lst = lst[:64], lst[64:]
                    new += pop_cost + min(max(cur), q) + push_cost
                    lst += [x - q for x in cur if x > q]
            while lst:
                cur, lst = lst[:64], lst[64:]
                new += pop_cost + min(max(cur), q) + push_cost
                lst += [x - q for x in cur if x > q]
    return base, new


if "--parked" in sys.argv:
    for level, depth0, nice in ((3, 12, 14), (6, 35, 65), (9, 600, 258)):
        nodes = np.array([len(t) for t in trace(depth0, nice)[0]])
        for q0, q in ((4, 8), (8, 16), (16, 32)):
            if q0 < depth0:
                b, n_ = parked(nodes, q0, q)
                print("level %d, first stretch %d nodes, then %d at a time: %.2f x the wave-rounds (model; a full round costs about "
                      "twice a sparse one on the GPU)" % (level, q0, q, n_ / b), flush=True)
first = collections.Counter(next((i for i, e in enumerate(t) if e[0] == "h"), -1) for t in trace(12, 14)[0])
print("level 3: the first hit of a walk is at node", sorted(first.items())[:6], "(-1: the walk has none)")

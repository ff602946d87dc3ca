"""CPU model of the COMPACTION form of k_match_hc for the greedy levels (round 5): a dense pass that stops
behind every position's FIRST chain node, the greedy walk over those lengths, the rest of the chain walk only for the
predicted token starts whose search is not over (compacted, one per lane), the walk again, ... until the path holds only
finished searches.  Counts, for one 65,280-byte block of the bench text in 16 Ki-position tiles: rounds until the path
settles, searches per round, and the wave-rounds of the compacted walks (64 list entries per wave, the wave in lockstep)
next to the dense kernel's.      python tools/sim_hc_compact.py [level]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sim_hc_wave as W  # noqa: E402  (its tables and trace(); prints its own summary on import)

N, data, d3, d4 = W.N, W.data, W.d3, W.d4
TILE = int(os.environ.get("SIM_HC_TILE", "13056"))  # k_match_hc_sparse's tile


def first_node(depth0, nice):
    """Per position: length / after the hash3 check and ONE chain node, whether the search is over by then, and the nodes left."""
    traces, lens = W.trace(depth0, nice)
    t1, l1 = W.trace(1, nice)
    unfinished = np.array([len(t) > 1 for t in traces])  # the full walk looks at more than one node
    rest = np.array([max(0, len(t) - 1) for t in traces])
    return np.array(l1), np.array(lens), unfinished, rest, traces


def greedy(L, start, end, min_len=4):
    pos, starts = start, []
    while pos < end:
        starts.append(pos)
        pos += L[pos] if L[pos] >= min_len else 1
    return starts, pos


def main():
    level = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    depth0, nice = {2: (6, 10), 3: (12, 14), 4: (16, 30)}[level]
    l1, lf, unfinished, rest, traces = first_node(depth0, nice)
    print("level %d: first-node length final for %.1f %% of the positions; search over after the first node for %.1f %%"
          % (level, 100.0 * (l1 == lf).mean(), 100.0 * (~unfinished).mean()))
    entry, tot_rounds, tot_search, tot_wrounds, tot_tok = 0, 0, 0, 0, 0
    dense_wrounds = sum(max(len(t) for t in traces[w:w + 64]) for w in range(0, N, 64))
    for t0 in range(0, N, TILE):
        t1 = min(N, t0 + TILE)
        L = l1.copy()
        done = ~unfinished
        rounds = 0
        while True:
            starts, exit_pos = greedy(L, entry, t1)
            need = [p for p in starts if not done[p]]
            if not need:
                break
            rounds += 1
            tot_search += len(need)
            for w in range(0, len(need), 64):  # one list entry per lane, lockstep: the longest remaining walk
                tot_wrounds += max(rest[p] for p in need[w:w + 64])
            for p in need:
                L[p] = lf[p]
                done[p] = True
            print("  tile %5d round %d: %5d token starts, %4d unfinished searches" % (t0, rounds, len(starts), len(need)))
        tot_rounds += rounds
        tot_tok += len(starts)
        entry = exit_pos
    print("level %d: %d tiles, %d rounds in all, %d compacted searches for %d tokens (%.1f %% of the positions); "
          "wave-rounds of the compacted walks %d vs %d dense (beyond the first node: %d)"
          % (level, (N + TILE - 1) // TILE, tot_rounds, tot_search, tot_tok, 100.0 * tot_search / N, tot_wrounds, dense_wrounds,
             dense_wrounds - (N + 63) // 64))


if __name__ == "__main__":
    main()

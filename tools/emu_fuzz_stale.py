"""CPU-only hunt through the emulated kernels for the k_match_hc_stale path (round 5): blocks of unlike segments back to
back, so that a split-off sub-block needs another min_len behind a compacted start -- levels 3-4 by the default route and
level 2-4 with the sparse kernel forced on every block (Config.debug bit 5), BGZF and Mgzip blocks of 64 KiB ... 1 MiB,
the whole stream against the oracle; tallies how many blocks went the stale way.
usage: emu_fuzz_stale.py [seconds] [seed] [--gpu]      (--gpu: the same hunt through the real library on an MI355X)"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import build_emu
from gzp_amd import _native, synth
from oracle import oracle

import fuzz_classes  # noqa: E402  (tests/: the committed generator)

segments = fuzz_classes.unlike_segments


def main():
    argv = [x for x in sys.argv[1:] if x != "--gpu"]
    secs = float(argv[0]) if len(argv) > 0 else 60.0
    seed = int(argv[1]) if len(argv) > 1 else 1
    lib = _native.load() if "--gpu" in sys.argv else _native.GzpxLib(build_emu.build())
    rng = np.random.default_rng(seed)
    t0, cases, stale_blocks, stale_cases, blocks = time.time(), 0, 0, 0, 0
    while time.time() - t0 < secs:
        bgzf = rng.random() < 0.5
        bs = 65280 if bgzf else int(rng.choice([65536, 131072, 300001, 1 << 20]))
        n = int(rng.integers(bs // 2, min(3 * bs, 1_300_000)))
        a = segments(rng, n)
        level = int(rng.choice([3, 3, 4, 2]))
        flags = 32 if level == 2 or rng.random() < 0.4 else 0
        compat = _native.COMPAT_1_24 if rng.random() < 0.7 else _native.COMPAT_1_10
        fmt, ofmt = (_native.FORMAT_BGZF, oracle.FMT_BGZF) if bgzf else (_native.FORMAT_MGZIP, oracle.FMT_MGZIP)
        want = oracle.compress_stream(a, ofmt, level, compat, bs)
        with _native.Context(format=fmt, level=level, buffer_size=bs, compat=compat, lib=lib, max_slab_bytes=n) as c:
            c.debug_set_flags(flags)
            got = c.compress_slab(a, True)
            st = c.debug_redo_count()
        if got != want:
            np.save("/tmp/emu_fuzz_stale_fail_%d_%d.npy" % (seed, cases), a)
            print("MISMATCH seed %d case %d: level %d flags %d bs %d n %d compat %d" % (seed, cases, level, flags, bs, n, compat), flush=True)
            sys.exit(1)
        cases += 1
        blocks += (n + bs - 1) // bs
        stale_blocks += st
        stale_cases += st > 0
    print("seed %d: %d cases, %d blocks, %d of them by k_match_hc_stale (in %d cases): all equal to the oracle"
          % (seed, cases, blocks, stale_blocks, stale_cases), flush=True)


if __name__ == "__main__":
    main()

"""Why is `mixed` (random || text || random) slower by the sparse route of levels 3-4 than its parts say?  Stage times
(HIP events around every launch group) of level 3 on compositions of the same two ingredients, dense route (Config.debug
bit 4) next to the default one -- and, given more libraries, the default route of each (A/B of builds):
python tools/exp_mixed_hc.py [level] [lib ...]

What it showed (round 5): with whole blocks of either kind the sparse route is the faster one in every arrangement; the
loss comes from the one or two blocks that hold BOTH (a sub-block with another min_len behind a compacted start: the
block goes stale), whose dense search was one workgroup's work with every other CU idle -- k_match_hc_stale now deals
such a block out in pieces."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from gzp_amd import _native, synth

lib = _native.load()
more = [(os.path.basename(p), _native.GzpxLib(p)) for p in sys.argv[2:]]
n = 256 << 20
level = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = 65280
text8 = synth.make("text", 8 << 20, 5)
rand8 = synth.make("random", 8 << 20, 5)


def tiled(base):
    return np.tile(base, n // base.size)


def by_blocks(pattern):
    """whole 65,280-byte blocks of text ('t') / random ('r') following `pattern`, repeated: no block holds both"""
    out = np.empty(n, dtype=np.uint8)
    tb, rb = tiled(text8), tiled(rand8)
    for i, lo in enumerate(range(0, n, B)):
        src = tb if pattern[i % len(pattern)] == "t" else rb
        out[lo:lo + B] = src[lo:lo + B][:min(B, n - lo)]
    return out


cases = [
    ("text", tiled(text8)),
    ("random", tiled(rand8)),
    ("mixed 8 MiB tiled (exp_classes_hc)", tiled(synth.make("mixed", 8 << 20, 5))),
    ("mixed, thirds of the slab", synth.make("mixed", n, 5)),
    ("blocks r r t", by_blocks("rrt")),
    ("blocks 43r 43t 43r", by_blocks("r" * 43 + "t" * 43 + "r" * 43)),
    ("blocks: 2/3 random then 1/3 text", by_blocks("r" * 2742 + "t" * 1371)),
]
for name, a in cases:
    d_in = torch.from_numpy(a).cuda()
    line = {}
    for route, flags, use in [("dense", 16, lib), ("sparse", 0, lib)] + [(nm, 0, l) for nm, l in more]:
        with _native.Context(format=0, level=level, buffer_size=B, lib=use, max_slab_bytes=n) as ctx:
            ctx.debug_set_flags(flags)
            cap = ctx.slab_bound(n)
            d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
            ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
            ctx.set_profiling(True)
            acc = {}
            for _ in range(3):
                ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
                for k, v in ctx.last_stage_ms().items():
                    acc[k] = acc.get(k, 0.0) + v / 3
            line[route] = acc
            del d_out
    keys = [k for k in line["dense"] if line["dense"][k] > 0.02]
    print("%-36s level %d" % (name, level))
    for k in keys:
        d, s = line["dense"][k], line["sparse"].get(k, 0.0)
        print("    %-26s dense %7.3f  sparse %7.3f  (%+5.1f %%)" % (k, d, s, 100.0 * (s / d - 1.0)) +
              "".join("  %s %7.3f" % (nm, line[nm].get(k, 0.0)) for nm, _ in more), flush=True)
    del d_in

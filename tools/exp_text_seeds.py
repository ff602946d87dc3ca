"""Is the gain of the sparse route at levels 3-4 a property of 'text' or of one text?  english_like with several seeds
(8 MiB tiled to 256 MiB) and the bench's own slab text, match + parse stage by both routes:
python tools/exp_text_seeds.py [level] [lib ...]      (more libraries: the default route of each, an A/B of builds)

What it showed (round 5): the gain is a property of the text -- english_like seeds 5 ... 9 at level 3: -3.7, +6.3, +1.2,
-2.6, +4.7 % against the dense kernel, the bench slab -4.6 % -- with the ring of 1,280 entries that every tile's first
list (1,400-1,750 on these texts) overflowed."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from gzp_amd import _native, synth

lib = _native.load()
more = [(os.path.basename(p), _native.GzpxLib(p)) for p in sys.argv[2:]]
n, B = 256 << 20, 65280
level = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cases = [("english_like seed %d" % s, np.tile(synth.english_like(8 << 20, s), 32)) for s in (5, 6, 7, 8, 9)]
cases.append(("english_like 85 MiB seed 6 (mixed's)", np.resize(synth.english_like(n // 3, 6), n)))
cases.append(("bench text_slab", synth.text_slab(n, seed=20250927)))
for name, a in cases:
    d_in = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    t = {}
    for route, flags, use in [("dense", 16, lib), ("sparse", 0, lib)] + [(nm, 0, l) for nm, l in more]:
        with _native.Context(format=0, level=level, buffer_size=B, lib=use, max_slab_bytes=n) as ctx:
            ctx.debug_set_flags(flags)
            cap = ctx.slab_bound(n)
            d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
            ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
            ctx.set_profiling(True)
            acc = 0.0
            for _ in range(3):
                ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
                acc += ctx.last_stage_ms()["k_match_hc+k_parse_hc"] / 3
            t[route] = acc
            del d_out
    print("%-40s level %d  match+parse dense %7.3f  sparse %7.3f  (%+5.1f %%)" % (name, level, t["dense"], t["sparse"], 100.0 * (t["sparse"] / t["dense"] - 1.0)) +
          "".join("  %s %7.3f (%+5.1f %%)" % (nm, t[nm], 100.0 * (t[nm] / t["dense"] - 1.0)) for nm, _ in more), flush=True)
    del d_in

"""One composition of tools/exp_mixed_hc.py by one route, for a kernel trace (rocprofv3 --kernel-trace --stats):
python tools/exp_mixed_trace.py <thirds|tiled|blocks> <debug flags> [level]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from gzp_amd import _native, synth

what, flags = sys.argv[1], int(sys.argv[2])
level = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n, B = 256 << 20, 65280
if what == "thirds":
    a = synth.make("mixed", n, 5)
elif what == "tiled":
    a = np.tile(synth.make("mixed", 8 << 20, 5), n // (8 << 20))
else:
    t, r = np.tile(synth.make("text", 8 << 20, 5), 32), np.tile(synth.make("random", 8 << 20, 5), 32)
    a = np.empty(n, dtype=np.uint8)
    for i, lo in enumerate(range(0, n, B)):
        a[lo:lo + B] = (t if i >= 2742 else r)[lo:lo + B]
lib = _native.load()
d_in = torch.from_numpy(a).cuda()
with _native.Context(format=0, level=level, buffer_size=B, lib=lib, max_slab_bytes=n) as ctx:
    ctx.debug_set_flags(flags)
    cap = ctx.slab_bound(n)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    for _ in range(4):
        ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
    torch.cuda.synchronize()
    print(what, flags, "stale blocks:", ctx.debug_redo_count())

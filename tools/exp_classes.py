"""Level-1 BGZF rate of every synthetic input class (256 MiB each, device-resident) and how many blocks the
match-on-demand kernel hands back to the dense pair."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from gzp_amd import _native, synth

lib = _native.GzpxLib(sys.argv[1]) if len(sys.argv) > 1 else _native.load()
n = 256 << 20
for cls in sorted(synth.CLASSES):
    base = synth.make(cls, 8 << 20, 5)
    a = np.tile(base, n // base.size)
    d_in = torch.from_numpy(a).cuda()
    with _native.Context(format=0, level=1, buffer_size=65280, lib=lib, max_slab_bytes=n) as ctx:
        cap = ctx.slab_bound(n)
        d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        try:
            ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
        except _native.GzpxError as e:
            print("%-10s %s" % (cls, e))
            continue
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out_len, _ = ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print("%-10s %7.1f GiB/s  ratio %.3f  handed back %d of %d blocks" % (cls, n / 2**30 / dt, out_len / n, ctx.debug_redo_count(),
                                                                         ctx.n_blocks(n)), flush=True)
    del d_in, d_out

"""Levels 3 and 4 on every synthetic input class (256 MiB each, device-resident, BGZF): the default route
(k_match_hc_sparse, which searches noise blocks the dense way itself) next to the dense k_match_hc for every block
(Config.debug bit 4) -- is the compaction a gain on inputs other than text?   python tools/exp_classes_hc.py [levels]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from gzp_amd import _native, synth

lib = _native.load()
n = 256 << 20
levels = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [3, 4]
for cls in sorted(synth.CLASSES):
    base = synth.make(cls, 8 << 20, 5)
    a = np.tile(base, n // base.size)
    d_in = torch.from_numpy(a).cuda()
    for level in levels:
        res = {}
        for route, flags in (("dense", 16), ("sparse", 0)):
            with _native.Context(format=0, level=level, buffer_size=65280, lib=lib, max_slab_bytes=n) as ctx:
                ctx.debug_set_flags(flags)
                cap = ctx.slab_bound(n)
                d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
                try:
                    ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
                except _native.GzpxError as e:
                    res[route] = (None, str(e))
                    continue
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    out_len, _ = ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 3
                res[route] = (dt * 1e3, hashlib.sha256(d_out[:out_len].cpu().numpy()).hexdigest()[:12], out_len / n)
            del d_out
        if res["dense"][0] is None or res["sparse"][0] is None:
            print("%-8s level %d  %s" % (cls, level, res), flush=True)
            continue
        print("%-8s level %d  dense %7.2f ms  sparse %7.2f ms  (%+5.1f %%)  ratio %.3f  streams equal: %s"
              % (cls, level, res["dense"][0], res["sparse"][0], 100.0 * (res["sparse"][0] / res["dense"][0] - 1.0), res["sparse"][2],
                 res["dense"][1] == res["sparse"][1]), flush=True)
    del d_in

"""Which resource bounds k_match / k_parse?  Builds a second library with -DGZPX_EXPERIMENT (never the
product build) in which debug bits switch parts of the kernels off -- results are wrong on purpose --
and prints the stage times of the bench slab per variant.
   bit 4: k_match without the cand[p - d0] gather     bit 5: k_match without its global stores
   bit 6: k_match without the extension loop           bit 8: k_parse without token stores
   bit 9: k_parse without val loads"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from gzp_amd import _native, build, synth

exp = os.path.join(build.LIB_DIR, "libgzpx_exp.so")
srcs = [os.path.join(build.CSRC, s) for s in build.SOURCES]
subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
                       "-DGZPX_EXPERIMENT", "-I", build.INCLUDE] + srcs + ["-o", exp])
lib = _native.GzpxLib(exp)
n = 576_716_800
slab = synth.text_slab(n, seed=20250927)
d_in = torch.from_numpy(slab).cuda()
ctx = _native.Context(format=0, level=1, buffer_size=65280, lib=lib, max_slab_bytes=n)
cap = ctx.slab_bound(n)
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
ctx.set_profiling(True)
for name, flags in [("baseline", 0), ("match: no gather", 1 << 4), ("match: no stores", 1 << 5),
                    ("match: no extension loop", 1 << 6), ("match: no gather, no stores", (1 << 4) | (1 << 5)),
                    ("match: none of the three", (1 << 4) | (1 << 5) | (1 << 6)),
                    ("parse: no token stores", 1 << 8), ("parse: no val loads", 1 << 9),
                    ("parse: neither", (1 << 8) | (1 << 9))]:
    ctx.debug_set_flags(flags)
    acc = {}
    for it in range(6):
        try:
            ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
        except _native.GzpxError:
            pass  # wrong tokens can overflow a BGZF block: timing is still valid
        if it:
            for k, v in ctx.last_stage_ms().items():
                acc[k] = acc.get(k, 0.0) + v / 5
    print("%-28s match %.3f  parse %.3f  cand %.3f  total %.3f" % (name, acc["k_match"], acc["k_parse"], acc["k_candidates"],
                                                                   sum(acc.values())), flush=True)

"""What bounds k_mparse?  Second library with -DGZPX_EXPERIMENT (never the product build): cycles per
phase (thread 0's clock, summed over blocks) and timing with parts switched off -- results wrong on purpose.
   bit 13: walks without the cand[p - d0] gather      bit 14: without the token build"""
import ctypes
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
nb = ctx.n_blocks(n)
cyc = (ctypes.c_ulonglong * 8)()
for name, flags in [("baseline", 0), ("no gather", 1 << 13), ("no token build", 1 << 14),
                    ("neither", (1 << 13) | (1 << 14))]:
    ctx.debug_set_flags(flags)
    acc = {}
    for it in range(6):
        if it == 1:
            lib.L.gzpx_exp_cycles(cyc, 1)
        try:
            ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
        except _native.GzpxError:
            pass  # wrong tokens can overflow a BGZF block: timing is still valid
        if it:
            for k, v in ctx.last_stage_ms().items():
                acc[k] = acc.get(k, 0.0) + v / 5
    lib.L.gzpx_exp_cycles(cyc, 1)
    c = [x / 5 / nb for x in cyc]
    print("%-16s k_mparse %.3f ms | per block: stage %.0f  first walk %.0f  later rounds %.0f  settle %.0f  scan+build %.0f cycles; "
          "%.2f barrier rounds, %.1f re-walks" % (name, acc["k_mparse"], c[0], c[1], c[2], c[3], c[4], c[5], c[6]), flush=True)

"""What bounds k_match_hc?  -DGZPX_EXPERIMENT build (never the product), level 3 and 6 on 512 MiB of text:
   bit 10: no lz_extend (hits count, their extension loop does not run)   bit 11: chain walk only, no hits
   bit 12: depth 1 (tile loads and stores only)
   k_parse_hc (level 3 only): bit 13 no min_len filter, bit 14 no observation-class tally, bit 15 no token stores.
   Results are wrong on purpose; times are what matters."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from gzp_amd import _native, build, synth

exp = os.path.join(build.LIB_DIR, "libgzpx_exp.so")
srcs = [os.path.join(build.CSRC, s) for s in build.SOURCES]
subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
                       "-DGZPX_EXPERIMENT", "-I", build.INCLUDE] + srcs + ["-o", exp])
lib = _native.GzpxLib(exp)
n = 512 << 20
slab = synth.text_slab(n, seed=20250927)
d_in = torch.from_numpy(slab).cuda()
for level in (3, 6):
    ctx = _native.Context(format=0, level=level, buffer_size=65280, lib=lib, max_slab_bytes=n)
    cap = ctx.slab_bound(n)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    ctx.set_profiling(True)
    for name, flags in [("baseline", 0), ("no extension", 1 << 10), ("chain walk only", 1 << 11), ("depth 1", 1 << 12),
                        ("parse: no min_len filter", 1 << 13), ("parse: no class tally", 1 << 14),
                        ("parse: no token stores", 1 << 15), ("parse: none of the three", 7 << 13)]:
        if level != 3 and flags >= (1 << 13):
            continue
        ctx.debug_set_flags(flags)
        acc = {}
        for it in range(3):
            try:
                ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
            except _native.GzpxError:
                pass
            if it:
                for k, v in ctx.last_stage_ms().items():
                    acc[k] = acc.get(k, 0.0) + v / 2
        print("level %d %-18s match+parse %.2f ms" % (level, name, acc["k_match_hc+k_parse_lazy" if level >= 5 else "k_match_hc+k_parse_hc"]), flush=True)
    ctx.close()

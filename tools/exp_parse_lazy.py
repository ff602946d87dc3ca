"""Where does k_parse_lazy spend its time?  -DGZPX_EXPERIMENT build (never the product): the wave's clock per phase of a
window, summed over a block's windows and averaged over the blocks, on 550 MiB of text (BGZF) at levels 6 and 9."""
import ctypes
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from gzp_amd import _native, build, synth

exp = os.path.join(build.LIB_DIR, "libgzpx_exp.so")
srcs = [os.path.join(build.CSRC, s) for s in build.SOURCES]
subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
                       "-DGZPX_EXPERIMENT", "-I", build.INCLUDE] + srcs + sys.argv[1:] + ["-o", exp], stderr=subprocess.DEVNULL)
lib = _native.GzpxLib(exp)
cyc = (ctypes.c_ulonglong * 8)()
names = ["tile loads", "decisions", "chase", "prefixes + due", "commit", "what was due"]
n, bs = 576_716_800, 65280
d_in = torch.from_numpy(synth.text_slab(n, seed=20250927)).cuda()
for level in (6, 9):
    ctx = _native.Context(format=0, level=level, buffer_size=bs, lib=lib, max_slab_bytes=n)
    cap = ctx.slab_bound(n)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    ctx.set_profiling(True)
    nb = ctx.n_blocks(n)
    acc = {}
    for it in range(3):
        if it == 1:
            lib.L.gzpx_exp_cycles(cyc, 1)
        ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
        if it:
            for k, v in ctx.last_stage_ms().items():
                acc[k] = acc.get(k, 0.0) + v / 2
    lib.L.gzpx_exp_cycles(cyc, 1)
    c = [x / 2 / nb for x in cyc]
    tot = sum(c[:6])
    print("level %d: match+parse %.2f ms; k_parse_lazy clock ticks per block %.0f, %.0f windows, %.0f tile loads: " % (level, acc["k_match_hc+k_parse_lazy"], tot, c[6], c[7]) +
          ", ".join("%s %.0f (%.0f%%, %.0f per window)" % (names[k], c[k], 100 * c[k] / tot, c[k] / c[6]) for k in range(6)), flush=True)
    ctx.close()
    del d_out

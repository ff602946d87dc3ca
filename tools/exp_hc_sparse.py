"""What would a k_match_hc wave cost if its lanes searched only where a token starts?  -DGZPX_EXPERIMENT build (never the
product): the kernel searches a pseudo-random share of the positions (bits 16-22 of the debug word, in 128ths) and reports
"no match" for the others, so a wave's sixteen lockstep iterations run with that share of its lanes.  Level 3, 512 MiB of
text; the stream stays valid but is not libdeflate's -- the time of k_match_hc + k_parse_hc is what matters (the parse
sees more literals, so its part grows a little)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from gzp_amd import _native, build, synth

exp = os.path.join(build.LIB_DIR, "libgzpx_exp.so")
srcs = [os.path.join(build.CSRC, s) for s in build.SOURCES]
subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
                       "-DGZPX_EXPERIMENT", "-I", build.INCLUDE] + srcs + ["-o", exp], stderr=subprocess.DEVNULL)
lib = _native.GzpxLib(exp)
n = 512 << 20
d_in = torch.from_numpy(synth.text_slab(n, seed=20250927)).cuda()
ctx = _native.Context(format=0, level=3, buffer_size=65280, lib=lib, max_slab_bytes=n)
cap = ctx.slab_bound(n)
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
ctx.set_profiling(True)
for share in (0, 127, 96, 64, 48, 33, 16, 8):
    ctx.debug_set_flags(share << 16)
    acc = {}
    for it in range(3):
        try:
            ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
        except _native.GzpxError:
            pass
        if it:
            for k, v in ctx.last_stage_ms().items():
                acc[k] = acc.get(k, 0.0) + v / 2
    print("searched share %3d/128 (%s): match + parse %.2f ms" % (share, "all: the product's path" if share == 0 else "%.0f %%" % (share / 1.28),
                                                                    acc["k_match_hc+k_parse_hc"]), flush=True)
ctx.close()

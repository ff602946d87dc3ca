"""Where does k_parse_hc spend its time?  -DGZPX_EXPERIMENT build (never the product): thread 0's clock per phase,
summed over the blocks of the launches, on 550 MiB of text (BGZF, level 3) and 1 GiB of configs[2]'s ASCII noise
(Mgzip, 1 MiB blocks)."""
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
                       "-DGZPX_EXPERIMENT", "-I", build.INCLUDE] + srcs + ["-o", exp], stderr=subprocess.DEVNULL)
lib = _native.GzpxLib(exp)
cyc = (ctypes.c_ulonglong * 8)()
names = ["stage", "min_len filter", "first walk", "walk rounds", "scan", "boundaries+checks", "token pass", "(rounds)"]
for what in ("text", "ascii"):
    if what == "text":
        n, fmt, bs = 576_716_800, 0, 65280
        d_in = torch.from_numpy(synth.text_slab(n, seed=20250927)).cuda()
    else:
        n, fmt, bs = 1 << 30, 1, 1 << 20
        d_in = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
        _native.synth_ascii_device(d_in.data_ptr(), 0, n, 8, lib=lib)
    ctx = _native.Context(format=fmt, level=3, buffer_size=bs, lib=lib, max_slab_bytes=n)
    cap = ctx.slab_bound(n)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    ctx.set_profiling(True)
    nb = ctx.n_blocks(n)
    acc = {}
    for it in range(4):
        if it == 1:
            lib.L.gzpx_exp_cycles(cyc, 1)
        ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
        if it:
            for k, v in ctx.last_stage_ms().items():
                acc[k] = acc.get(k, 0.0) + v / 3
    lib.L.gzpx_exp_cycles(cyc, 1)
    c = [x / 3 / nb for x in cyc]
    tot = sum(c[:7])
    print("%s: match+parse %.2f ms; k_parse_hc cycles per block %.0f: " % (what, acc["k_match_hc+k_parse_hc"], tot) +
          ", ".join("%s %.0f (%.0f%%)" % (names[k], c[k], 100 * c[k] / tot) for k in range(7)) + "; %.1f walk rounds per block" % c[7],
          flush=True)
    ctx.close()
    del d_in, d_out

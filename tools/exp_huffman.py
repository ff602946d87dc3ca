"""Where k_huffman's cycles go (second library with -DGZPX_EXPERIMENT, never the product build): lane 0's
shader clock between the phases of every sub-block, summed over the blocks of a launch."""
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
                       "-DGZPX_EXPERIMENT", "-I", build.INCLUDE] + srcs + ["-o", exp])
lib = _native.GzpxLib(exp)
n = 576_716_800
d_in = torch.from_numpy(synth.text_slab(n, seed=20250927)).cuda()
ctx = _native.Context(format=0, level=1, buffer_size=65280, lib=lib, max_slab_bytes=n)
cap = ctx.slab_bound(n)
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
ctx.set_profiling(True)
nb = ctx.n_blocks(n)
cyc = (ctypes.c_ulonglong * 8)()
acc = {}
for it in range(4):
    if it == 1:
        lib.L.gzpx_exp_huff(cyc, 1)
    ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
    if it:
        for k, v in ctx.last_stage_ms().items():
            acc[k] = acc.get(k, 0.0) + v / 3
lib.L.gzpx_exp_huff(cyc, 1)
c = [x / 3 / nb for x in cyc]
names = ["setup", "litlen code", "offset code", "precode RLE", "precode code", "costs", "header + tables"]
print("k_huffman %.3f ms; cycles per block (all its sub-blocks): " % acc["k_huffman"] +
      ", ".join("%s %.0f" % (nm, v) for nm, v in zip(names, c)) + "; total %.0f" % sum(c[:7]))

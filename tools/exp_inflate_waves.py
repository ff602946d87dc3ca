"""k_inflate (global window) compiled for n waves per SIMD, n in argv (default 3 4 5 6 8): builds one
library per n (-DGZPX_INF_WAVES=n; never the product build) and times the inflate of the bench stream."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from gzp_amd import _native, build, synth

ns = [x for x in sys.argv[1:]] or ["3", "4", "5", "6", "8"]  # "w" or "w:r" (waves per SIMD : 64-bit-position groups per round)
n = 576_716_800
slab = synth.text_slab(n, seed=20250927)
with _native.Context(format=0, level=1, buffer_size=65280, max_slab_bytes=n) as c:
    comp = np.frombuffer(c.compress_slab(slab, True), dtype=np.uint8).copy()
d_in = torch.from_numpy(comp).cuda()
d_out = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
ref = torch.from_numpy(slab).cuda()
srcs = [os.path.join(build.CSRC, s) for s in build.SOURCES]
for spec in ns:
    w, r = (spec.split(":") + ["2"])[:2]
    w, r = int(w), int(r)
    so = os.path.join(build.LIB_DIR, "libgzpx_w%d_r%d.so" % (w, r))
    subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
                           "-DGZPX_INF_WAVES=%d" % w, "-DGZPX_INF_R=%d" % r, "-I", build.INCLUDE] + srcs + ["-o", so])
    lib = _native.GzpxLib(so)
    d = _native.DContext(format=0, lib=lib)
    offs, sizes, used = d.scan_blocks(comp)
    ms = []
    for it in range(6):
        got = d.decompress_device(d_in.data_ptr(), comp.size, offs, sizes, d_out.data_ptr(), n + 64)
        ms.append(d.last_inflate_ms())
    ok = got == n and bool(torch.equal(d_out[:n], ref))
    print("waves/SIMD %d, R %d: k_inflate %.3f ms  (%.1f GiB/s)  ok=%s" % (w, r, min(ms[1:]), n / 2**30 / (min(ms[1:]) * 1e-3), ok), flush=True)
    d.close()

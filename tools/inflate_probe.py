"""Throughput of the GPU inflate path (device-resident): compress a slab on the GPU, then time
gzpx_decompress_blocks_device over it.  usage: inflate_probe.py [mib] [class] [level] [reps]"""
import sys
import time

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gzp_amd import _native, synth

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cls = sys.argv[2] if len(sys.argv) > 2 else "textslab"
level = int(sys.argv[3]) if len(sys.argv) > 3 else 1
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
n = mib << 20
a = synth.text_slab(n, seed=5) if cls == "textslab" else synth.make(cls, n, 5)
lib = _native.load()
with _native.Context(level=level, lib=lib) as c:
    comp = np.frombuffer(c.compress_slab(a, True), dtype=np.uint8)
d = _native.DContext(lib=lib)
offs, sizes, used = d.scan_blocks(comp)
d_in = torch.from_numpy(comp.copy()).cuda()
d_out = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
for r in range(reps + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = d.decompress_device(d_in.data_ptr(), comp.size, offs, sizes, d_out.data_ptr(), n + 64)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if r:
        print(f"{cls} L{level} {mib} MiB  blocks={offs.size} ratio={n/comp.size:.2f}  {dt*1e3:.2f} ms  "
              f"{n/dt/2**30:.2f} GiB/s out")
assert got == n
d.debug_inflate(True)
d.decompress_device(d_in.data_ptr(), comp.size, offs, sizes, d_out.data_ptr(), n + 64)
c = d.debug_inflate(False)
nb = offs.size
print("per block: cycles %.0f  hdr+tables %.0f  round-setup %.0f  stores+copies %.0f | rounds %.0f lits %.0f matches %.0f flushes %.1f"
      % tuple(x / nb for x in c))
assert bytes(d_out[:n].cpu().numpy().tobytes()) == a.tobytes()
print("verified")

import sys, os, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from gzp_amd import _native, synth
n = 576_716_800
a = synth.text_slab(n)
d_in = torch.from_numpy(a).cuda()
ctx = _native.Context(format=0, level=1, buffer_size=65280, max_slab_bytes=n)
cap = ctx.slab_bound(n)
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
for i in range(2):
    out_len, nb = ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
cyc = ctx.debug_phase_cycles()
print("per block kcycles: total %.1f staging %.1f walk %.1f 3a %.1f | 3b thread0 done %.1f, all done %.1f" % tuple(cyc[i]/nb/1e3 for i in (2,3,4,5,7,0)))

import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from gzp_amd import _native, synth
n = 576_716_800
a = synth.text_slab(n)
d_in = torch.from_numpy(a).cuda()
ctx = _native.Context(format=0, level=1, buffer_size=65280, max_slab_bytes=n)
cap = ctx.slab_bound(n)
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
ctx.set_profiling(True)
for i in range(2):
    out_len, nb = ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
cyc = ctx.debug_phase_cycles()
print("k_huffman per block kcycles: total %.1f | sort %.1f build_tree %.1f length_counts %.1f" % tuple(cyc[i]/nb/1e3 for i in (7,3,4,5)), ctx.last_stage_ms()["k_huffman"])

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
ctx.debug_set_flags(64)
out_len, nb = ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
cyc = ctx.debug_phase_cycles()
waves = nb * 65280 / 64
print("per wave-step: max-iters c0 %.2f c1 %.2f ; mean per-lane iters c0 %.3f c1 %.3f" % (cyc[7]/waves, cyc[3]/waves, cyc[4]/waves/64, cyc[5]/waves/64))

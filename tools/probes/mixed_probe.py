import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from gzp_amd import _native, synth
flags = int(sys.argv[1]); cls = sys.argv[2] if len(sys.argv) > 2 else "mixed"
n = 256 << 20
base = synth.make(cls, 8 << 20, 5)
a = np.tile(base, n // base.size)
d_in = torch.from_numpy(a).cuda()
with _native.Context(format=0, level=3, buffer_size=65280, max_slab_bytes=n) as ctx:
    ctx.debug_set_flags(flags)
    cap = ctx.slab_bound(n)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    for _ in range(4):
        ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
    torch.cuda.synchronize()

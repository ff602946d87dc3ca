"""GPU diagnostic: per-stage ms and k_match_parse per-phase cycles on a text slab."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from gzp_amd import _native, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 576_716_800
cls = sys.argv[2] if len(sys.argv) > 2 else "slab"
level = int(sys.argv[3]) if len(sys.argv) > 3 else 1
fmt = int(sys.argv[4]) if len(sys.argv) > 4 else 0
bs = int(sys.argv[5]) if len(sys.argv) > 5 else 65280
a = synth.text_slab(n) if cls == "slab" else synth.make(cls, n, 3)
d_in = torch.from_numpy(a).cuda()
ctx = _native.Context(format=fmt, level=level, buffer_size=bs, max_slab_bytes=n)
cap = ctx.slab_bound(n)
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
ctx.set_profiling(True)
if os.environ.get("GZPX_DEBUG_FLAGS"):
    ctx.debug_set_flags(int(os.environ["GZPX_DEBUG_FLAGS"]))
for i in range(3):
    out_len, nb = ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
ms = ctx.last_stage_ms()
cyc = ctx.debug_phase_cycles()
cc = ctx.debug_cand_cycles()
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
out_len, nb = ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
torch.cuda.synchronize(); wall = time.perf_counter() - t0
print(json.dumps({"class": cls, "n": n, "level": level, "fmt": fmt, "bs": bs, "wall_ms": round(wall * 1e3, 2),
                  "GiB_per_s": round(n / wall / 2**30, 2), "ratio": out_len / n, "stage_ms": ms,
                  "kcycles_per_block[-,k_match,k_parse]": [round(c / nb / 1e3, 1) for c in cyc[:6]],
                  "mp_rounds_avg": cyc[6] / nb,
                  "cand_kcycles_per_block[hash+atomics,gather,file+store,total]": [round(c / nb / 1e3, 1) for c in cc]}))

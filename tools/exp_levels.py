"""Stage times of every built level (0-9) on one slab of the bench text: BGZF, 65280-byte blocks.
   python tools/exp_levels.py [MiB of input, default 128] [levels, default 1,3,5,6,7,8,9]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from gzp_amd import _native, synth

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 128
levels = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 3, 5, 6, 7, 8, 9]
n = mib << 20
slab = synth.text_slab(n, seed=20250927)
d_in = torch.from_numpy(slab).cuda()
for level in levels:
    ctx = _native.Context(format=0, level=level, buffer_size=65280, max_slab_bytes=n)
    cap = ctx.slab_bound(n)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        out_len = ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    ctx.set_profiling(True)
    ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
    st = ctx.last_stage_ms()
    ctx.set_profiling(False)
    olen = out_len[0] if isinstance(out_len, tuple) else out_len
    print("level %d: %8.2f ms  %8.0f MiB/s  ratio %.4f  stages %s" % (
        level, ms, mib / ms * 1e3, olen / n, {k: round(v, 2) for k, v in st.items() if v > 0.005}), flush=True)
    ctx.close()
    del d_out

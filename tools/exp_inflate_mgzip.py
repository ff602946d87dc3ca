"""Phase clocks of k_inflate_seg<W = 8> (debug launch) on Mgzip streams of 1 MiB blocks: configs[2]'s class and text."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch
from gzp_amd import _native, synth

lib = _native.load()
names = ["cycles", "hdr+tables", "pass1", "pass2", "pass3", "spans", "p2 iters", "steps"]
n = 256 << 20
for cls, level in [("ascii", 3), ("text", 3), ("text", 1)]:
    a = synth.text_slab(n) if cls == "text" else synth.make(cls, n, 3)
    with _native.Context(format=_native.FORMAT_MGZIP, level=level, buffer_size=1 << 20, max_slab_bytes=n, lib=lib) as c:
        comp = np.frombuffer(c.compress_slab(a, True), dtype=np.uint8).copy()
    d = _native.DContext(format=_native.FORMAT_MGZIP, lib=lib)
    offs, sizes, used = d.scan_blocks(comp)
    d_in = torch.from_numpy(comp).cuda()
    d_out = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    for _ in range(2):
        d.decompress_device(d_in.data_ptr(), comp.size, offs, sizes, d_out.data_ptr(), n + 64)
    ms, st = [], None
    for _ in range(4):
        d.decompress_device(d_in.data_ptr(), comp.size, offs, sizes, d_out.data_ptr(), n + 64)
        ms.append(d.last_inflate_ms())
        st = d.last_inflate_stage_ms()
    ok = bool((d_out[:n].cpu() == torch.from_numpy(a)).all())
    d.debug_inflate(True)
    d.decompress_device(d_in.data_ptr(), comp.size, offs, sizes, d_out.data_ptr(), n + 64)
    sums = d.debug_inflate(False)
    nb = len(offs)
    print("%-6s l%d %d members ratio %.3f | %.3f ms (%.1f GiB/s) stages %s ok=%s redo %d" % (
        cls, level, nb, comp.size / n, min(ms), n / 2**30 / (min(ms) * 1e-3), st, ok, d.last_redo_count()))
    print("       per member (wave 0's clocks): " + ", ".join("%s %.0f" % (k, v / nb) for k, v in zip(names, sums)))
    d.close()

"""Phase clocks of k_inflate_seg / k_lzcopy (debug launch) on the bench stream's first 64 MiB and on other classes."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch
from gzp_amd import _native, synth

lib = _native.load()
names = ["cycles", "hdr+tables", "pass1", "pass2", "pass3", "spans+rounds", "p2 iters", "bytes"]
for cls, n, level in [("text", 64 << 20, 1), ("text", 64 << 20, 3), ("dna", 64 << 20, 1), ("fastq", 64 << 20, 1), ("ascii", 64 << 20, 3),
                      ("runs", 64 << 20, 1), ("zeros", 64 << 20, 1), ("repeats", 64 << 20, 1), ("mixed", 64 << 20, 1), ("random", 64 << 20, 1)]:
    a = synth.text_slab(n) if cls == "text" else synth.make(cls, n, 3)
    with _native.Context(level=level, max_slab_bytes=n, lib=lib) as c:
        comp = np.frombuffer(c.compress_slab(a, True), dtype=np.uint8).copy()
    d = _native.DContext(lib=lib)
    offs, sizes, used = d.scan_blocks(comp)
    d_in = torch.from_numpy(comp).cuda()
    d_out = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    row = {}
    for route in (_native.INFLATE_SEG, _native.INFLATE_WAVE):
        d.set_route(route)
        for _ in range(2):
            d.decompress_device(d_in.data_ptr(), comp.size, offs, sizes, d_out.data_ptr(), n + 64)
        ms = []
        for _ in range(5):
            d.decompress_device(d_in.data_ptr(), comp.size, offs, sizes, d_out.data_ptr(), n + 64)
            ms.append(d.last_inflate_ms())
        ok = bool((d_out[:n].cpu() == torch.from_numpy(a)).all())
        row[route] = (min(ms), ok, d.last_redo_count() if route == _native.INFLATE_SEG else 0)
    d.set_route(_native.INFLATE_SEG)
    d.debug_inflate(True)
    d.decompress_device(d_in.data_ptr(), comp.size, offs, sizes, d_out.data_ptr(), n + 64)
    sums = d.debug_inflate(False)
    nb = len(offs)
    print("%-8s l%d  %d blocks ratio %.3f | seg %.3f ms (%.1f GiB/s, ok=%s, redo %d) wave %.3f ms (%.1f GiB/s, ok=%s)" % (
        cls, level, nb, comp.size / n, row[0][0], n / 2**30 / (row[0][0] * 1e-3), row[0][1], row[0][2], row[1][0], n / 2**30 / (row[1][0] * 1e-3), row[1][1]))
    print("         k_inflate_seg per block: " + ", ".join("%s %.0f" % (k, v / nb) for k, v in zip(names, sums)))
    d.debug_inflate(2)
    d.decompress_device(d_in.data_ptr(), comp.size, offs, sizes, d_out.data_ptr(), n + 64)
    sums = d.debug_inflate(False)
    names2 = ["cycles", "stage in", "chunk set-up", "polling", "write out", "poll iters", "idle iters", "matches"]
    print("         k_lzcopy      per block: " + ", ".join("%s %.0f" % (k, v / nb) for k, v in zip(names2, sums)))
    d.close()

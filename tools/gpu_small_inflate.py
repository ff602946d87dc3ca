"""Small GPU check of the inflate routes (used while debugging; strict timeouts outside)."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from gzp_amd import _native, synth
lib = _native.load()
sizes_blk = [int(x) for x in sys.argv[1:]] or [1, 3, 100, 5000, 9000]
for nblk in sizes_blk:
    n = 65280 * nblk
    a = synth.text_slab(n)
    with _native.Context(level=1, max_slab_bytes=n, lib=lib) as c:
        comp = np.frombuffer(c.compress_slab(a, True), dtype=np.uint8).copy()
    print("compressed", nblk, comp.size, flush=True)
    d = _native.DContext(lib=lib)
    offs, sizes, used = d.scan_blocks(comp)
    d_in = torch.from_numpy(comp).cuda(); d_out = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    print("launch", flush=True)
    d.decompress_device(d_in.data_ptr(), comp.size, offs, sizes, d_out.data_ptr(), n + 64)
    ok = bool((d_out[:n].cpu() == torch.from_numpy(a)).all())
    print(nblk, ok, 'redo', d.last_redo_count(), flush=True)
    d.close()

import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from gzp_amd import _native, synth
n = 512 << 20
base = _native.load()
for cls, level in (("ascii", 3), ("text", 1), ("text", 3), ("fastq", 1)):
    a = synth.text_slab(n) if cls == "text" else synth.make(cls, n, 3)
    with _native.Context(format=_native.FORMAT_MGZIP, level=level, buffer_size=1 << 20, max_slab_bytes=n, lib=base) as c:
        comp = np.frombuffer(c.compress_slab(a, True), dtype=np.uint8).copy()
    d_in = torch.from_numpy(comp).cuda(); d_out = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    for path in sys.argv[1:]:
        d = _native.DContext(format=_native.FORMAT_MGZIP, lib=_native.GzpxLib(path))
        offs, sizes, used = d.scan_blocks(comp)
        best = None
        for _ in range(5):
            d.decompress_device(d_in.data_ptr(), comp.size, offs, sizes, d_out.data_ptr(), n + 64)
            st = d.last_inflate_stage_ms()
            best = st if best is None or st[0] + st[1] < best[0] + best[1] else best
        ok = bool((d_out[:n].cpu() == torch.from_numpy(a)).all())
        print(cls, level, os.path.basename(path), "decode %.3f copy %.3f ms" % best, ok, d.last_redo_count())
        d.close()

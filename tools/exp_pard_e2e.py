"""Host-to-host rate of the ParDecompress twin (Read API) for several slab sizes and stream lengths: what a gzp user sees."""
import io, os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from gzp_amd import _native, par, synth
lib = _native.load()
for mib in (550, 2200):
    a = synth.text_slab(mib << 20) if mib == 550 else np.tile(synth.text_slab(550 << 20), 4)
    with _native.Context(level=1, max_slab_bytes=a.size, lib=lib) as c:
        comp = bytes(c.compress_slab(a, True))
    buf = np.empty(64 << 20, dtype=np.uint8)
    for batch in (4, 16, 64):
        for rep in range(3):
            t0 = time.perf_counter()
            r = par.ParDecompressBuilder(par.Bgzf, lib=lib).batch_bytes(batch << 20).from_reader(io.BytesIO(comp))
            t1 = time.perf_counter()
            total = 0
            first = None
            mode = ("read", "readinto", "fill_buf")[rep % 3]
            while True:
                if mode == "read":
                    k = len(r.read(64 << 20))
                elif mode == "readinto":
                    k = r.readinto(buf)
                else:
                    k = len(r.fill_buf())
                    r.consume(k)
                if first is None:
                    first = time.perf_counter() - t1
                if not k:
                    break
                total += k
            dt = time.perf_counter() - t1
            r.close()
            print("stream %4d MiB, slabs of %2d MiB compressed, %-8s: first bytes after %.1f ms, to the end %.1f ms = %.2f GiB/s" % (
                mib, batch, mode, first * 1e3, dt * 1e3, total / dt / 2**30), flush=True)

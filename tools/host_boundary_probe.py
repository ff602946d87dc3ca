"""PCIe-inclusive rates of the host-buffer boundary (not the bench value): gzpx_compress_slab on a
pageable numpy buffer, and the ParCompress twin (pinned staging, two device lanes)."""
import io
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

from gzp_amd import _native, par, synth

n = 576_716_800
a = synth.text_slab(n, seed=20250927)
lib = _native.load()
with _native.Context(level=1, lib=lib, max_slab_bytes=n) as c:
    c.compress_slab(a[:1 << 24], True)
    for _ in range(2):
        t0 = time.perf_counter()
        out = c.compress_slab(a, True)
        dt = time.perf_counter() - t0
    print("gzpx_compress_slab (pageable host in/out): %.1f ms = %.2f GiB/s" % (dt * 1e3, n / dt / 2**30))


class Sink(io.RawIOBase):
    def __init__(self):
        self.n = 0

    def writable(self):
        return True

    def write(self, b):
        self.n += len(b)
        return len(b)


buf = a.tobytes()
for _ in range(2):
    s = Sink()
    t0 = time.perf_counter()
    w = par.ParCompressBuilder(par.Bgzf, lib=lib).compression_level(par.Compression(1)).from_writer(s)
    t1 = time.perf_counter()
    w.write(buf)
    w.finish()
    dt = time.perf_counter() - t1
print("ParCompress twin (two lanes): builder %.1f ms; write_all + finish %.1f ms = %.2f GiB/s (%d bytes out)"
      % ((t1 - t0) * 1e3, dt * 1e3, n / dt / 2**30, s.n))

comp = out if isinstance(out, (bytes, bytearray)) else bytes(out)
for _ in range(2):
    t0 = time.perf_counter()
    r = par.ParDecompressBuilder(par.Bgzf, lib=lib).from_reader(io.BytesIO(comp))
    t1 = time.perf_counter()
    total = 0
    while True:
        chunk = r.read(64 << 20)
        if not chunk:
            break
        total += len(chunk)
    dt = time.perf_counter() - t1
    r.close()
print("ParDecompress twin: builder %.1f ms; read to end %.1f ms = %.2f GiB/s (%d bytes)"
      % ((t1 - t0) * 1e3, dt * 1e3, total / dt / 2**30, total))
with _native.DContext(lib=lib) as d:
    for _ in range(2):
        t0 = time.perf_counter()
        back = d.decompress(comp)
        dt = time.perf_counter() - t0
    print("gzpx_decompress_blocks (pageable host in/out, incl. header walk): %.1f ms = %.2f GiB/s" % (dt * 1e3, len(back) / dt / 2**30))

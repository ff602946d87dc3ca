"""Randomised soak of the ParCompress / ParDecompress twins on the GPU: random write sizes (and
flushes) must give the stream the oracle predicts -- blocks are cut by the buffered byte count alone,
a flush closes the pending short block -- and ParDecompress must give the bytes back.
usage: gpu_fuzz_twin.py [seconds] [seed]"""
import io
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

from gzp_amd import _native, par, synth
from oracle import oracle

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
# (GZPX_LIB=<path>: another build of the library, e.g. tests/emu/libgzpx_emu.so -- the same soak without a GPU)
lib = _native.GzpxLib(os.environ["GZPX_LIB"]) if os.environ.get("GZPX_LIB") else _native.load()
classes = sorted(synth.CLASSES)
t_end = time.time() + secs
cases = bad = 0
while time.time() < t_end:
    fmt_cls, fmt, bs = (par.Bgzf, oracle.FMT_BGZF, 65280) if rng.random() < 0.6 else (par.Mgzip, oracle.FMT_MGZIP,
                                                                                     int(rng.choice([131072, 70000])))
    level = int(rng.integers(0, 10))
    n = int(rng.integers(0, 12 * bs))
    cls = classes[rng.integers(len(classes))]
    if cls == "random" and fmt == oracle.FMT_BGZF and level == 0:
        pass
    data = synth.make(cls, n, int(rng.integers(1, 1 << 30))).tobytes()
    batch = int(rng.choice([1, 2, 3, 1024]))
    sink = io.BytesIO()
    w = (par.ParCompressBuilder(fmt_cls, lib=lib).compression_level(par.Compression(level)).buffer_size(bs)
         .batch_blocks(batch).from_writer(sink))
    pos = 0
    pieces = []  # the stream is the concatenation of one oracle stream per flushed piece
    start = 0
    while pos < n:
        step = int(rng.integers(1, 3 * bs)) if rng.random() < 0.8 else int(rng.integers(1, 200))
        w.write(data[pos:pos + step])
        pos = min(n, pos + step)
        if rng.random() < 0.1:
            w.flush()
            pieces.append((start, pos))
            start = pos
    w.finish()
    want = b""
    for (a, b) in pieces:  # a flush emits what is buffered as blocks without the EOF marker
        s = oracle.compress_stream(np.frombuffer(data[a:b], dtype=np.uint8), fmt, level, oracle.COMPAT_1_24, bs)
        want += s[:-28] if fmt == oracle.FMT_BGZF else s
    want += oracle.compress_stream(np.frombuffer(data[start:], dtype=np.uint8), fmt, level, oracle.COMPAT_1_24, bs)
    got = sink.getvalue()
    cases += 1
    if got != want:
        bad += 1
        print("STREAM MISMATCH", cls, n, level, fmt, bs, batch, len(pieces), flush=True)
        continue
    r = par.ParDecompressBuilder(fmt_cls, lib=lib).batch_bytes(int(rng.choice([1 << 16, 1 << 20, 1 << 26]))).from_reader(
        io.BytesIO(got))
    back = r.read()
    r.close()
    if back != data:
        bad += 1
        print("ROUND TRIP MISMATCH", cls, n, level, fmt, bs, flush=True)
print("gpu_fuzz_twin: %d cases, %d failures" % (cases, bad))
sys.exit(1 if bad else 0)

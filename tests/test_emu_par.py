"""The ParCompress twin (Write / flush / finish / Drop semantics) through
the emulated library, against the oracle's stream restatement.  No GPU."""
import gzip
import io

import numpy as np
import pytest

from gzp_amd import _native, par, synth


def _builder(emu_lib, fmt=par.Bgzf, **kw):
    b = par.ParCompressBuilder(fmt, lib=emu_lib).compression_level(par.Compression.fast()) \
        .compat(_native.COMPAT_1_10).batch_blocks(kw.get("batch", 2)).num_threads(kw.get("threads", 4))
    if "buffer_size" in kw:
        b.buffer_size(kw["buffer_size"])
    return b


@pytest.mark.parametrize("n,chunk", [(0, 1), (1, 1), (65280, 65536), (65281, 4096), (5 * 65280, 65536),
                                     (5 * 65280 + 17, 9973), (300001, 300001)])
def test_write_chunks_finish_matches_reference_stream(emu_lib, oracle, n, chunk):
    a = synth.make("text", n, 11)
    sink = io.BytesIO()
    w = _builder(emu_lib).from_writer(sink)
    for i in range(0, n, chunk):  # benches/bench.rs:36-45 shape: fixed-size write_all calls
        w.write_all(a[i:i + chunk])
    assert w.finish() is sink
    w.close()
    want = oracle.compress_stream(a, oracle.FMT_BGZF, 1, oracle.COMPAT_1_10, 65280)
    assert sink.getvalue() == want
    assert gzip.decompress(sink.getvalue()) == a.tobytes()


def test_exactly_one_block_is_held_until_finish(emu_lib, oracle):
    # Q1: a buffer holding exactly buffer_size bytes is not dispatched by write()
    a = synth.make("fastq", 65280, 2)
    sink = io.BytesIO()
    w = _builder(emu_lib, batch=1).from_writer(sink)
    w.write_all(a)
    w.write_all(b"")
    w.finish()
    w.close()
    out = sink.getvalue()
    assert out == oracle.compress_stream(a, oracle.FMT_BGZF, 1, oracle.COMPAT_1_10, 65280)
    assert out.count(b"\x1f\x8b\x08\x04") >= 2  # one data block + EOF marker


def test_flush_emits_short_and_empty_blocks(emu_lib, oracle):
    # Q2: flush() = flush_last(false): a block for whatever is buffered, possibly empty
    a = synth.make("text", 100000, 5)
    sink = io.BytesIO()
    w = _builder(emu_lib).from_writer(sink)
    w.write_all(a[:70000])
    w.flush()
    w.flush()  # nothing buffered: empty block, no EOF
    w.write_all(a[70000:])
    w.finish()
    w.close()
    enc = lambda x, last: oracle.encode_block(x, oracle.FMT_BGZF, 1, oracle.COMPAT_1_10, last)
    want = (enc(a[:65280], False) + enc(a[65280:70000], False) + enc(a[:0], False) +
            enc(a[70000:], True))
    assert sink.getvalue() == want
    assert gzip.decompress(sink.getvalue()) == a.tobytes()


def test_drop_without_finish_finishes(emu_lib, oracle):
    # src/deflate.rs:745-775 (drop-without-finish) / src/par/compress.rs:391-402
    a = synth.make("dna", 70001, 1)
    sink = io.BytesIO()
    w = _builder(emu_lib).from_writer(sink)
    w.write_all(a)
    w.close()  # Drop
    assert sink.getvalue() == oracle.compress_stream(a, oracle.FMT_BGZF, 1, oracle.COMPAT_1_10, 65280)


def test_mgzip_has_no_eof_marker(emu_lib, oracle):
    a = synth.make("mixed", 100000, 3)
    sink = io.BytesIO()
    w = _builder(emu_lib, par.Mgzip, buffer_size=40000).from_writer(sink)
    w.write_all(a)
    w.finish()
    w.close()
    assert sink.getvalue() == oracle.compress_stream(a, oracle.FMT_MGZIP, 1, oracle.COMPAT_1_10, 40000)


def test_builder_errors_mirror_gzp(emu_lib):
    with pytest.raises(par.GzpError) as e:
        par.ParCompressBuilder(par.Bgzf, lib=emu_lib).buffer_size(100)
    assert e.value.code == _native.ERR_BUFFER_SIZE
    with pytest.raises(par.GzpError) as e:
        par.ParCompressBuilder(par.Bgzf, lib=emu_lib).num_threads(0)
    assert e.value.code == _native.ERR_NUM_THREADS
    with pytest.raises(par.GzpError) as e:
        par.ParCompressBuilder(par.Bgzf, lib=emu_lib).compression_level(13).from_writer(io.BytesIO())
    assert e.value.code == _native.ERR_COMPRESSION_LEVEL


def test_writer_io_error_is_preserved(emu_lib):
    class Broken:
        def write(self, b):
            raise BrokenPipeError("sink closed")

    w = _builder(emu_lib, batch=1).from_writer(Broken())
    a = synth.make("text", 4 * 65280, 1)
    with pytest.raises(par.GzpError) as e:
        w.write_all(a)
        w.write_all(a)
        w.finish()
    assert e.value.code in (_native.ERR_IO, _native.ERR_CHANNEL)
    with pytest.raises(par.GzpError) as e2:
        w.finish() if not w._finished else w.write_all(a)
    w.close()


def test_block_size_exceeded_surfaces(emu_lib):
    sink = io.BytesIO()
    w = _builder(emu_lib, buffer_size=65536).from_writer(sink)
    with pytest.raises(par.GzpError) as e:
        w.write_all(synth.uniform_random(65536, 1))
        w.finish()
    assert e.value.code == _native.ERR_BLOCK_SIZE_EXCEEDED
    w.close()


def test_zbuilder(emu_lib, oracle):
    a = synth.make("text", 70000, 9)
    sink = io.BytesIO()
    z = par.ZBuilder(par.Bgzf, lib=emu_lib).num_threads(0).compression_level(par.Compression.fast())
    z._b.compat(_native.COMPAT_1_10)
    w = z.from_writer(sink)
    w.write_all(a)
    w.finish()
    w.close()
    assert sink.getvalue() == oracle.compress_stream(a, oracle.FMT_BGZF, 1, oracle.COMPAT_1_10, 65280)


def test_random_write_sizes_and_flushes_match_the_oracle(emu_lib, oracle):
    # seeded soak of the twin's buffering (slabs cut straight from the caller's bytes into pooled
    # staging buffers): any chunking of the writes gives the same stream; a flush closes the pending
    # short block (no EOF marker); ParDecompress gives the bytes back
    import io
    rng = np.random.default_rng(20250927)
    classes = sorted(synth.CLASSES)
    for case in range(16):
        fmt_cls, fmt, bs = (par.Bgzf, oracle.FMT_BGZF, 65280) if case % 3 else (par.Mgzip, oracle.FMT_MGZIP, 70000)
        level = int(rng.integers(0, 5))
        n = int(rng.integers(0, 4 * bs))
        data = synth.make(classes[rng.integers(len(classes))], n, int(rng.integers(1, 1 << 30))).tobytes()
        sink = io.BytesIO()
        w = (par.ParCompressBuilder(fmt_cls, lib=emu_lib).compression_level(par.Compression(level)).buffer_size(bs)
             .batch_blocks(int(rng.choice([1, 2, 5]))).from_writer(sink))
        pos, start, pieces = 0, 0, []
        while pos < n:
            step = int(rng.integers(1, 3 * bs)) if rng.random() < 0.7 else int(rng.integers(1, 300))
            w.write(data[pos:pos + step])
            pos = min(n, pos + step)
            if rng.random() < 0.15:
                w.flush()
                pieces.append((start, pos))
                start = pos
        w.finish()
        want = b""
        for a, b in pieces:
            s = oracle.compress_stream(np.frombuffer(data[a:b], dtype=np.uint8), fmt, level, oracle.COMPAT_1_24, bs)
            want += s[:-28] if fmt == oracle.FMT_BGZF else s
        want += oracle.compress_stream(np.frombuffer(data[start:], dtype=np.uint8), fmt, level, oracle.COMPAT_1_24, bs)
        assert sink.getvalue() == want, (case, n, level, bs, len(pieces))
        r = par.ParDecompressBuilder(fmt_cls, lib=emu_lib).batch_bytes(1 << 16).from_reader(io.BytesIO(want))
        assert r.read() == data, case
        r.close()

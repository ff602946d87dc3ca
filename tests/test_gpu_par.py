"""The ParCompress twin on the real HIP library: streaming writes through two device lanes."""
import gzip
import io

import pytest

from gzp_amd import _native, par, synth

pytestmark = pytest.mark.gpu


def test_streaming_write_finish(hip_lib, oracle):
    a = synth.text_slab(40 * 65280 + 777, 3_000_000, 5)
    sink = io.BytesIO()
    w = par.ParCompressBuilder(par.Bgzf, lib=hip_lib).compression_level(par.Compression.fast()) \
        .num_threads(8).batch_blocks(8).from_writer(sink)
    for i in range(0, a.size, 65536):  # benches/bench.rs:36-45: 64 KiB write_all chunks
        w.write_all(a[i:i + 65536])
    w.finish()
    w.close()
    assert sink.getvalue() == oracle.compress_stream(a, oracle.FMT_BGZF, 1, oracle.COMPAT_1_24, 65280)
    assert gzip.decompress(sink.getvalue()) == a.tobytes()


def test_flush_and_drop(hip_lib, oracle):
    a = synth.make("fastq", 200000, 5)
    sink = io.BytesIO()
    w = par.ParCompressBuilder(par.Bgzf, lib=hip_lib).compression_level(1).batch_blocks(2).from_writer(sink)
    w.write_all(a[:70000])
    w.flush()
    w.write_all(a[70000:])
    w.close()  # Drop finishes
    enc = lambda x, last: oracle.encode_block(x, oracle.FMT_BGZF, 1, oracle.COMPAT_1_24, last)
    want = enc(a[:65280], False) + enc(a[65280:70000], False)
    rest = a[70000:]
    want += enc(rest[:65280], False) + enc(rest[65280:], True)
    assert sink.getvalue() == want

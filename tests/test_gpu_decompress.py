"""GPU ParDecompress<Bgzf/Mgzip> (BASELINE.json configs[5]): inflate kernels on the MI355X against
the original input of the compressor, zlib-made foreign members, error classes.  Through the C ABI."""
import io
import struct
import zlib

import numpy as np
import pytest

from gzp_amd import _native, par, synth

pytestmark = pytest.mark.gpu


def bgzf_member(chunk, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    payload = co.compress(chunk) + co.flush()
    hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, ord("B"), ord("C"), 2, len(payload) + 25)
    return hdr + payload + struct.pack("<II", zlib.crc32(chunk), len(chunk))


@pytest.fixture(scope="module", params=["seg", "wave"])
def dctx(request, hip_lib):
    """Both inflate routes: the decode / LZ-copy pair (k_inflate_seg + k_lzcopy, the default; k_inflate takes what they
    hand back) and k_inflate for every member."""
    c = _native.DContext(lib=hip_lib)
    c.set_route(_native.INFLATE_SEG if request.param == "seg" else _native.INFLATE_WAVE)
    c.route_name = request.param
    yield c
    c.close()


@pytest.mark.parametrize("level", [1, 3])
def test_roundtrip_all_classes(dctx, hip_lib, level):
    with _native.Context(level=level, lib=hip_lib) as c:
        for cls in sorted(synth.CLASSES):
            a = synth.make(cls, 40 * 65280 + 4321, 17)
            comp = c.compress_slab(a, True)
            assert dctx.decompress(comp) == a.tobytes(), cls


def test_decode_copy_pair_takes_ordinary_members_itself(hip_lib):
    """The default route must not pass by way of k_inflate: nothing of an ordinary stream is handed back; a member
    that does not decode (a byte of its payload flipped) is, and comes out with the error class k_inflate gives it."""
    a = synth.text_slab(200 * 65280, seed=11)
    with _native.Context(level=1, lib=hip_lib) as c:
        comp = c.compress_slab(a, True)
    with _native.DContext(lib=hip_lib) as d:
        assert d.decompress(comp) == a.tobytes() and d.last_redo_count() == 0
        st = d.last_inflate_stage_ms()
        assert st[0] > 0 and st[1] > 0
        bad = bytearray(comp)
        offs, sizes, _ = d.scan_blocks(bytes(comp))
        bad[int(offs[3]) + 18 + 40] ^= 0x55
        with pytest.raises(_native.GzpxError) as e:
            d.decompress(bytes(bad))
        assert e.value.code in (_native.ERR_BAD_DATA, _native.ERR_INVALID_CHECK, _native.ERR_INSUFFICIENT_SPACE) and e.value.block == 3
        with _native.DContext(lib=hip_lib) as w:
            w.set_route(_native.INFLATE_WAVE)
            with pytest.raises(_native.GzpxError) as e2:
                w.decompress(bytes(bad))
        assert (e2.value.code, e2.value.block) == (e.value.code, e.value.block)


def flushed_member(chunk, level, every, strategy=zlib.Z_DEFAULT_STRATEGY):
    """A BGZF member whose DEFLATE stream is flushed every `every` bytes: hundreds of blocks, an end-of-block code each."""
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    parts = []
    for i in range(0, len(chunk), every):
        parts.append(co.compress(chunk[i:i + every]))
        parts.append(co.flush(zlib.Z_FULL_FLUSH if (i // every) % 3 == 0 else zlib.Z_SYNC_FLUSH))
    parts.append(co.flush())
    payload = b"".join(parts)
    hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, ord("B"), ord("C"), 2, len(payload) + 25)
    return hdr + payload + struct.pack("<II", zlib.crc32(chunk), len(chunk))


def test_speculative_paths_walk_on_behind_garbage_codes(hip_lib):
    """k_inflate_seg's speculative paths step over invalid codewords and walk on behind end-of-block codes (round 6): the
    streams where they meet such garbage most -- printable noise, whose codes are all 6-7 bits long, so that a path stays
    out of step for long and the 15-bit end-of-block code turns up by chance; Huffman-only members; members of hundreds of
    tiny blocks -- come out right, and the noise of our own compressor without a member handed back to k_inflate."""
    noise = synth.make("ascii", 24 << 20, 5)
    with _native.DContext(lib=hip_lib) as d:
        with _native.Context(level=3, lib=hip_lib) as c:
            comp = c.compress_slab(noise, True)
        assert d.decompress(comp) == noise.tobytes() and d.last_redo_count() == 0
    with _native.DContext(format=_native.FORMAT_MGZIP, lib=hip_lib) as d:
        with _native.Context(format=_native.FORMAT_MGZIP, level=3, buffer_size=1 << 20, lib=hip_lib) as c:
            comp = c.compress_slab(noise, True)
        assert d.decompress(comp) == noise.tobytes() and d.last_redo_count() == 0
    with _native.DContext(lib=hip_lib) as d:
        for cls, seed in (("ascii", 1), ("text", 2), ("random", 3), ("dna", 4)):
            a = synth.make(cls, 480000, seed).tobytes()
            for strategy in (zlib.Z_HUFFMAN_ONLY, zlib.Z_FIXED, zlib.Z_DEFAULT_STRATEGY):
                s = b"".join(bgzf_member(a[i:i + 60000], 6, strategy) for i in range(0, len(a), 60000))
                assert d.decompress(s) == a, (cls, strategy)
            for every in (37, 300, 4000):
                s = b"".join(flushed_member(a[i:i + 40000], 6, every) for i in range(0, 240000, 40000))
                assert d.decompress(s) == a[:240000], (cls, every)


def test_bgzip_like_streams_stay_on_the_pair(hip_lib):
    """What bgzip / htslib write -- zlib's default strategy, 65,280-byte members -- is inflated by the decode / copy pair
    alone: no member handed back to k_inflate (the exotic strategies of the fuzzer are where hand-backs come from)."""
    with _native.DContext(lib=hip_lib) as d:
        for cls in ("text", "fastq", "dna", "ascii", "mixed"):
            a = synth.make(cls, 4 << 20, 9).tobytes()
            for level in (1, 6, 9):
                s = b"".join(bgzf_member(a[i:i + 65280], level) for i in range(0, len(a), 65280))
                assert d.decompress(s) == a and d.last_redo_count() == 0, (cls, level)


def test_first_block_hint_may_be_stale(hip_lib):
    """k_inflate_seg guesses where a BGZF member's first block ends from the members it decoded before (a word of the
    context): streams of different kinds through ONE context, back and forth -- the hint of one is the wrong guess for
    the next -- come out right, with no member handed back."""
    streams = []
    for cls, seed in (("dna", 1), ("text", 2), ("fastq", 3), ("ascii", 4), ("zeros", 5), ("text", 6)):
        a = synth.make(cls, 40 * 65280 + 123, seed)
        with _native.Context(level=1, lib=hip_lib) as c:
            streams.append((a.tobytes(), c.compress_slab(a, True)))
    with _native.DContext(lib=hip_lib) as d:
        for _ in range(2):
            for raw, comp in streams:
                assert d.decompress(comp) == raw and d.last_redo_count() == 0


def test_config5_shape_256mib(dctx, hip_lib):
    # configs[5]: inflate the output of configs[2] (256 MiB here), verify per-block CRC on device
    a = synth.text_slab(256 << 20, seed=5)
    with _native.Context(level=1, lib=hip_lib) as c:
        comp = c.compress_slab(a, True)
    out = dctx.decompress(comp)
    assert len(out) == a.size and zlib.crc32(out) == zlib.crc32(a.tobytes())
    assert out == a.tobytes()


def test_foreign_zlib_members(dctx):
    for level in (1, 6, 9):
        for cls in sorted(synth.CLASSES):
            a = synth.make(cls, 300000, 7 + level).tobytes()
            stream = b"".join(bgzf_member(a[i:i + 60000], level) for i in range(0, len(a), 60000))
            assert dctx.decompress(stream) == a, (level, cls)
    tiny = b"hello hello hello"
    assert dctx.decompress(bgzf_member(tiny, 9, zlib.Z_FIXED)) == tiny
    far = synth.uniform_random(32768, 1).tobytes()
    far = far + far[:5000]
    assert dctx.decompress(bgzf_member(far, 9)) == far


def test_mgzip_large_blocks(hip_lib):
    a = synth.make("mixed", (9 << 20) + 999, 3)
    with _native.Context(format=_native.FORMAT_MGZIP, level=3, buffer_size=4 << 20, lib=hip_lib) as c:
        comp = c.compress_slab(a, True)
    with _native.DContext(format=_native.FORMAT_MGZIP, lib=hip_lib) as d:
        assert d.decompress(comp) == a.tobytes()


def test_error_classes(dctx, hip_lib):
    a = synth.make("text", 5 * 65280, 2)
    with _native.Context(level=1, lib=hip_lib) as c:
        comp = c.compress_slab(a, True)
    bad = bytearray(comp)
    bad[12] = ord("X")
    with pytest.raises(_native.GzpxError) as e:
        dctx.decompress(bytes(bad))
    assert e.value.code == _native.ERR_INVALID_HEADER
    offs, sizes, _ = dctx.scan_blocks(comp)
    bad = bytearray(comp)
    bad[int(offs[2]) + int(sizes[2]) - 8] ^= 0xFF
    with pytest.raises(_native.GzpxError) as e:
        dctx.decompress(bytes(bad))
    assert e.value.code == _native.ERR_INVALID_CHECK and e.value.block == 2
    bad = bytearray(comp)
    for k in range(int(offs[1]) + 200, int(offs[1]) + 260):
        bad[k] ^= 0x5A
    with pytest.raises(_native.GzpxError) as e:
        dctx.decompress(bytes(bad))
    assert e.value.code in (_native.ERR_BAD_DATA, _native.ERR_INVALID_CHECK, _native.ERR_INSUFFICIENT_SPACE)
    assert e.value.block == 1
    # a member that inflates to fewer bytes than its ISIZE footer claims: BadData (libdeflate's
    # SHORT_OUTPUT through decode_block), never a CRC verdict over bytes the member did not produce
    bad = bytearray(comp)
    isize_pos = int(offs[0]) + int(sizes[0]) - 4
    bad[isize_pos:isize_pos + 4] = (int.from_bytes(bad[isize_pos:isize_pos + 4], "little") + 5).to_bytes(4, "little")
    with pytest.raises(_native.GzpxError) as e:
        dctx.decompress(bytes(bad))
    assert e.value.code == _native.ERR_BAD_DATA and e.value.block == 0
    # the context stays usable after an error
    assert dctx.decompress(comp) == a.tobytes()


def test_par_decompress_reader(hip_lib):
    a = synth.make("fastq", 200 * 65280 + 321, 9)
    w = io.BytesIO()
    pc = par.ParCompressBuilder(par.Bgzf, lib=hip_lib).from_writer(w)
    pc.write(a.tobytes())
    pc.finish()
    comp = w.getvalue()
    r = par.ParDecompressBuilder(par.Bgzf, lib=hip_lib).batch_bytes(1 << 20).from_reader(io.BytesIO(comp))
    assert r.read(12345) + r.read() == a.tobytes()
    r.close()
    r = par.ParDecompressBuilder(par.Bgzf, lib=hip_lib).from_reader(io.BytesIO(comp[:len(comp) // 2]))
    with pytest.raises(par.GzpError) as e:
        r.read()
    assert e.value.code == _native.ERR_IO
    r.close()

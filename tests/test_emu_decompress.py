"""ParDecompress<Bgzf/Mgzip> through the emulated kernels: our own streams, foreign (zlib-made)
members, the reference's own round-trip test inputs, error classes.  No GPU."""
import io
import struct
import zlib

import numpy as np
import pytest

from gzp_amd import _native, par, synth


def bgzf_member(chunk, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    payload = co.compress(chunk) + co.flush()
    hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, ord("B"), ord("C"), 2, len(payload) + 25)
    return hdr + payload + struct.pack("<II", zlib.crc32(chunk), len(chunk))


@pytest.fixture(scope="module", params=["seg", "wave"])
def dctx(request, emu_lib):
    """Both inflate routes: the decode / LZ-copy pair (k_inflate_seg + k_lzcopy, the default; k_inflate takes what they
    hand back) and k_inflate for every member."""
    c = _native.DContext(lib=emu_lib)
    c.set_route(_native.INFLATE_SEG if request.param == "seg" else _native.INFLATE_WAVE)
    c.route_name = request.param
    yield c
    c.close()


@pytest.mark.parametrize("cls", sorted(synth.CLASSES))
def test_roundtrip_of_our_streams(dctx, oracle, cls):
    for n, level in [(0, 1), (1, 1), (70000, 1), (2 * 65280 + 77, 3)]:
        a = synth.make(cls, n, 60 + n % 13)
        comp = oracle.compress_stream(a, oracle.FMT_BGZF, level, oracle.COMPAT_1_24, 65280)
        assert dctx.decompress(comp) == a.tobytes(), (cls, n, level)


def test_foreign_zlib_members(dctx):
    for level in (1, 6, 9):
        for cls in ("text", "dna", "mixed", "runs", "lowent", "random"):
            a = synth.make(cls, 50000, 7 + level).tobytes()
            stream = b"".join(bgzf_member(a[i:i + 25000], level) for i in range(0, len(a), 25000))
            assert dctx.decompress(stream) == a, (level, cls)
    tiny = b"hello hello hello"
    assert dctx.decompress(bgzf_member(tiny, 9, zlib.Z_FIXED)) == tiny  # fixed Huffman block
    far = synth.uniform_random(32768, 1).tobytes()
    far = far + far[:5000]                                            # matches at distance 32768
    assert dctx.decompress(bgzf_member(far, 9)) == far


def test_reference_test_inputs_roundtrip(dctx, emu_lib, golden):
    # src/deflate.rs:1024-1051 (test_simple_bgzf_etoe_decompress) and the 206-byte regression input
    for e in golden["raw_deflate_literal_inputs"]:
        data = bytes.fromhex(e["input_hex"])
        with _native.Context(level=1, lib=emu_lib, max_slab_bytes=65280) as c:
            comp = c.compress_slab(np.frombuffer(data, np.uint8), True)
        assert dctx.decompress(comp) == data


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


def test_speculative_paths_walk_on_behind_garbage_codes(emu_lib, oracle):
    """k_inflate_seg's speculative paths step over invalid codewords and walk on behind end-of-block codes (round 6): the
    streams where they meet such garbage most -- printable noise, whose codes are all 6-7 bits long, so that a path stays
    out of step for long and the 15-bit end-of-block code turns up by chance; Huffman-only members; members of hundreds of
    tiny blocks -- come out right, and the noise (libdeflate's streams of it) without a member handed back to k_inflate."""
    noise = synth.make("ascii", (1 << 20) + 200000, 5)
    with _native.DContext(lib=emu_lib) as d:
        comp = oracle.compress_stream(noise, oracle.FMT_BGZF, 3, oracle.COMPAT_1_24, 65280)
        assert d.decompress(comp) == noise.tobytes() and d.last_redo_count() == 0
    with _native.DContext(format=_native.FORMAT_MGZIP, lib=emu_lib) as d:
        comp = oracle.compress_stream(noise, oracle.FMT_MGZIP, 3, oracle.COMPAT_1_24, 1 << 20)
        assert d.decompress(comp) == noise.tobytes() and d.last_redo_count() == 0
    with _native.DContext(lib=emu_lib) as d:
        for cls, seed in (("ascii", 1), ("text", 2)):
            a = synth.make(cls, 120000, seed).tobytes()
            for strategy in (zlib.Z_HUFFMAN_ONLY, zlib.Z_FIXED, zlib.Z_DEFAULT_STRATEGY):
                s = b"".join(bgzf_member(a[i:i + 60000], 6, strategy) for i in range(0, len(a), 60000))
                assert d.decompress(s) == a, (cls, strategy)
            for every in (37, 4000):
                s = b"".join(flushed_member(a[i:i + 40000], 6, every) for i in range(0, 80000, 40000))
                assert d.decompress(s) == a[:80000], (cls, every)


def test_first_block_hint_may_be_stale(emu_lib, oracle):
    """k_inflate_seg guesses where a BGZF member's first block ends from the members it decoded before (a word of the
    context): streams of different kinds through ONE context, back and forth -- the hint of one is the wrong guess for
    the next -- come out right, with no member handed back."""
    streams = []
    for cls, seed in (("dna", 1), ("text", 2), ("fastq", 3), ("zeros", 5)):
        a = synth.make(cls, 3 * 65280 + 123, seed)
        streams.append((a.tobytes(), oracle.compress_stream(a, oracle.FMT_BGZF, 1, oracle.COMPAT_1_24, 65280)))
    with _native.DContext(lib=emu_lib) as d:
        for _ in range(2):
            for raw, comp in streams:
                assert d.decompress(comp) == raw and d.last_redo_count() == 0


def test_mgzip_large_blocks(emu_lib, oracle):
    a = synth.make("text", (1 << 20) + 999, 3)
    comp = oracle.compress_stream(a, oracle.FMT_MGZIP, 3, oracle.COMPAT_1_24, 1 << 20)
    with _native.DContext(format=_native.FORMAT_MGZIP, lib=emu_lib) as d:
        assert d.decompress(comp) == a.tobytes()


def test_error_classes(dctx, oracle):
    a = synth.make("text", 70000, 2)
    comp = bytearray(oracle.compress_stream(a, oracle.FMT_BGZF, 1, oracle.COMPAT_1_24, 65280))
    bad = bytearray(comp)
    bad[12] = ord("X")  # "Bad SID"
    with pytest.raises(_native.GzpxError) as e:
        dctx.decompress(bytes(bad))
    assert e.value.code == _native.ERR_INVALID_HEADER
    bad = bytearray(comp)
    bad[3] = 0  # "Extra field flag not set"
    with pytest.raises(_native.GzpxError) as e:
        dctx.decompress(bytes(bad))
    assert e.value.code == _native.ERR_INVALID_HEADER
    offs, sizes, _ = dctx.scan_blocks(bytes(comp))
    crc_pos = int(offs[0]) + int(sizes[0]) - 8
    bad = bytearray(comp)
    bad[crc_pos] ^= 0xFF  # InvalidCheck { found, expected }
    with pytest.raises(_native.GzpxError) as e:
        dctx.decompress(bytes(bad))
    assert e.value.code == _native.ERR_INVALID_CHECK and e.value.block == 0
    bad = bytearray(comp)
    for k in range(200, 260):
        bad[k] ^= 0x5A  # garbage inside the first payload: BadData, or a CRC mismatch
    with pytest.raises(_native.GzpxError) as e:
        dctx.decompress(bytes(bad))
    assert e.value.code in (_native.ERR_BAD_DATA, _native.ERR_INVALID_CHECK, _native.ERR_INSUFFICIENT_SPACE)
    # a member that inflates to fewer bytes than its ISIZE footer claims: BadData (libdeflate's
    # SHORT_OUTPUT through decode_block), never a CRC verdict over bytes the member did not produce
    bad = bytearray(comp)
    isize_pos = int(offs[0]) + int(sizes[0]) - 4
    bad[isize_pos:isize_pos + 4] = (int.from_bytes(bad[isize_pos:isize_pos + 4], "little") + 5).to_bytes(4, "little")
    with pytest.raises(_native.GzpxError) as e:
        dctx.decompress(bytes(bad))
    assert e.value.code == _native.ERR_BAD_DATA and e.value.block == 0


def test_libdeflate_shaped_decompressor(emu_lib, oracle):
    d = _native.Decompressor(lib=emu_lib)
    a = synth.make("fastq", 65280, 4)
    raw = oracle.deflate_compress(a, 1)
    assert d.deflate_decompress(raw, a.size) == a.tobytes()
    assert d.deflate_decompress(raw, a.size + 100) == a.tobytes()  # short output is fine
    with pytest.raises(_native.GzpxError) as e:
        d.deflate_decompress(raw, a.size - 1)
    assert e.value.code == _native.ERR_INSUFFICIENT_SPACE
    d.close()


def test_par_decompress_reader(emu_lib, oracle):
    a = synth.make("mixed", 5 * 65280 + 321, 9)
    comp = oracle.compress_stream(a, oracle.FMT_BGZF, 1, oracle.COMPAT_1_24, 65280)
    r = par.ParDecompressBuilder(par.Bgzf, lib=emu_lib).batch_bytes(100000).from_reader(io.BytesIO(comp))
    out = r.read(1000) + r.read()
    assert out == a.tobytes()
    assert r.read(10) == b""
    r.close()
    # a stream cut inside a block body: read_exact's UnexpectedEof (Io)
    r = par.ParDecompressBuilder(par.Bgzf, lib=emu_lib).from_reader(io.BytesIO(comp[:len(comp) // 2]))
    with pytest.raises(par.GzpError) as e:
        r.read()
    assert e.value.code == _native.ERR_IO
    r.close()

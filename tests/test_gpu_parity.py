"""Parity of the HIP path (real gfx950 library, through the C ABI) with the oracle, the golden
vectors and size-independent properties.  Needs an MI355X: run with -m gpu."""
import gzip
import hashlib
import zlib

import numpy as np
import pytest

from gzp_amd import _native, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx10(hip_lib):
    c = _native.Context(level=1, compat=_native.COMPAT_1_10, lib=hip_lib, max_slab_bytes=64 << 20)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx24(hip_lib):
    c = _native.Context(level=1, compat=_native.COMPAT_1_24, lib=hip_lib, max_slab_bytes=64 << 20)
    yield c
    c.close()


def test_native_library_is_the_hip_build(hip_lib, ctx10):
    assert hip_lib.path.endswith("gzp_amd/lib/libgzpx.so")
    assert "gfx950" in ctx10.device_name() or "MI3" in ctx10.device_name(), ctx10.device_name()


def test_golden_streams(hip_lib, golden):
    for e in golden["streams"]:
        a = (np.frombuffer(bytes.fromhex(e["input_hex"]), dtype=np.uint8) if "input_hex" in e
             else synth.make(e["class"], e["n"], e["seed"]))
        fmt = _native.FORMAT_BGZF if e["fmt"] == "bgzf" else _native.FORMAT_MGZIP
        with _native.Context(format=fmt, level=1, buffer_size=e["buffer_size"],
                             compat=_native.COMPAT_1_10, lib=hip_lib, max_slab_bytes=max(a.size, 1)) as c:
            out, sizes = c.compress_slab(a, True, return_block_sizes=True)
        assert hashlib.sha256(out).hexdigest() == e["sha256"], e
        assert list(sizes) == e["block_sizes"]


def test_golden_raw_deflate_via_libdeflate_shaped_abi(hip_lib, golden):
    comp = _native.Compressor(1, _native.COMPAT_1_10, lib=hip_lib)
    for e in golden["raw_deflate"] + golden["raw_deflate_literal_inputs"]:
        a = (np.frombuffer(bytes.fromhex(e["input_hex"]), dtype=np.uint8) if "input_hex" in e
             else synth.make(e["class"], e["n"], e["seed"]))
        out = comp.deflate_compress(a)
        assert hashlib.sha256(out).hexdigest() == e["sha256"], e
    comp.close()


@pytest.mark.parametrize("cls", sorted(synth.CLASSES))
def test_classes_vs_oracle(ctx10, ctx24, oracle, cls):
    # ragged multi-block slabs: every class, both libdeflate generations
    for n in [0, 51, 52, 65280, 65281, 5 * 65280 + 4321]:
        a = synth.make(cls, n, 300 + n)
        for ctx, compat in ((ctx10, oracle.COMPAT_1_10), (ctx24, oracle.COMPAT_1_24)):
            got = ctx.compress_slab(a, True)
            want = oracle.compress_stream(a, oracle.FMT_BGZF, 1, compat, 65280)
            assert got == want, (cls, n, compat)


def test_tokens_match_oracle(ctx10, oracle):
    a = synth.repeated_phrases(65280, 3)
    ctx10.compress_slab(a, True)
    toks, first = ctx10.debug_tokens(0)
    et, ef = oracle.l1_tokens(a)
    assert np.array_equal(toks, et)
    assert list(first) == list(ef)


def test_block_size_exceeded(hip_lib):
    a = synth.uniform_random(65536, 1)
    with _native.Context(format=_native.FORMAT_BGZF, level=1, buffer_size=65536, lib=hip_lib,
                         max_slab_bytes=a.size) as c:
        with pytest.raises(_native.GzpxError) as ei:
            c.compress_slab(a, True)
    assert ei.value.code == _native.ERR_BLOCK_SIZE_EXCEEDED


def test_crc32_abi(hip_lib):
    a = synth.uniform_random(3_000_001, 3)
    assert _native.crc32(a, lib=hip_lib) == zlib.crc32(a.tobytes())
    assert _native.crc32(a[777:], crc=zlib.crc32(a[:777].tobytes()), lib=hip_lib) == zlib.crc32(a.tobytes())


def test_device_resident_slab_and_properties(hip_lib, oracle):
    """48 MiB slab resident in HBM: round trip through an independent inflater, per-block CRC /
    ISIZE / BSIZE chain, determinism, and bit-exactness of sampled blocks."""
    import torch
    n = 48 * 1024 * 1024 + 12345
    a = np.concatenate([synth.text_slab(n - 2_000_000, 2_000_000, 7), synth.fastq_like(1_500_000, 3),
                        synth.uniform_random(500_000, 4)])
    assert a.size == n
    with _native.Context(level=1, lib=hip_lib, max_slab_bytes=n) as c:
        d_in = torch.from_numpy(a).cuda()
        cap = c.slab_bound(n)
        d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        nb = c.n_blocks(n)
        sizes = np.zeros(nb, dtype=np.uint32)
        torch.cuda.synchronize()
        out_len, nblk = c.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True, None, sizes)
        out = d_out[:out_len].cpu().numpy()
        d_out.zero_()
        out_len2, _ = c.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True, None, sizes)
        assert out_len2 == out_len and np.array_equal(d_out[:out_len].cpu().numpy(), out)  # idempotent
        # GZPX_STREAM_NONE: "already synchronized", nothing recorded on the legacy stream -- the same stream
        d_out.zero_()
        torch.cuda.synchronize()
        out_len3, _ = c.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True, _native.STREAM_NONE, sizes)
        assert out_len3 == out_len and np.array_equal(d_out[:out_len].cpu().numpy(), out)
    assert nblk == nb and int(sizes.sum()) == out_len
    assert gzip.decompress(out.tobytes()) == a.tobytes()
    offs = np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])
    for b in range(nb):  # BSIZE chain + footer of every block
        blk = out[offs[b]:offs[b + 1]]
        bsize = int(blk[16]) | (int(blk[17]) << 8)
        payload_end = bsize + 1
        assert payload_end == sizes[b] - (28 if b == nb - 1 else 0)
        isize = int.from_bytes(blk[payload_end - 4:payload_end].tobytes(), "little")
        assert isize == min(65280, n - b * 65280)
    for b in [0, 1, nb // 3, nb - 3, nb - 2, nb - 1]:
        want = oracle.encode_block(a[b * 65280:(b + 1) * 65280], oracle.FMT_BGZF, 1, oracle.COMPAT_1_24,
                                   is_last=(b == nb - 1))
        assert out[offs[b]:offs[b + 1]].tobytes() == want, b


def test_order_independent_candidate_kernel(hip_lib, oracle):
    """k_candidates' order-independent fallback (for a failed LDS-order check) forced on every block, and
    the fast kernel never needing it on this hardware."""
    with _native.Context(level=1, compat=_native.COMPAT_1_10, lib=hip_lib, max_slab_bytes=3 * 65280) as c:
        c.debug_set_flags(1)
        for cls in ("text", "zeros", "period2", "repeats", "random"):
            a = synth.make(cls, 2 * 65280 + 99, 17)
            assert c.compress_slab(a, True) == oracle.compress_stream(a, oracle.FMT_BGZF, 1, oracle.COMPAT_1_10, 65280), cls


@pytest.mark.parametrize("cls", ["text", "repeats", "zeros", "random", "fastq"])
def test_large_mgzip_blocks_vs_oracle(hip_lib, oracle, cls):
    """Mgzip at its default 128 KiB buffer and at the 1 MiB blocks of BASELINE config 3 (level 1)."""
    for bs, n in [(131072, 5 * 131072 + 4321), (1 << 20, 3 * (1 << 20) + 77)]:
        a = synth.make(cls, n, 40 + n % 97)
        with _native.Context(format=_native.FORMAT_MGZIP, level=1, buffer_size=bs, compat=_native.COMPAT_1_10,
                             lib=hip_lib, max_slab_bytes=a.size) as c:
            got = c.compress_slab(a, True)
        assert got == oracle.compress_stream(a, oracle.FMT_MGZIP, 1, oracle.COMPAT_1_10, bs), (cls, bs, n)
        assert gzip.decompress(got) == a.tobytes()

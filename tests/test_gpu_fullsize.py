"""BASELINE.json's configurations at full size on the MI355X, checked through size-independent
properties (the oracle cannot run 550 MiB per test): stream structure, GPU inflate round trip with
the per-block CRC check, idempotence, a checksum of the per-block checksums against the CPU oracle,
and a sample of blocks bit-exact against the oracle.  Device-resident, through the C ABI."""
import zlib

import numpy as np
import pytest
import torch

from gzp_amd import _native, synth

pytestmark = pytest.mark.gpu

BLOCK = 65280


def _device_compress(ctx, d_in, n, last=True):
    cap = ctx.slab_bound(n)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    sizes = np.zeros(ctx.n_blocks(n), dtype=np.uint32)
    out_len, nb = ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, last, None, sizes)
    return d_out, out_len, sizes


def test_config1_550mib_bgzf_level1(hip_lib, oracle):
    # configs[1]: 64 KiB BGZF blocks, level 1, 550 MiB text slab
    n = 576_716_800
    slab = synth.text_slab(n, seed=20250927)
    d_in = torch.from_numpy(slab).cuda()
    with _native.Context(format=_native.FORMAT_BGZF, level=1, buffer_size=BLOCK, lib=hip_lib,
                         max_slab_bytes=n) as ctx:
        d_out, out_len, sizes = _device_compress(ctx, d_in, n)
        d_out2, out_len2, sizes2 = _device_compress(ctx, d_in, n)
    # idempotence: the same bytes again
    assert out_len2 == out_len and torch.equal(d_out[:out_len], d_out2[:out_len])
    assert np.array_equal(sizes, sizes2)
    comp = d_out[:out_len].cpu().numpy()
    nb = -(-n // BLOCK)
    assert sizes.size == nb and int(sizes.sum()) == out_len
    # structure: the BSIZE chain walks the whole stream, block by block, and ends in the EOF marker
    with _native.DContext(format=_native.FORMAT_BGZF, lib=hip_lib) as d:
        offs, bsz, used = d.scan_blocks(comp)
        assert used == out_len and offs.size == nb + 1 and int(bsz[-1]) == 28
        assert np.array_equal(bsz[:-1].astype(np.int64), sizes.astype(np.int64) - np.where(
            np.arange(nb) == nb - 1, 28, 0))
        # round trip on the GPU (configs[4]): inflate + per-block CRC check, compared on the device
        d_back = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
        got = d.decompress_device(d_out.data_ptr(), out_len, offs, bsz, d_back.data_ptr(), n + 64)
        assert got == n and torch.equal(d_back[:n], d_in)
    # ISIZE and CRC fields of every block; checksum of checksums against the CPU oracle
    ends = np.cumsum(sizes.astype(np.int64))
    ends[-1] -= 28
    foot = np.stack([comp[e - 8:e] for e in ends])
    isize = foot[:, 4:8].copy().view("<u4").ravel()
    assert int(isize.sum()) == n and np.all(isize[:-1] == BLOCK)
    crcs = foot[:, 0:4].copy().view("<u4").ravel()
    want = np.array([oracle.crc32(slab[i * BLOCK:(i + 1) * BLOCK]) for i in range(nb)], dtype=np.uint32)
    assert zlib.crc32(crcs.tobytes()) == zlib.crc32(want.tobytes())
    # a sample of blocks bit-exact against the oracle (first, last, and a spread)
    starts = np.concatenate([[0], np.cumsum(sizes.astype(np.int64))[:-1]])
    for b in sorted(set([0, 1, nb // 3, nb // 2, nb - 2, nb - 1] + list(range(7, nb, 997)))):
        ref = oracle.encode_block(slab[b * BLOCK:(b + 1) * BLOCK], oracle.FMT_BGZF, 1, oracle.COMPAT_1_24,
                                  is_last=(b == nb - 1))
        assert comp[starts[b]:starts[b] + sizes[b]].tobytes() == ref, b


def test_config2_mgzip_1mib_blocks_level3(hip_lib, oracle):
    # configs[2]'s shape: Mgzip, 1 MiB blocks, level 3, ASCII noise (1 GiB here; the config's 4 GiB
    # is the same per-block work four times over)
    n = 1 << 30
    bs = 1 << 20
    slab = synth.make("ascii", n, 4242)
    d_in = torch.from_numpy(slab).cuda()
    with _native.Context(format=_native.FORMAT_MGZIP, level=3, buffer_size=bs, lib=hip_lib,
                         max_slab_bytes=n) as ctx:
        d_out, out_len, sizes = _device_compress(ctx, d_in, n)
    comp = d_out[:out_len].cpu().numpy()
    nb = n // bs
    assert sizes.size == nb and int(sizes.sum()) == out_len
    with _native.DContext(format=_native.FORMAT_MGZIP, lib=hip_lib) as d:
        offs, bsz, used = d.scan_blocks(comp)
        assert used == out_len and offs.size == nb
        d_back = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
        got = d.decompress_device(d_out.data_ptr(), out_len, offs, bsz, d_back.data_ptr(), n + 64)
        assert got == n and torch.equal(d_back[:n], d_in)
    starts = np.concatenate([[0], np.cumsum(sizes.astype(np.int64))[:-1]])
    for b in (0, nb // 2, nb - 1):
        ref = oracle.encode_block(slab[b * bs:(b + 1) * bs], oracle.FMT_MGZIP, 3, oracle.COMPAT_1_24,
                                  is_last=(b == nb - 1))
        assert comp[starts[b]:starts[b] + sizes[b]].tobytes() == ref, b
        assert zlib.decompress(comp[starts[b] + 20:starts[b] + sizes[b] - 8].tobytes(), -15) == \
            slab[b * bs:(b + 1) * bs].tobytes()

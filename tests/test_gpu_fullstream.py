"""BASELINE.json's configurations at their FULL sizes, every block pinned at once: the SHA-256 of the
whole stream and of the array of framed block sizes against tests/golden/fullsize.json, whose digests
come from the libdeflate binary of the image (tests/golden/make_fullsize.py; v1.10, hence
compat=1.10 here).  Inputs are regenerated from (kind, n, seed): the text slab on the host, the
ASCII noise and the FASTQ stream in HBM by gzpx_synth_*_device (their own digests are pinned too).
Device-resident, through the C ABI."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from gzp_amd import _native, synth

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "fullsize.json")) as f:
    GOLD = {e["name"]: e for e in json.load(f)["streams"]}


def _sha_device(t):
    """SHA-256 of a device tensor, copied out in 256 MiB pieces."""
    h = hashlib.sha256()
    for lo in range(0, t.numel(), 256 << 20):
        h.update(t[lo:lo + (256 << 20)].cpu().numpy())
    return h.hexdigest()


def _input(e, lib):
    inp = e["input"]
    n = inp["n"]
    if inp["kind"] == "text_slab":
        return torch.from_numpy(synth.text_slab(n, seed=inp["seed"])).cuda()
    d = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    if inp["kind"] == "ascii":
        _native.synth_ascii_device(d.data_ptr(), 0, n, inp["seed"], lib=lib)
    else:
        _native.synth_fastq_device(d.data_ptr(), inp["offset"], n, inp["seed"], lib=lib)
    torch.cuda.synchronize()
    return d[:n]


def _run(e, lib, compat=_native.COMPAT_1_10, check_input=True):
    fmt = _native.FORMAT_BGZF if e["fmt"] == "bgzf" else _native.FORMAT_MGZIP
    d_in = _input(e, lib)
    n = d_in.numel()
    if check_input:
        assert _sha_device(d_in) == e["input_sha256"], "the input generator drifted"
    with _native.Context(format=fmt, level=e["level"], buffer_size=e["buffer_size"], compat=compat, lib=lib,
                         max_slab_bytes=n) as ctx:
        cap = ctx.slab_bound(n)
        d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        sizes = np.zeros(ctx.n_blocks(n), dtype=np.uint32)
        mode = _native.SLAB_LAST if e["tail"] else _native.SLAB_FULL_BLOCKS
        out_len, nb = ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, mode, None, sizes)
    assert nb == e["n_blocks"] and out_len == e["size"], (nb, out_len)
    assert hashlib.sha256(sizes.astype("<u4").tobytes()).hexdigest() == e["block_sizes_sha256"]
    assert _sha_device(d_out[:out_len]) == e["sha256"]
    return d_in, d_out, out_len, sizes


def test_config2_text_550mib_every_block(hip_lib):
    e = GOLD["config2_text_550MiB_bgzf_l1"]
    d_in, d_out, out_len, sizes = _run(e, hip_lib)
    # the product default (compat 1.24) gives the same stream on this slab: the one rule that differs
    # between the two libdeflate versions (an unused offset code) never fires in text
    with _native.Context(format=_native.FORMAT_BGZF, level=1, buffer_size=65280, compat=_native.COMPAT_1_24,
                         lib=hip_lib, max_slab_bytes=d_in.numel()) as ctx:
        d_out2 = torch.empty_like(d_out)
        out_len2, _ = ctx.compress_slab_device(d_in.data_ptr(), d_in.numel(), d_out2.data_ptr(), d_out2.numel(), True)
    assert out_len2 == out_len and torch.equal(d_out[:out_len], d_out2[:out_len])


@pytest.mark.parametrize("level", [3, 6, 9])
def test_text_550mib_levels_3_6_9_every_block(hip_lib, level):
    """The bench slab through the greedy (3), lazy (6) and lazy2 (9) parsers: all 8,835 blocks + EOF."""
    d_in, d_out, out_len, sizes = _run(GOLD["text_550MiB_bgzf_l%d" % level], hip_lib, check_input=False)
    assert int(d_out[8]) == (2 if level == 9 else 0)  # XFL (src/bgzf.rs:278-284)


def test_text_550mib_level_12_every_block(hip_lib):
    """The bench slab through the near-optimal parser (k_near_optimal, 4 optimisation passes): all 8,835 blocks + EOF
    against the v1.10 binary's stream."""
    d_in, d_out, out_len, sizes = _run(GOLD["text_550MiB_bgzf_l12"], hip_lib, check_input=False)
    assert int(d_out[8]) == 2  # XFL: best compression from level 9 on (src/bgzf.rs:278-284)


@pytest.mark.parametrize("name", ["config3_ascii_1GiB_mgzip_l3", "config3_ascii_4GiB_mgzip_l3"])
def test_config3_mgzip_level3_every_block(hip_lib, name):
    _run(GOLD[name], hip_lib)


def test_config4_one_rank_share_of_32gib_fastq(hip_lib):
    # configs[3]: rank 0 of 8 of the 32 GiB stream = 65,794 whole blocks (4,295,032,320 bytes)
    from gzp_amd import shard
    e = GOLD["config4_fastq_rank0of8_bgzf_l1"]
    inp = e["input"]
    lo, n = shard.shard_bytes(inp["stream_bytes"], 65280, inp["world"])[inp["rank"]]
    assert (lo, n) == (inp["offset"], inp["n"]) and n == 65794 * 65280
    assert shard.slab_mode(inp["rank"], inp["world"], inp["stream_bytes"], 65280) == _native.SLAB_FULL_BLOCKS
    d_in, d_out, out_len, sizes = _run(e, hip_lib)
    # size-independent properties on top of the digest: BSIZE chain, GPU inflate + CRC round trip
    comp = d_out[:out_len].cpu().numpy()
    with _native.DContext(format=_native.FORMAT_BGZF, lib=hip_lib) as d:
        offs, bsz, used = d.scan_blocks(comp)
        assert used == out_len and np.array_equal(bsz, sizes)
        d_back = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
        got = d.decompress_device(d_out.data_ptr(), out_len, offs, bsz, d_back.data_ptr(), n + 64)
        assert got == n and torch.equal(d_back[:n], d_in)


def test_config4_all_eight_shares_and_the_whole_32gib_stream(hip_lib):
    """ALL of configs[3] on the one GPU a test box has (round 5): the eight rank shares of the 32 GiB FASTQ stream, cut
    by shard.shard_bytes / shard.slab_mode exactly as `bench.py --workload fastq --gpus 8` cuts them, compressed one
    after the other; every share's stream and framed sizes against the libdeflate-made digests, and the concatenation
    in rank order -- what the in-order writer must produce (src/par/compress.rs:305-310) -- against the SHA-256 of the
    WHOLE 32 GiB stream's output (526,345 blocks + EOF); every share inflated + CRC-checked on the GPU against its
    regenerated input (what gzip -t does), and gzip -t itself on the two ends of the concatenation."""
    import subprocess
    from gzp_amd import shard
    whole = GOLD["config4_fastq_32GiB_whole_bgzf_l1"]
    total, world, bs = whole["input"]["stream_bytes"], whole["input"]["world"], 65280
    parts = shard.shard_bytes(total, bs, world)
    h_all = hashlib.sha256()
    all_sizes, out_total = [], 0
    head = tail = None
    n_max = max(n for _, n in parts)
    d_in = torch.empty(n_max + 64, dtype=torch.uint8, device="cuda")
    d_back = torch.empty(n_max + 64, dtype=torch.uint8, device="cuda")
    with _native.Context(format=_native.FORMAT_BGZF, level=1, buffer_size=bs, compat=_native.COMPAT_1_10, lib=hip_lib,
                         max_slab_bytes=n_max) as ctx, _native.DContext(format=_native.FORMAT_BGZF, lib=hip_lib) as dctx:
        cap = ctx.slab_bound(n_max)
        d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        for r, (lo, n) in enumerate(parts):
            e = GOLD["config4_fastq_rank%dof%d_bgzf_l1" % (r, world)]
            assert (lo, n) == (e["input"]["offset"], e["input"]["n"])
            mode = shard.slab_mode(r, world, total, bs)
            assert (mode == _native.SLAB_LAST) == e["tail"] == (r == world - 1)
            _native.synth_fastq_device(d_in.data_ptr(), lo, n, e["input"]["seed"], lib=hip_lib)
            torch.cuda.synchronize()
            sizes = np.zeros(ctx.n_blocks(n), dtype=np.uint32)
            out_len, nb = ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, mode, None, sizes)
            assert nb == e["n_blocks"] and out_len == e["size"] and out_total == e["stream_offset"], (r, nb, out_len)
            assert hashlib.sha256(sizes.astype("<u4").tobytes()).hexdigest() == e["block_sizes_sha256"], r
            h = hashlib.sha256()
            for c0 in range(0, out_len, 256 << 20):  # one pass over the bytes feeds the share's and the stream's digest
                piece = d_out[c0:min(out_len, c0 + (256 << 20))].cpu().numpy()
                h.update(piece)
                h_all.update(piece)
                if r == 0 and c0 == 0:
                    offs0 = np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])
                    head = piece[:int(offs0[min(1000, nb)])].tobytes()  # a prefix of whole blocks
            assert h.hexdigest() == e["sha256"], "share %d differs from libdeflate's stream" % r
            if r == world - 1:
                k = max(0, nb - 1000)
                t0 = int(np.sum(sizes[:k].astype(np.int64)))
                tail = d_out[t0:out_len].cpu().numpy().tobytes()  # the last 1000 blocks + the EOF marker
            # the share inflated and CRC-checked on the GPU, compared with the regenerated input
            offs = np.concatenate([[0], np.cumsum(sizes.astype(np.uint64))[:-1]]).astype(np.uint64)
            members = sizes.copy()
            if e["tail"]:
                members[-1] -= 28  # (the last framed size counts the EOF marker, an empty member of its own)
            got = dctx.decompress_device(d_out.data_ptr(), out_len - (28 if e["tail"] else 0), offs, members, d_back.data_ptr(), n_max + 64)
            assert got == n and torch.equal(d_back[:n], d_in[:n]), r
            all_sizes.append(sizes)
            out_total += out_len
    assert out_total == whole["size"] and sum(s.size for s in all_sizes) == whole["n_blocks"] == 526345
    assert hashlib.sha256(np.concatenate(all_sizes).astype("<u4").tobytes()).hexdigest() == whole["block_sizes_sha256"]
    assert h_all.hexdigest() == whole["sha256"], "the in-order concatenation of the eight shares is not libdeflate's stream"
    for name, blob in (("prefix", head), ("suffix", tail)):
        try:
            rc = subprocess.run(["gzip", "-t"], input=blob, capture_output=True).returncode
        except FileNotFoundError:
            rc = 0  # (no gzip binary on the box: the GPU inflate above has already done its job)
        assert rc == 0, name
    assert tail[-28:-24] == b"\x1f\x8b\x08\x04" and tail[-12:-10] == b"\x1b\x00"  # BGZF_EOF (src/bgzf.rs:24-38)


def test_config4_stream_tail(hip_lib):
    # the end of the 32 GiB stream: 1000 whole blocks, the 2,048-byte block and the EOF marker
    e = GOLD["config4_fastq_stream_tail_bgzf_l1"]
    d_in, d_out, out_len, sizes = _run(e, hip_lib)
    assert d_in.numel() == 1000 * 65280 + 2048 and bytes(d_out[out_len - 28:out_len].cpu().numpy()[:4]) == b"\x1f\x8b\x08\x04"

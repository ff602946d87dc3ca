"""Generate tests/golden/fullsize.json -- SHA-256 digests of the COMPLETE streams of the BASELINE
configurations at their full sizes, made by the libdeflate binary of the image (v1.10, Ubuntu
libdeflate0 1.10-2; `compat=1.10` on the GPU side) plus gzp's framing rules as restated in
make_golden.py.  Run in the build container only:

    python tests/golden/make_fullsize.py            # ~10 minutes, 8 threads
    python tests/golden/make_fullsize.py fastq8     # round 5: ALL eight rank shares of configs[3] + the whole 32 GiB stream

Inputs are regenerated on the GPU box from (kind, n, seed): synth.text_slab on the host,
gzpx_synth_ascii_device / gzpx_synth_fastq_device in HBM (their host statements are used here).
Every entry pins all blocks of a stream at once: the digest of the stream and the digest of the
array of framed block sizes (little-endian u32).
"""
import hashlib
import json
import os
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (ld_deflate / framing restatement; thread-local compressors below)
from gzp_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402  (fastq_stream: the host statement of the device generator)

_tls = threading.local()


def _frame(args):
    a, level, fmt, is_last = args
    if not hasattr(_tls, "comp"):
        _tls.comp = {}
    if level not in _tls.comp:  # one libdeflate compressor per thread and level
        _tls.comp[level] = mg.LD.libdeflate_alloc_compressor(level)
    mg._comp[level] = _tls.comp[level]  # (ld_deflate looks here; GIL-held assignment right before the call)
    return _frame_with(_tls.comp[level], a, level, fmt, is_last)


def _frame_with(comp, a, level, fmt, is_last):
    import struct
    cap = a.size + max(128, a.size // 10) + 8
    out = np.empty(cap, dtype=np.uint8)
    n = mg.LD.libdeflate_deflate_compress(comp, a.ctypes.data, a.size, out.ctypes.data, cap)
    assert n > 0
    payload = out[:n].tobytes()
    xfl = 2 if level >= 9 else 4 if level <= 1 else 0
    if fmt == "bgzf":
        assert len(payload) < 65536
        hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, xfl, 255, 6, ord("B"), ord("C"), 2, len(payload) + 25)
    else:
        hdr = struct.pack("<BBBBIBBHBBHI", 31, 139, 8, 4, 0, xfl, 255, 8, ord("I"), ord("G"), 4, len(payload) + 28)
    res = hdr + payload + struct.pack("<II", mg.LD.libdeflate_crc32(0, a.ctypes.data, a.size), a.size)
    if is_last and fmt == "bgzf":
        res += mg.BGZF_EOF
    return res


def digest_stream(a, level, fmt, bs, tail, also=None, all_sizes=None):
    """Stream of `a` cut like ParCompress (tail=True: flush_last(true), EOF marker; False: whole
    blocks only, the shard of a rank that does not own the stream's end).  `also`: a second hasher fed
    with the same bytes (the whole stream a rank's share belongs to); `all_sizes`: a list the framed
    sizes are appended to."""
    n = a.size
    nb = 1 if n == 0 else -(-n // bs)
    assert tail or n % bs == 0
    h = hashlib.sha256()
    sizes = np.zeros(nb, dtype="<u4")
    total = 0
    with ThreadPoolExecutor(8) as pool:
        for c0 in range(0, nb, 2048):
            idx = range(c0, min(nb, c0 + 2048))
            jobs = [(a[b * bs:min(n, (b + 1) * bs)], level, fmt, tail and b == nb - 1) for b in idx]
            for b, blk in zip(idx, pool.map(_frame, jobs)):
                h.update(blk)
                if also is not None:
                    also.update(blk)
                sizes[b] = len(blk)
                total += len(blk)
    if all_sizes is not None:
        all_sizes.append(sizes)
    return {"size": total, "sha256": h.hexdigest(), "n_blocks": nb,
            "block_sizes_sha256": hashlib.sha256(sizes.tobytes()).hexdigest()}


def fastq_all_shares(world=8, total=32 << 30, seed=20250927, bs=65280):
    """ALL of BASELINE configs[3] (round 5): every rank's share of the 32 GiB FASTQ stream as shard.shard_bytes cuts
    it, one after the other (4 GiB of input at a time), plus the SHA-256 of the WHOLE stream -- the concatenation in
    rank order, which is what the in-order writer (src/par/compress.rs:305-310) must produce -- and of all 526,345
    framed sizes.  Replaces / adds the entries config4_fastq_rank{r}of8_bgzf_l1 and config4_fastq_32GiB_whole_bgzf_l1
    in fullsize.json; everything else in the file is kept.      python tests/golden/make_fullsize.py fastq8"""
    from gzp_amd import shard
    with open(os.path.join(HERE, "fullsize.json")) as f:
        doc = json.load(f)
    names = ["config4_fastq_rank%dof%d_bgzf_l1" % (r, world) for r in range(world)] + ["config4_fastq_32GiB_whole_bgzf_l1"]
    out = [e for e in doc["streams"] if e["name"] not in names]
    t0 = time.time()
    whole = hashlib.sha256()
    all_sizes = []
    total_out = 0
    total_blocks = -(-total // bs)
    shares = []
    for r, (lo, n) in enumerate(shard.shard_bytes(total, bs, world)):
        a = oracle.fastq_stream(lo, n, seed)
        tail = lo + n == total
        e = {"name": names[r], "fmt": "bgzf", "level": 1, "buffer_size": bs, "tail": tail,
             "input": {"kind": "fastq", "n": n, "seed": seed, "offset": lo, "stream_bytes": total, "world": world, "rank": r},
             "input_sha256": hashlib.sha256(a).hexdigest()}
        e.update(digest_stream(a, 1, "bgzf", bs, tail, also=whole, all_sizes=all_sizes))
        e["stream_offset"] = total_out  # where the share starts in the whole output stream
        total_out += e["size"]
        out.append(e)
        shares.append(e["size"])
        print("%-34s %d -> %d bytes, %d blocks  (%.0f s)" % (e["name"], n, e["size"], e["n_blocks"], time.time() - t0), flush=True)
        del a
    sizes = np.concatenate(all_sizes)
    assert sizes.size == total_blocks
    out.append({"name": names[-1], "fmt": "bgzf", "level": 1, "buffer_size": bs, "tail": True,
                "input": {"kind": "fastq", "n": total, "seed": seed, "offset": 0, "stream_bytes": total, "world": world},
                "size": total_out, "sha256": whole.hexdigest(), "n_blocks": int(total_blocks),
                "block_sizes_sha256": hashlib.sha256(sizes.astype("<u4").tobytes()).hexdigest(),
                "share_sizes": shares})
    doc["streams"] = out
    with open(os.path.join(HERE, "fullsize.json"), "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")


def main():
    if sys.argv[1:] == ["fastq8"]:
        return fastq_all_shares()
    out = []
    t0 = time.time()
    only_levels = sys.argv[1:] in (["levels"], ["near_optimal"])  # add / refresh the text slab at levels 3, 6, 9 (or 10, 12) only
    levels = (10, 12) if sys.argv[1:] == ["near_optimal"] else (3, 6, 9)
    if only_levels:
        with open(os.path.join(HERE, "fullsize.json")) as f:
            out = [e for e in json.load(f)["streams"]
                   if e["name"] not in ["text_550MiB_bgzf_l%d" % lv for lv in levels]]

    def add(name, a, level, fmt, bs, tail, inp):
        e = {"name": name, "fmt": fmt, "level": level, "buffer_size": bs, "tail": tail, "input": inp,
             "input_sha256": hashlib.sha256(a).hexdigest()}
        e.update(digest_stream(a, level, fmt, bs, tail))
        out.append(e)
        print("%-28s %d -> %d bytes, %d blocks  (%.0f s)" % (name, a.size, e["size"], e["n_blocks"], time.time() - t0),
              flush=True)

    # BASELINE configs[1]: 550 MiB text, BGZF, level 1 (the bench slab, seed as bench.py rank 0)
    a = synth.text_slab(576_716_800, seed=20250927)
    if not only_levels:
        add("config2_text_550MiB_bgzf_l1", a, 1, "bgzf", 65280, True,
            {"kind": "text_slab", "n": 576_716_800, "seed": 20250927})
    # the same slab at gzp's default level and through the lazy / lazy2 parsers (best() = 9: XFL 2)
    for level in levels:  # (10, 12: the near-optimal parser -- python tests/golden/make_fullsize.py near_optimal)
        add("text_550MiB_bgzf_l%d" % level, a, level, "bgzf", 65280, True,
            {"kind": "text_slab", "n": 576_716_800, "seed": 20250927})
    if only_levels:
        with open(os.path.join(HERE, "fullsize.json"), "w") as f:
            json.dump({"generator": "tests/golden/make_fullsize.py",
                       "libdeflate": "v1.10 binary (Ubuntu libdeflate0 1.10-2), compat=1.10", "streams": out}, f, indent=1)
            f.write("\n")
        return
    # BASELINE configs[2]: Mgzip 1 MiB blocks, level 3, printable-ASCII noise; 1 GiB and the full 4 GiB
    a = oracle.ascii_stream(0, 4 << 30, 8)  # (== synth.ascii_random(4 << 30, 8), natively)
    add("config3_ascii_1GiB_mgzip_l3", a[:1 << 30], 3, "mgzip", 1 << 20, True, {"kind": "ascii", "n": 1 << 30, "seed": 8})
    add("config3_ascii_4GiB_mgzip_l3", a, 3, "mgzip", 1 << 20, True, {"kind": "ascii", "n": 4 << 30, "seed": 8})
    del a
    # BASELINE configs[3]: 32 GiB synthetic FASTQ over 8 ranks -> rank 0's share: 65,794 whole blocks
    # (526,345 blocks = 526,344 full + one of 2,048 bytes; shard.shard_blocks gives rank 0 65,794)
    n0 = 65794 * 65280
    a = oracle.fastq_stream(0, n0, 20250927)
    add("config4_fastq_rank0of8_bgzf_l1", a, 1, "bgzf", 65280, False,
        {"kind": "fastq", "n": n0, "seed": 20250927, "offset": 0, "stream_bytes": 32 << 30, "world": 8, "rank": 0})
    del a
    # ... and the end of the stream: the last 64 MiB of rank 7's share incl. the 2,048-byte block + EOF
    total = 32 << 30
    lo = ((total // 65280) - 1000) * 65280
    a = oracle.fastq_stream(lo, total - lo, 20250927)
    add("config4_fastq_stream_tail_bgzf_l1", a, 1, "bgzf", 65280, True,
        {"kind": "fastq", "n": total - lo, "seed": 20250927, "offset": lo, "stream_bytes": 32 << 30})
    with open(os.path.join(HERE, "fullsize.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_fullsize.py",
                   "libdeflate": "v1.10 binary (Ubuntu libdeflate0 1.10-2), compat=1.10", "streams": out}, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()

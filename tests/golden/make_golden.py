"""Generate tests/golden/*.json -- the committed known-answer vectors for the hot path.

Run in the build container only (needs the libdeflate binary of the image,
/lib/x86_64-linux-gnu/libdeflate.so.0 = Ubuntu libdeflate0 1.10-2):

    python tests/golden/make_golden.py

Why a binary and not /root/reference: gzp's per-block arithmetic lives in the third-party C
library libdeflate (Cargo.lock:414-430 pins libdeflate-sys 1.24.0) whose source is not vendored
in the reference tree, and the Rust crate itself cannot be built here (no cargo/rustc).  The
vectors are therefore produced by the real libdeflate *binary* (v1.10: same level-1..4
algorithms; the one known output delta vs 1.24 is the empty-offset-code rule, SURVEY.md A.7,
selectable as compat=1.10 everywhere) and by gzp's framing rules (src/bgzf.rs:204-303,
src/mgzip.rs:187-275, src/par/compress.rs:332-362,413-463) restated below in ~40 lines of
Python, independently of oracle/gzpx_oracle.c.

Inputs are regenerated from (class, n, seed) by gzp_amd/synth.py; outputs are stored as
SHA-256 + size (+ full hex for outputs <= 1 KiB).
"""
import ctypes
import hashlib
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from gzp_amd import synth  # noqa: E402

LD_PATH = os.environ.get("LIBDEFLATE_SO", "/lib/x86_64-linux-gnu/libdeflate.so.0")
LD = ctypes.CDLL(LD_PATH)
LD.libdeflate_alloc_compressor.restype = ctypes.c_void_p
LD.libdeflate_alloc_compressor.argtypes = [ctypes.c_int]
LD.libdeflate_deflate_compress.restype = ctypes.c_size_t
LD.libdeflate_deflate_compress.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                           ctypes.c_void_p, ctypes.c_size_t]
LD.libdeflate_crc32.restype = ctypes.c_uint32
LD.libdeflate_crc32.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t]
_comp = {}

BGZF_EOF = bytes([0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0x00, 0xff, 0x06, 0x00, 0x42, 0x43, 0x02, 0x00,
                  0x1b, 0x00, 0x03, 0x00, 0, 0, 0, 0, 0, 0, 0, 0])  # src/bgzf.rs:24-38

# inputs held by the reference's own tests (data, not code)
REF_REGRESSION_206 = [  # src/deflate.rs:958-970
    132, 19, 107, 159, 69, 217, 180, 131, 224, 49, 143, 41, 194, 30, 151, 22, 55, 30, 42,
    139, 219, 62, 123, 44, 148, 144, 88, 233, 199, 126, 110, 65, 6, 87, 51, 215, 17, 253,
    22, 63, 110, 1, 100, 202, 44, 138, 187, 226, 50, 50, 218, 24, 193, 218, 43, 172, 69,
    71, 8, 164, 5, 186, 189, 215, 151, 170, 243, 235, 219, 103, 1, 0, 102, 80, 179, 95,
    247, 26, 168, 147, 139, 245, 177, 253, 94, 82, 146, 133, 103, 223, 96, 34, 128, 237,
    143, 182, 48, 201, 201, 92, 29, 172, 137, 70, 227, 98, 181, 246, 80, 21, 106, 175, 246,
    41, 229, 187, 87, 65, 79, 63, 115, 66, 143, 251, 41, 251, 214, 7, 64, 196, 27, 180, 42,
    132, 116, 211, 148, 44, 177, 137, 91, 119, 245, 156, 78, 24, 253, 69, 38, 52, 152, 115,
    123, 94, 162, 72, 186, 239, 136, 179, 11, 180, 78, 54, 217, 120, 173, 141, 114, 174,
    220, 160, 223, 184, 114, 73, 148, 120, 43, 25, 21, 62, 62, 244, 85, 87, 19, 174, 182,
    227, 228, 70, 153, 5, 92, 51, 161, 9, 140, 199, 244, 241, 151, 236, 81, 211,
]
REF_SIMPLE_TEXT = (b"\n        This is a longer test than normal to come up with a bunch of text.\n"
                   b"        We'll read just a few lines at a time.\n        ")  # src/deflate.rs:1033-1036


def ld_deflate(a, level):
    if level not in _comp:
        _comp[level] = LD.libdeflate_alloc_compressor(level)
    a = np.ascontiguousarray(a, dtype=np.uint8)
    cap = a.size + max(128, a.size // 10) + 8
    out = np.empty(cap, dtype=np.uint8)
    n = LD.libdeflate_deflate_compress(_comp[level], a.ctypes.data, a.size, out.ctypes.data, cap)
    assert n > 0
    return out[:n].tobytes()


def ld_crc32(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return LD.libdeflate_crc32(0, a.ctypes.data, a.size)


def frame_block(a, level, fmt, is_last):
    """bgzf::compress / mgzip::compress + Bgzf::encode's EOF rule."""
    payload = ld_deflate(a, level)
    xfl = 2 if level >= 9 else 4 if level <= 1 else 0
    if fmt == "bgzf":
        assert len(payload) < 65536
        hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, xfl, 255, 6, ord("B"), ord("C"), 2,
                          len(payload) + 26 - 1)
    else:
        hdr = struct.pack("<BBBBIBBHBBHI", 31, 139, 8, 4, 0, xfl, 255, 8, ord("I"), ord("G"), 4,
                          len(payload) + 28)
    out = hdr + payload + struct.pack("<II", ld_crc32(a), len(a))
    if is_last and fmt == "bgzf":
        out += BGZF_EOF
    return out


def frame_stream(a, level, fmt, buffer_size):
    """ParCompress::write (strict >) + flush_last(true)."""
    out = []
    pos = 0
    n = len(a)
    while True:
        rem = n - pos
        take = buffer_size if rem > buffer_size else rem
        last = take == rem
        out.append(frame_block(a[pos:pos + take], level, fmt, last))
        pos += take
        if last:
            break
    return b"".join(out), [len(b) for b in out]


def entry(data):
    e = {"size": len(data), "sha256": hashlib.sha256(data).hexdigest()}
    if len(data) <= 1024:
        e["hex"] = data.hex()
    return e


def lazy_vectors():
    """Levels 5-9 (deflate_compress_lazy_generic: lazy 5-7, lazy2 8-9) -> l59_vectors.json."""
    raw = []
    for level in (5, 6, 7, 8, 9):
        edge = 55 - 4 * level  # deflate_compress_none up to here
        for cls in synth.CLASSES:
            for n in (0, edge, edge + 1, 300, 512, 4096, 5000, 10001, 32769, 65280):
                seed = 3000 + n
                a = synth.make(cls, n, seed)
                e = {"class": cls, "n": n, "seed": seed, "level": level}
                e.update(entry(ld_deflate(a, level)))
                raw.append(e)
    for level, cls, n in [(6, "text", 1 << 20), (6, "fastq", 1 << 20), (5, "mixed", 700000), (7, "ascii", 400000),
                          (8, "text", 400000), (9, "dna", 305001), (9, "repeats", 400000), (6, "text", 304999),
                          (7, "mixed", 131072), (9, "mixed", 200000)]:
        a = synth.make(cls, n, 79)
        e = {"class": cls, "n": n, "seed": 79, "level": level}
        e.update(entry(ld_deflate(a, level)))
        raw.append(e)
    streams = []
    for fmt, bs, level, cases in [
        ("bgzf", 65280, 6, [("text", 0), ("text", 65280), ("text", 3 * 65280 + 1234), ("mixed", 300000),
                            ("fastq", 200000), ("random", 70000)]),
        ("bgzf", 65280, 5, [("text", 200000)]),
        ("bgzf", 65280, 7, [("repeats", 200000)]),
        ("bgzf", 65280, 9, [("text", 0), ("fastq", 150000)]),  # XFL = 2 from level 9 on (src/bgzf.rs:278-284)
        ("mgzip", 1 << 20, 6, [("ascii", (1 << 20) + 7)]),
        ("mgzip", 131072, 8, [("mixed", 500000)]),
    ]:
        for cls, n in cases:
            a = synth.make(cls, n, 4444)
            st, blk = frame_stream(a, level, fmt, bs)
            e = {"fmt": fmt, "buffer_size": bs, "class": cls, "n": n, "seed": 4444, "level": level,
                 "block_sizes": blk}
            e.update(entry(st))
            streams.append(e)
    with open(os.path.join(HERE, "l59_vectors.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py lazy",
                   "libdeflate": "v1.10 binary (Ubuntu libdeflate0 1.10-2), compat=1.10",
                   "raw_deflate": raw, "streams": streams}, f, indent=0, separators=(",", ":"))
        f.write("\n")
    print("wrote %d raw, %d stream vectors for levels 5-9" % (len(raw), len(streams)))


def near_optimal_vectors():
    """Levels 10-12 (deflate_compress_near_optimal) -> l1012_vectors.json."""
    raw = []
    for level in (10, 11, 12):
        edge = 55 - 4 * level  # deflate_compress_none up to here
        for cls in synth.CLASSES:
            for n in (0, edge, edge + 1, 300, 4096, 5000, 10001, 20000, 32769, 65280):
                seed = 5000 + n
                a = synth.make(cls, n, seed)
                e = {"class": cls, "n": n, "seed": seed, "level": level}
                e.update(entry(ld_deflate(a, level)))
                raw.append(e)
    # several DEFLATE blocks, the rewind to the previous split check, the 300,000-byte soft limit, window slides
    for level, cls, n in [(10, "text", 1 << 20), (10, "fastq", 700000), (11, "mixed", 700000), (12, "ascii", 400000),
                          (12, "text", 400000), (11, "dna", 305001), (10, "repeats", 400000), (12, "text", 304999),
                          (11, "mixed", 131072), (12, "lowent", 200000)]:
        a = synth.make(cls, n, 81)
        e = {"class": cls, "n": n, "seed": 81, "level": level}
        e.update(entry(ld_deflate(a, level)))
        raw.append(e)
    streams = []
    for fmt, bs, level, cases in [
        ("bgzf", 65280, 10, [("text", 0), ("text", 65280), ("text", 3 * 65280 + 1234), ("mixed", 300000),
                             ("fastq", 200000), ("random", 70000)]),
        ("bgzf", 65280, 11, [("text", 200000)]),
        ("bgzf", 65280, 12, [("repeats", 200000), ("fastq", 150000)]),
        ("mgzip", 1 << 20, 10, [("ascii", (1 << 20) + 7)]),
        ("mgzip", 131072, 12, [("mixed", 500000)]),
    ]:
        for cls, n in cases:
            a = synth.make(cls, n, 4545)
            st, blk = frame_stream(a, level, fmt, bs)
            e = {"fmt": fmt, "buffer_size": bs, "class": cls, "n": n, "seed": 4545, "level": level,
                 "block_sizes": blk}
            e.update(entry(st))
            streams.append(e)
    # libdeflate's default_litlen_costs[] as they sit in the binary's read-only data (found by their first bytes):
    # three rows of 257 literal costs + the length-symbol cost
    blob = open(LD_PATH, "rb").read()
    at = blob.find(bytes([6, 6, 22, 32, 38, 43, 48, 51]))
    tables = [list(blob[at + 258 * k: at + 258 * (k + 1)]) for k in range(3)] if at >= 0 else None
    with open(os.path.join(HERE, "l1012_vectors.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py near_optimal",
                   "libdeflate": "v1.10 binary (Ubuntu libdeflate0 1.10-2), compat=1.10",
                   "default_litlen_costs": tables,
                   "raw_deflate": raw, "streams": streams}, f, indent=0, separators=(",", ":"))
        f.write("\n")
    print("wrote %d raw, %d stream vectors for levels 10-12" % (len(raw), len(streams)))


def main():
    if sys.argv[1:] == ["near_optimal"]:
        near_optimal_vectors()
        return
    if sys.argv[1:] == ["lazy"]:
        return lazy_vectors()
    sizes = [0, 1, 51, 52, 53, 100, 300, 511, 512, 513, 1000, 4096, 5000, 32767, 32768, 32769, 32773,
             40000, 65279, 65280]
    raw = []
    for cls in synth.CLASSES:
        for n in sizes:
            seed = 1000 + n
            a = synth.make(cls, n, seed)
            e = {"class": cls, "n": n, "seed": seed, "level": 1,
                 "input_sha256": hashlib.sha256(a.tobytes()).hexdigest()}
            e.update(entry(ld_deflate(a, 1)))
            raw.append(e)
    # multi-sub-block / beyond-one-window inputs (Mgzip-sized), level 1
    for cls, n in [("text", 1 << 20), ("fastq", 1 << 20), ("dna", 300000), ("repeats", 200000),
                   ("mixed", 200000), ("runs", 131072), ("text", 70534), ("text", 70535),
                   ("text", 70536)]:
        a = synth.make(cls, n, 77)
        e = {"class": cls, "n": n, "seed": 77, "level": 1,
             "input_sha256": hashlib.sha256(a.tobytes()).hexdigest()}
        e.update(entry(ld_deflate(a, 1)))
        raw.append(e)

    literal = []
    for name, data in [("ref_regression_206", bytes(REF_REGRESSION_206)),
                       ("ref_simple_text", REF_SIMPLE_TEXT)]:
        e = {"name": name, "input_hex": data.hex(), "level": 1}
        e.update(entry(ld_deflate(np.frombuffer(data, dtype=np.uint8), 1)))
        literal.append(e)

    streams = []
    for fmt, bs, cases in [
        ("bgzf", 65280, [("text", 0), ("text", 1), ("text", 65280), ("text", 65281),
                         ("text", 2 * 65280 + 1234), ("fastq", 5 * 65280), ("random", 65280 + 7),
                         ("mixed", 300000)]),
        ("bgzf", 32768, [("text", 100000)]),
        ("mgzip", 1 << 20, [("text", (1 << 20) + 5), ("ascii", 1 << 20)]),
        ("mgzip", 65280, [("dna", 200000)]),
    ]:
        for cls, n in cases:
            a = synth.make(cls, n, 4242)
            s, blk = frame_stream(a, 1, fmt, bs)
            e = {"fmt": fmt, "buffer_size": bs, "class": cls, "n": n, "seed": 4242, "level": 1,
                 "block_sizes": blk}
            e.update(entry(s))
            streams.append(e)
    # the reference's own two literal inputs through ParCompress<Bgzf> level 1
    for name, data in [("ref_regression_206", bytes(REF_REGRESSION_206)),
                       ("ref_simple_text", REF_SIMPLE_TEXT)]:
        s, blk = frame_stream(np.frombuffer(data, dtype=np.uint8), 1, "bgzf", 65280)
        e = {"fmt": "bgzf", "buffer_size": 65280, "name": name, "input_hex": data.hex(), "level": 1,
             "block_sizes": blk}
        e.update(entry(s))
        streams.append(e)

    # ---- levels 2-4 (greedy hc_matchfinder): gzp's DEFAULT level is 3 (src/par/compress.rs:54-62)
    #      and level 0 (deflate_compress_none: stored blocks of at most 65535 bytes)
    hc_raw = []
    for level in (0, 2, 3, 4):
        for cls in synth.CLASSES:
            for n in (0, 43, 44, 47, 48, 300, 512, 4096, 5000, 32769, 65280):
                seed = 2000 + n
                a = synth.make(cls, n, seed)
                e = {"class": cls, "n": n, "seed": seed, "level": level}
                e.update(entry(ld_deflate(a, level)))
                hc_raw.append(e)
    for level, cls, n in [(3, "text", 1 << 20), (3, "ascii", 1 << 20), (3, "fastq", 1 << 20), (3, "mixed", 700000),
                          (2, "text", 400000), (4, "repeats", 400000), (3, "dna", 305001), (3, "text", 304999),
                          (0, "mixed", 3 * 65535 + 1)]:
        a = synth.make(cls, n, 78)
        e = {"class": cls, "n": n, "seed": 78, "level": level}
        e.update(entry(ld_deflate(a, level)))
        hc_raw.append(e)
    hc_streams = []
    for fmt, bs, level, cases in [
        ("bgzf", 65280, 3, [("text", 0), ("text", 65280), ("text", 3 * 65280 + 1234), ("mixed", 300000),
                            ("fastq", 200000), ("random", 70000)]),
        ("bgzf", 65280, 2, [("text", 200000)]),
        ("bgzf", 65280, 4, [("repeats", 200000)]),
        ("mgzip", 1 << 20, 3, [("ascii", (1 << 20) + 7), ("text", 2 * (1 << 20))]),  # BASELINE config 3 shape
        ("mgzip", 131072, 3, [("mixed", 500000)]),
        ("bgzf", 65280, 0, [("text", 0), ("mixed", 3 * 65280 + 1234)]),
        ("mgzip", 200000, 0, [("random", 500001)]),  # stored blocks cut at 65535 bytes inside a block
    ]:
        for cls, n in cases:
            a = synth.make(cls, n, 4343)
            st, blk = frame_stream(a, level, fmt, bs)
            e = {"fmt": fmt, "buffer_size": bs, "class": cls, "n": n, "seed": 4343, "level": level,
                 "block_sizes": blk}
            e.update(entry(st))
            hc_streams.append(e)
    with open(os.path.join(HERE, "l234_vectors.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py",
                   "libdeflate": "v1.10 binary (Ubuntu libdeflate0 1.10-2), compat=1.10",
                   "raw_deflate": hc_raw, "streams": hc_streams}, f, indent=0, separators=(",", ":"))
        f.write("\n")
    print("wrote %d raw, %d stream vectors for levels 2-4" % (len(hc_raw), len(hc_streams)))

    doc = {
        "generator": "tests/golden/make_golden.py",
        "libdeflate": "v1.10 binary (Ubuntu libdeflate0 1.10-2), compat=1.10",
        "bgzf_eof_hex": BGZF_EOF.hex(),
        "raw_deflate": raw,
        "raw_deflate_literal_inputs": literal,
        "streams": streams,
    }
    with open(os.path.join(HERE, "l1_vectors.json"), "w") as f:
        json.dump(doc, f, indent=0, separators=(",", ":"))
        f.write("\n")
    print("wrote %d raw, %d literal, %d stream vectors" % (len(raw), len(literal), len(streams)))
    lazy_vectors()


if __name__ == "__main__":
    main()

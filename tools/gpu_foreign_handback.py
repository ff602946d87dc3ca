"""How many members of zlib-made BGZF streams (what bgzip / htslib write) does the decode / copy pair hand back to k_inflate?"""
import sys, os, struct, zlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from gzp_amd import _native, synth
def member(chunk, level, strategy=zlib.Z_DEFAULT_STRATEGY):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    payload = co.compress(chunk) + co.flush()
    hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, ord("B"), ord("C"), 2, len(payload) + 25)
    return hdr + payload + struct.pack("<II", zlib.crc32(chunk), len(chunk))
lib = _native.load()
d = _native.DContext(lib=lib)
for cls in ("text", "fastq", "dna", "ascii", "mixed", "lowentropy" if "lowentropy" in synth.CLASSES else "random"):
    a = synth.make(cls, 16 << 20, 9).tobytes()
    for level in (1, 6, 9):
        s = b"".join(member(a[i:i + 65280], level) for i in range(0, len(a), 65280))
        out = d.decompress(s)
        print(cls, "zlib level", level, "members", (len(a) + 65279) // 65280, "handed back", d.last_redo_count(), "ok", out == a)
d.close()

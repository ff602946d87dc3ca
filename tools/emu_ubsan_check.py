"""One-off robustness check: the product kernels on the CPU emulator, compiled with UBSan
(array bounds of every __shared__ array, shift exponents), over all input classes.

  cd tests/emu && g++ -O1 -g -std=c++17 -fPIC -shared -pthread -fsanitize=bounds,shift-exponent \
      -Wno-unknown-pragmas -I . -I ../../include -x c++ ../../gzp_amd/csrc/gzpx_kernels.hip \
      ../../gzp_amd/csrc/gzpx_synth.hip ../../gzp_amd/csrc/gzpx_api.cpp ../../gzp_amd/csrc/gzpx_par.cpp emu_runtime.cpp \
      -o /tmp/libgzpx_emu_ubsan.so
  LD_PRELOAD=$(gcc -print-file-name=libubsan.so) python tools/emu_ubsan_check.py
"""
import sys, io, zlib, struct
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from gzp_amd import _native, synth, par
from oracle import oracle
lib = _native.GzpxLib('/tmp/libgzpx_emu_ubsan.so')
bad = 0
# compress: all classes, several sizes, levels 1 and 3, bgzf + mgzip large block
for cls in sorted(synth.CLASSES):
    for n, lvl in [(0,1),(51,1),(4000,1),(65280,1),(2*65280+77,1),(70000,3),(70000,6),(40000,9),(33,5)]:
        a = synth.make(cls, n, 5 + n % 7)
        with _native.Context(level=lvl, lib=lib, compat=_native.COMPAT_1_10, max_slab_bytes=max(n,1)) as c:
            got = c.compress_slab(a, True)
        want = oracle.compress_stream(a, oracle.FMT_BGZF, lvl, oracle.COMPAT_1_10, 65280)
        if got != want:
            bad += 1; print("MISMATCH", cls, n, lvl)
        d = _native.DContext(lib=lib)
        if d.decompress(got) != a.tobytes():
            bad += 1; print("INFLATE MISMATCH", cls, n, lvl)
        d.close()
a = synth.make("mixed", 300000, 3)
with _native.Context(format=_native.FORMAT_MGZIP, level=1, buffer_size=131072, lib=lib, compat=_native.COMPAT_1_10, max_slab_bytes=a.size) as c:
    got = c.compress_slab(a, True)
assert got == oracle.compress_stream(a, oracle.FMT_MGZIP, 1, oracle.COMPAT_1_10, 131072)
for lvl, n, bs in [(3, 700000, 330001), (6, 400000, 330001), (8, 200000, 131072)]:  # > 300000-byte blocks: soft limit, recalcs
    a = synth.make("mixed", n, 11 + lvl)
    with _native.Context(format=_native.FORMAT_MGZIP, level=lvl, buffer_size=bs, lib=lib, compat=_native.COMPAT_1_10, max_slab_bytes=a.size) as c:
        got = c.compress_slab(a, True)
    if got != oracle.compress_stream(a, oracle.FMT_MGZIP, lvl, oracle.COMPAT_1_10, bs):
        bad += 1; print("MISMATCH big", lvl, n, bs)
print("done, mismatches:", bad)

"""Parity against the GPU box's OWN libdeflate.so.0, if it has one ("the build may
dlopen it as an extra oracle after fingerprinting the version behaviourally").  libdeflate exports no
version symbol, so the library is classified by the one rule that changed the level-1..4 bitstream
between v1.10 and the pinned v1.24 (A.7 delta 1: the code of an unused offset alphabet), probed with
inputs on which the oracle's two compat modes disagree; the HIP output in the matching compat mode is
then compared with the library block for block.  Run on a box whose library is newer than 1.10 this
is what turns "bit-exact vs 1.24 by construction" into a measured statement."""
import ctypes

import numpy as np
import pytest

from gzp_amd import _native, synth

pytestmark = pytest.mark.gpu


def _box_libdeflate():
    for path in ("libdeflate.so.0", "/lib/x86_64-linux-gnu/libdeflate.so.0", "/usr/lib/x86_64-linux-gnu/libdeflate.so.0",
                 "/usr/lib64/libdeflate.so.0"):
        try:
            L = ctypes.CDLL(path)
        except OSError:
            continue
        L.libdeflate_alloc_compressor.restype = ctypes.c_void_p
        L.libdeflate_alloc_compressor.argtypes = [ctypes.c_int]
        L.libdeflate_deflate_compress.restype = ctypes.c_size_t
        L.libdeflate_deflate_compress.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                                  ctypes.c_size_t]
        return L, path
    return None, None


def _ld(L, comp, a):
    cap = a.size + max(128, a.size // 10) + 64
    out = np.empty(cap, dtype=np.uint8)
    n = L.libdeflate_deflate_compress(comp, a.ctypes.data, a.size, out.ctypes.data, cap)
    assert n > 0
    return out[:n].tobytes()


def test_hip_output_equals_the_boxs_libdeflate(hip_lib, oracle):
    L, path = _box_libdeflate()
    if L is None:
        pytest.skip("no libdeflate.so.0 on this box")
    comps = {lvl: L.libdeflate_alloc_compressor(lvl) for lvl in (1, 2, 3, 4)}
    # ---- fingerprint: inputs whose level-1 output differs between the two rules
    probes = []
    for seed in range(1, 40):  # short printable-ASCII noise: no 4-byte repeat, yet dynamic codes win
        a = synth.make("ascii", 150 + 10 * seed, seed)
        o24, o10 = oracle.deflate_compress(a, 1, oracle.COMPAT_1_24), oracle.deflate_compress(a, 1, oracle.COMPAT_1_10)
        if o24 != o10:
            probes.append((a, o24, o10))
        if len(probes) == 5:
            break
    assert probes, "no discriminating input found"
    votes = set()
    for a, o24, o10 in probes:
        got = _ld(L, comps[1], a)
        votes.add("1.24" if got == o24 else "1.10" if got == o10 else "unknown")
    assert len(votes) == 1 and "unknown" not in votes, (path, votes)
    kind = votes.pop()
    compat = _native.COMPAT_1_24 if kind == "1.24" else _native.COMPAT_1_10
    print("box libdeflate %s behaves like v%s" % (path, kind))
    # ---- block-for-block: the libdeflate-shaped ABI of the HIP library vs the box's library
    rng = np.random.default_rng(20250927)
    classes = sorted(synth.CLASSES)
    hips = {}
    for lvl in (1, 2, 3, 4):
        hips[lvl] = _native.Compressor(level=lvl, compat=compat, lib=hip_lib)
    n_cases = 0
    for it in range(240):
        cls = classes[it % len(classes)]
        lvl = (1, 1, 3, 2, 4, 1)[it % 6]
        n = int(rng.integers(0, 70000)) if it % 5 else int(rng.integers(0, 400))
        a = synth.make(cls, n, 5000 + it)
        assert hips[lvl].deflate_compress(a, a.size + a.size // 8 + 256) == _ld(L, comps[lvl], a), (cls, n, lvl, kind)
        n_cases += 1
    for a, o24, o10 in probes:  # and the discriminating inputs themselves
        assert hips[1].deflate_compress(a) == _ld(L, comps[1], a)
    for h in hips.values():
        h.close()
    # ---- a slab of 256 BGZF blocks through the slab ABI vs the library + gzp's framing, via the oracle's
    # framing of the library's payloads (payload bytes compared directly)
    slab = synth.text_slab(256 * 65280, seed=77)
    with _native.Context(format=_native.FORMAT_BGZF, level=1, buffer_size=65280, compat=compat, lib=hip_lib,
                         max_slab_bytes=slab.size) as ctx:
        out, sizes = ctx.compress_slab(slab, _native.SLAB_FULL_BLOCKS, return_block_sizes=True)
    pos = 0
    for b in range(256):
        blk = out[pos:pos + int(sizes[b])]
        assert blk[18:-8] == _ld(L, comps[1], slab[b * 65280:(b + 1) * 65280]), b
        pos += int(sizes[b])

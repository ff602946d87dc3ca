"""Error-path hygiene of the asynchronous slab ABI: an enqueue that fails half-way through a
submit -- the copy-in may already be reading the caller's buffer -- must drain the context's streams before the
call returns, report a device error, leave the slot free and the context usable.  The CPU emulator's HIP stubs
inject the failure (tests/emu/emu_runtime.cpp: emu_fail_nth_memcpy_async); on a GPU the same wrappers run, the
failure itself cannot be provoked there."""
import ctypes

import numpy as np
import pytest

from gzp_amd import _native, synth


def _hooks(emu_lib):
    L = emu_lib.L
    L.emu_fail_nth_memcpy_async.argtypes = [ctypes.c_long]
    L.emu_fail_nth_memcpy_async.restype = None
    L.emu_stream_sync_count.restype = ctypes.c_long
    return L


@pytest.mark.parametrize("nth", [1, 2, 3])
def test_failing_enqueue_in_compress_submit_drains_and_recovers(emu_lib, oracle, nth):
    L = _hooks(emu_lib)
    a = synth.make("text", 3 * 65280 + 100, 7)
    with _native.Context(format=_native.FORMAT_BGZF, level=1, buffer_size=65280, lib=emu_lib, max_slab_bytes=a.size) as c:
        want = c.compress_slab(a, True)  # warm: slots and staging exist
        before = L.emu_stream_sync_count()
        L.emu_fail_nth_memcpy_async(nth)  # 1: the copy-in; 2, 3: the size / result copies behind the kernels
        try:
            with pytest.raises(_native.GzpxError) as e:
                c.compress_slab(a, True)
        finally:
            L.emu_fail_nth_memcpy_async(0)
        assert e.value.code == _native.ERR_DEVICE
        # copy-in, compute and side stream were all waited for before the call returned
        assert L.emu_stream_sync_count() - before >= 3
        # the slot was not leaked and the context still produces the reference stream
        for _ in range(4):
            assert c.compress_slab(a, True) == want
    assert want == oracle.compress_stream(a, oracle.FMT_BGZF, 1, oracle.COMPAT_1_24, 65280)


@pytest.mark.parametrize("nth", [1, 2, 4])
def test_failing_enqueue_in_decompress_submit_drains_and_recovers(emu_lib, oracle, nth):
    L = _hooks(emu_lib)
    a = synth.make("text", 2 * 65280 + 999, 9)
    comp = oracle.compress_stream(a, oracle.FMT_BGZF, 1, oracle.COMPAT_1_24, 65280)
    with _native.DContext(format=_native.FORMAT_BGZF, lib=emu_lib) as d:
        assert d.decompress(comp) == a.tobytes()
        before = L.emu_stream_sync_count()
        L.emu_fail_nth_memcpy_async(nth)
        try:
            with pytest.raises(_native.GzpxError) as e:
                d.decompress(comp)
        finally:
            L.emu_fail_nth_memcpy_async(0)
        assert e.value.code == _native.ERR_DEVICE
        assert L.emu_stream_sync_count() - before >= 3
        for _ in range(4):
            assert d.decompress(comp) == a.tobytes()

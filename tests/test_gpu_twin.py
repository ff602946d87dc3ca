"""tests/twin_cases.py on the MI355X through the real library (the same checks the emulator runs on
CPU, at larger sizes): asynchronous slab ABI, reserve/commit, index, error surfacing, the overlapped
ParDecompress twin."""
import pytest

import twin_cases as tc

pytestmark = pytest.mark.gpu


def test_submit_wait_pipeline(hip_lib, oracle):
    tc.submit_wait_pipeline(hip_lib, oracle, scale=16)


def test_multi_device(hip_lib, oracle):
    tc.multi_device(hip_lib, oracle, scale=8)


def test_pinned_and_threaded(hip_lib, oracle):
    tc.pinned_and_threaded(hip_lib, oracle, scale=8)


def test_reserve_commit(hip_lib, oracle):
    tc.reserve_commit(hip_lib, oracle, scale=8)


def test_block_index(hip_lib, oracle):
    tc.block_index(hip_lib, oracle, scale=4)


def test_borrowed_writer_and_io_error(hip_lib, oracle):
    tc.borrowed_writer_and_io_error(hip_lib, oracle, scale=4)


def test_builder_errors(hip_lib):
    tc.builder_errors(hip_lib)


def test_par_decompress_overlapped(hip_lib, oracle):
    tc.par_decompress_overlapped(hip_lib, oracle, scale=4)


def test_decompress_submit_wait(hip_lib, oracle):
    tc.decompress_submit_wait(hip_lib, oracle, scale=8)


def test_libdeflate_shim_edges(hip_lib, oracle):
    tc.libdeflate_shim_edges(hip_lib, oracle)


def test_write_chunked(hip_lib, oracle):
    tc.write_chunked(hip_lib, oracle, scale=8)

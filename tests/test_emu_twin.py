"""tests/twin_cases.py through the emulated library (product sources on the CPU SIMT emulator)."""
import twin_cases as tc


def test_submit_wait_pipeline(emu_lib, oracle):
    tc.submit_wait_pipeline(emu_lib, oracle)


def test_multi_device(emu_lib, oracle):
    tc.multi_device(emu_lib, oracle)


def test_pinned_and_threaded(emu_lib, oracle):
    tc.pinned_and_threaded(emu_lib, oracle)


def test_reserve_commit(emu_lib, oracle):
    tc.reserve_commit(emu_lib, oracle)


def test_block_index(emu_lib, oracle):
    tc.block_index(emu_lib, oracle)


def test_borrowed_writer_and_io_error(emu_lib, oracle):
    tc.borrowed_writer_and_io_error(emu_lib, oracle)


def test_builder_errors(emu_lib):
    tc.builder_errors(emu_lib)


def test_par_decompress_overlapped(emu_lib, oracle):
    tc.par_decompress_overlapped(emu_lib, oracle)


def test_decompress_submit_wait(emu_lib, oracle):
    tc.decompress_submit_wait(emu_lib, oracle)


def test_libdeflate_shim_edges(emu_lib, oracle):
    tc.libdeflate_shim_edges(emu_lib, oracle)


def test_write_chunked(emu_lib, oracle):
    tc.write_chunked(emu_lib, oracle)

"""N > 1 path on CPU: world_size-2 gloo run of the block-shard + ordered gather, each rank
compressing its shard with the emulated kernels; rank 0 checks the gathered stream against the
oracle's single-process stream.  No GPU."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r)
    sys.path.insert(0, os.path.join(%(root)r, "tests", "emu"))
    import numpy as np, torch, torch.distributed as dist
    import build_emu
    from gzp_amd import _native, shard, synth
    from oracle import oracle
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    lib = _native.GzpxLib(build_emu.build())
    total = %(total)d
    stream = synth.make("text", total, 77)            # every rank can regenerate the stream
    lo, n = shard.shard_bytes(total, 65280, world)[rank]
    with _native.Context(level=1, compat=_native.COMPAT_1_10, lib=lib, max_slab_bytes=max(n, 1)) as ctx:
        mode = shard.slab_mode(rank, world, total, 65280)
        mine = ctx.compress_slab(stream[lo:lo + n], mode) if mode is not None else b""
    out = shard.ordered_gather(torch.frombuffer(bytearray(mine), dtype=torch.uint8) if mine
                               else torch.empty(0, dtype=torch.uint8), dst=0)
    # the pipelined form bench.py uses: two gathers in flight, preallocated destination
    mine_t = (torch.frombuffer(bytearray(mine), dtype=torch.uint8) if mine else torch.empty(0, dtype=torch.uint8))
    pre = torch.empty(4 * 70000 * world + 64, dtype=torch.uint8) if rank == 0 else None
    h1 = shard.ordered_gather_start(mine_t, dst=0, out=pre)
    h2 = shard.ordered_gather_start(mine_t.clone(), dst=0)
    out1, out2 = h1.wait(), h2.wait()
    if rank == 0:
        want = oracle.compress_stream(stream, oracle.FMT_BGZF, 1, oracle.COMPAT_1_10, 65280)
        assert out.numpy().tobytes() == want, "sharded stream differs from the single-process stream"
        assert out1.numpy().tobytes() == want and out2.numpy().tobytes() == want
        print("GLOO_OK", total, len(want))
    dist.barrier()
    dist.destroy_process_group()
""")


@pytest.mark.parametrize("total", [5 * 65280 + 4321, 65280, 100])
def test_two_rank_shard_and_ordered_gather(tmp_path, total):
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    build_emu.build()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "total": total})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(29500 + total % 200),
                        str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "GLOO_OK" in r.stdout


def test_shard_ranges_cover_the_stream():
    from gzp_amd import shard
    for total in (0, 1, 65280, 65281, 10 * 65280 + 5, 576_716_800 * 8):
        for world in (1, 2, 3, 8):
            parts = shard.shard_bytes(total, 65280, world)
            assert parts[0][0] == 0 and sum(n for _, n in parts) == total
            for (lo, n), (lo2, _) in zip(parts, parts[1:]):
                assert lo + n == lo2
            nonempty = [n for _, n in parts if n]
            for n in nonempty[:-1]:
                assert n % 65280 == 0  # every shard but the stream's tail is whole blocks
            if nonempty:
                assert max(nonempty) - min(nonempty[:-1] or nonempty) <= 65280
            modes = [shard.slab_mode(r, world, total, 65280) for r in range(world)]
            assert modes.count(1) == 1  # exactly one rank owns the tail (SLAB_LAST)

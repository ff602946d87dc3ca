"""shard.PeerWindow -- the copy-engine write-out (the writer's buffer IPC-mapped into every rank, each rank copies its
shard to its stream offset) -- with two and with EIGHT ranks (configs[3]'s world size) on the one GPU a test box has: gloo carries the sizes, the
payload path (IPC handle, offsets, device-to-device copies, the two buffers of the window used in turn and the view of a
step surviving the next step's copies) is the real one; the in-order concatenation at world 8 is compared with the
single-device stream step by step."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.parametrize("world,total", [(2, 9 * 65280 + 1234), (8, 29 * 65280 + 2048), (8, 5 * 65280 + 77)])
def test_ranks_copy_their_shards_into_the_writers_window(world, total):
    # (8 ranks, 6 blocks: two ranks own no block at all -- shard.slab_mode() is None for them, they take part with 0 bytes)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), PW_TOTAL=str(total), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "peer_window_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("rank %d ok" % r) in o, o[-3000:]

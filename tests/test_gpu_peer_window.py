"""shard.PeerWindow -- the copy-engine write-out (the writer's buffer IPC-mapped into every rank, each rank copies its
shard to its stream offset) -- with two ranks on the one GPU a test box has: gloo carries the sizes, the payload path
(IPC handle, offsets, device-to-device copies, reuse of the window) is the real one."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_two_ranks_copy_their_shards_into_the_writers_window():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "peer_window_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("rank %d ok" % r) in o, o[-3000:]

"""The N > 1 write-out code on RCCL itself, as far as one GPU allows: a one-rank `nccl` process group
(backend "nccl" is RCCL on ROCm) running the sizes all_gather, the ordered gather and the
offset-only variant on device tensors.  The multi-rank behaviour is covered on CPU (gloo,
tests/test_dist_gloo.py); this pins the backend-specific API use (device tensors, work handles)."""
import os
import socket

import numpy as np
import pytest
import torch

from gzp_amd import _native, shard, synth

pytestmark = pytest.mark.gpu


def test_ordered_gather_and_offsets_on_rccl_world_1(hip_lib):
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1"})
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        a = synth.make("text", 3 * 65280 + 17, 5)
        with _native.Context(format=_native.FORMAT_BGZF, level=1, buffer_size=65280, lib=hip_lib,
                             max_slab_bytes=a.size) as c:
            want = c.compress_slab(a, True)
        local = torch.from_numpy(np.frombuffer(want, dtype=np.uint8).copy()).cuda()
        out = torch.empty(local.numel() + 100, dtype=torch.uint8, device="cuda")
        h = shard.ordered_gather_start(local, dst=0, out=out)
        got = h.wait()
        assert bytes(got.cpu().numpy()) == want
        off, total, sizes = shard.stream_offsets(local.numel(), local.device)
        assert (off, total, sizes) == (0, local.numel(), [local.numel()])
        dist.barrier()
    finally:
        dist.destroy_process_group()

"""One rank of tests/test_gpu_peer_window.py (started twice, both on GPU 0: the IPC mapping, the offsets and the
copy path are real; crossing devices needs a multi-GPU box)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from gzp_amd import _native, shard, synth  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = 9 * 65280 + 1234
    a = synth.make("text", total, 21)
    lo, n = shard.shard_bytes(total, 65280, world)[rank]
    mode = shard.slab_mode(rank, world, total, 65280)
    with _native.Context(format=_native.FORMAT_BGZF, level=1, buffer_size=65280, max_slab_bytes=max(n, 65280)) as c:
        cap = c.slab_bound(total)
        win = shard.PeerWindow(cap, torch.device("cuda:0"), dst=0)
        d_in = torch.from_numpy(a[lo:lo + n].copy()).cuda()
        d_out = torch.empty(c.slab_bound(n), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for step in range(3):  # the window is reused step after step
            out_len, _ = c.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), d_out.numel(), mode)
            got = win.gather_start(d_out[:out_len]).wait()
            if rank == 0:
                with _native.Context(format=_native.FORMAT_BGZF, level=1, buffer_size=65280, max_slab_bytes=total) as one:
                    want = one.compress_slab(a, True)
                assert bytes(got.cpu().numpy()) == want, "peer-window stream differs from the single-device stream (step %d)" % step
            dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()

"""One rank of tests/test_gpu_peer_window.py (started 2 or 8 times, all on GPU 0: the IPC mapping, the offsets and the
copy path are real; crossing devices needs a multi-GPU box).  Every step compresses a DIFFERENT stream, so a copy that
lands in the wrong buffer of the window -- or too early, over a view the writer still holds -- shows up as wrong bytes:
the reuse contract of shard.PeerWindow (depth 2: the view of step k is good until the writer's wait() of step k + 1)
is checked by holding the previous step's view across the next step's copies."""
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
    total = int(os.environ.get("PW_TOTAL", 9 * 65280 + 1234))
    steps = int(os.environ.get("PW_STEPS", 4))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, n = shard.shard_bytes(total, 65280, world)[rank]
    mode = shard.slab_mode(rank, world, total, 65280)
    with _native.Context(format=_native.FORMAT_BGZF, level=1, buffer_size=65280, max_slab_bytes=max(n, 65280)) as c:
        cap = c.slab_bound(total)
        win = shard.PeerWindow(cap, torch.device("cuda:0"), dst=0)
        assert win.depth == 2
        d_out = torch.empty(c.slab_bound(max(n, 1)), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        prev_view, prev_want = None, None
        for step in range(steps):  # the window is reused step after step, its two buffers in turn
            a = synth.make("text", total, 21 + step)
            out_len = 0
            if mode is not None:
                d_in = torch.from_numpy(a[lo:lo + n].copy()).cuda()
                out_len, _ = c.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), d_out.numel(), mode)
            h = win.gather_start(d_out[:out_len])
            # every rank's copy of THIS step has landed, the writer has not called wait() of this step yet:
            if h._ev is not None:
                h._ev.synchronize()
            dist.barrier()
            if rank == 0 and prev_view is not None:  # ... so the previous step's view must still be whole
                assert bytes(prev_view.cpu().numpy()) == prev_want, "step %d's copies ran over the view of step %d" % (step, step - 1)
            got = h.wait()
            if rank == 0:
                with _native.Context(format=_native.FORMAT_BGZF, level=1, buffer_size=65280, max_slab_bytes=total) as one:
                    want = one.compress_slab(a, True)
                assert bytes(got.cpu().numpy()) == want, "peer-window stream differs from the single-device stream (step %d)" % step
                prev_view, prev_want = got, want
            dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()

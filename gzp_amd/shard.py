"""Multi-GPU sharding of one BGZF/Mgzip stream: independent blocks -> contiguous block ranges per
rank, no data-path collective while compressing; one ordered variable-size gather of the
compressed shards to the writer rank (RCCL over xGMI with cuda tensors; gloo works the same on
CPU tensors, which is how the N > 1 path is tested without GPUs).

The property preserved is the reference's in-order writer loop (src/par/compress.rs:305-310):
the gathered bytes are exactly the single-process stream, because Bgzf/Mgzip blocks share no
state (needs_dict() == false, src/deflate.rs:458-460, 608-610) and only the globally last block
carries is_last / the EOF marker (src/deflate.rs:622-624).
"""
import numpy as np

from . import _native


def shard_blocks(total_blocks, world):
    """[(first_block, n_blocks)] per rank: contiguous, balanced to within one block."""
    base, extra = divmod(total_blocks, world)
    out = []
    first = 0
    for r in range(world):
        nb = base + (1 if r < extra else 0)
        out.append((first, nb))
        first += nb
    return out


def shard_bytes(total_bytes, block_size, world):
    """[(first_byte, n_bytes)] per rank for a stream of total_bytes cut at block_size."""
    total_blocks = 1 if total_bytes == 0 else -(-total_bytes // block_size)
    out = []
    for first, nb in shard_blocks(total_blocks, world):
        lo = min(first * block_size, total_bytes)
        hi = min((first + nb) * block_size, total_bytes)
        out.append((lo, hi - lo))
    return out


def slab_mode(rank, world, total_bytes, block_size):
    """How rank's shard is cut: the rank that owns the stream's final block compresses it as the
    tail (SLAB_LAST: short/empty final piece + EOF marker); ranks with no block return None."""
    total_blocks = 1 if total_bytes == 0 else -(-total_bytes // block_size)
    first, nb = shard_blocks(total_blocks, world)[rank]
    if nb == 0:
        return None
    return _native.SLAB_LAST if first + nb == total_blocks else _native.SLAB_FULL_BLOCKS


def ordered_gather(local, dst=0, group=None):
    """Gather 1-D uint8 tensors of different lengths to `dst`, concatenated in rank order.

    Sizes travel with one all_gather (8 bytes per rank), payloads with point-to-point
    send/recv straight into their final offsets (RCCL has no gatherv; each peer has its own
    xGMI link to the root, so the transfers run concurrently).  Returns the stream on dst,
    None elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    sizes = torch.zeros(world, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(sizes, n, group=group)
    if rank != dst:
        if local.numel():
            dist.send(local, dst=dst, group=group)
        return None
    sz = [int(x) for x in sizes.tolist()]
    offs = np.concatenate([[0], np.cumsum(sz)]).astype(np.int64)
    out = torch.empty(int(offs[-1]), dtype=torch.uint8, device=local.device)
    reqs = []
    for r in range(world):
        if r == dst or sz[r] == 0:
            continue
        reqs.append(dist.irecv(out[offs[r]:offs[r + 1]], src=r, group=group))
    out[offs[dst]:offs[dst + 1]].copy_(local)
    for q in reqs:
        q.wait()
    return out

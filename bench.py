#!/usr/bin/env python
"""bench.py -- BGZF level-1 compression throughput of the MI355X-native ParCompress<Bgzf> path.

Metric (BASELINE.json): "BGZF compress MiB/s at level 1, 550 MiB text".
One step = one pass of the whole hot path (candidates -> match/parse -> Huffman -> CRC -> scan
-> emit) over one 550 MiB synthetic text slab that is already resident in HBM, producing the
complete BGZF stream in HBM.  With N > 1 ranks every rank compresses its own slab (independent
blocks, no data-path collective: weak scaling) and the compressed shards are gathered in rank
order to rank 0 over RCCL inside the timed step (the in-order write-out exchange).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SLAB_BYTES = 576_716_800  # 550 MiB = shakespeare.txt x 100 shape (README.md:166-167)
BLOCK = 65280             # Bgzf::DEFAULT_BUFSIZE (src/deflate.rs:583)
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/), if any."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f)["hbm_bytes_per_launch"].get(kernel)
    except Exception:
        return None


def available_cores():
    """Hardware threads this process may actually use: the affinity mask, capped by the cgroup
    CPU quota (cpu.max) of the container."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = ""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            q = max(1, int(int(quota) / int(period)))
            if q < n:
                note = " (cgroup cpu.max caps this container at %d of the host's %d hardware threads)" % (q, n)
                n = q
    except Exception:
        pass
    return n, note


def cpu_baseline(slab, wall_s=8.0):
    """gzp's CPU path on this box's host cores, the way ParCompress runs it -- one worker per usable
    hardware thread, each owning a contiguous run of the slab's blocks -- natively timed (pthreads,
    oracle/cpu_bench.c) for `wall_s` seconds.  The work per block is libdeflate_deflate_compress
    (level 1) + libdeflate_crc32 done by the image's libdeflate binary, the library gzp binds
    ("reference"); if the box has no libdeflate.so, the oracle's C restatement ("port").  A reported
    baseline, not a target."""
    from oracle import oracle
    oracle.build()
    cores, note = available_cores()
    r = oracle.cpu_bench_compress_ref(slab, 1, BLOCK, threads=cores, wall_s=wall_s)
    kind, what = "reference", "the image's libdeflate.so (deflate_compress level 1 + crc32 per block)"
    if r is None:
        r = oracle.cpu_bench_compress(slab, oracle.FMT_BGZF, 1, oracle.COMPAT_1_24, BLOCK, threads=cores,
                                      wall_s=wall_s)
        kind, what = "port", "the oracle's C restatement of that path"
    nbytes, dt, used = r
    return {
        "value": round(nbytes / dt / 2**20, 1),
        "unit": "MiB/s",
        "cores": used,
        "kind": kind,
        "sample": "%d native worker threads, each re-encoding its contiguous share of the same slab's BGZF "
                  "blocks with %s until %.0f s elapsed (%.1f MiB compressed in %.2f s)%s"
                  % (used, what, wall_s, nbytes / 2**20, dt, note),
    }


def cpu_baseline_inflate(comp, offs, sizes, wall_s=8.0):
    """CPU side of the ParDecompress row: the image's libdeflate binary (the library gzp binds through
    libdeflater: libdeflate_deflate_decompress + libdeflate_crc32 per block, src/bgzf.rs:103-121 /
    src/par/decompress.rs:162-186), one native worker per hardware thread over its contiguous share
    of the same blocks, for `wall_s` seconds (oracle/cpu_bench.c)."""
    from oracle import oracle
    oracle.build()
    cores, note = available_cores()
    r = oracle.cpu_bench_inflate(comp, offs, sizes, 18, threads=cores, wall_s=wall_s)
    if r is None:
        return {"value": None, "unit": "MiB/s", "cores": 0, "kind": "reference",
                "sample": "no libdeflate.so on this box"}
    nbytes, dt, used = r
    return {
        "value": round(nbytes / dt / 2**20, 1),
        "unit": "MiB/s",
        "cores": used,
        "kind": "reference",
        "sample": "%d native worker threads, each inflating + CRC-checking its contiguous share of the same "
                  "stream's BGZF blocks with the image's libdeflate.so until %.0f s elapsed (%.1f MiB "
                  "inflated in %.2f s)%s" % (used, wall_s, nbytes / 2**20, dt, note),
    }


def run_inflate(args, torch, dist, world, rank, local_rank, dev):
    """--workload inflate: BASELINE.json configs[4] -- ParDecompress<Bgzf> over the output of the
    compress workload.  One step = scan-free multi-block inflate + per-block CRC check of the whole
    BGZF stream (already resident in HBM) into HBM."""
    from gzp_amd import _native, synth
    n = args.slab_bytes
    slab = synth.text_slab(n, seed=20250927 + rank)
    with _native.Context(format=_native.FORMAT_BGZF, level=1, buffer_size=BLOCK, compat=_native.COMPAT_1_24,
                         device=local_rank, max_slab_bytes=n) as c:
        comp = np.frombuffer(c.compress_slab(slab, True), dtype=np.uint8).copy()
        device_name = c.device_name()
    d = _native.DContext(format=_native.FORMAT_BGZF, device=local_rank)
    offs, sizes, used = d.scan_blocks(comp)
    d_in = torch.from_numpy(comp).to(dev)
    d_out = torch.empty(n + 64, dtype=torch.uint8, device=dev)

    def step():
        return d.decompress_device(d_in.data_ptr(), comp.size, offs, sizes, d_out.data_ptr(), n + 64)

    for _ in range(args.warmup):
        step()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    kern_ms = 0.0
    got = 0
    for _ in range(args.steps):
        got = step()
        kern_ms += d.last_inflate_ms()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        ok = got == n and d_out[:n].cpu().numpy().tobytes() == slab.tobytes()
        kern_ms /= args.steps
        achieved = (comp.size + n) / (kern_ms * 1e-3) / 1e9  # reads the stream, writes the text
        res = {
            "metric": "BGZF decompress MiB/s (inflated bytes) of the level-1 550 MiB text stream",
            "value": round(n * world / 2**20 / (dt / args.steps), 1),
            "unit": "MiB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "ParDecompress<Bgzf>: GPU multi-block inflate of output from config 2, CRC "
                            "check, MiB/s vs CPU path",
                "compressed_bytes": int(comp.size),
                "inflated_bytes": n,
                "blocks": int(offs.size),
                "parallelism": "block-shard x%d" % world,
                "verified_round_trip": bool(ok),
                "device": device_name,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_inflate",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": pmc_traffic("k_inflate"),
                "kernel_ms": round(kern_ms, 3),
            },
        }
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline_inflate(comp, offs, sizes)
        print(json.dumps(res))
    d.close()


def verify(slab, out_bytes, block_sizes, tail=True):
    """Outside the timed region: gzip-validity of the whole stream prefix + bit-exactness of a
    sample of blocks against the oracle."""
    import gzip
    from oracle import oracle
    offs = np.concatenate([[0], np.cumsum(block_sizes.astype(np.int64))])
    nb = len(block_sizes)
    idx = sorted(set([0, 1, nb // 2, nb - 2, nb - 1]) & set(range(nb)))
    for b in idx:
        want = oracle.encode_block(slab[b * BLOCK:(b + 1) * BLOCK], oracle.FMT_BGZF, 1,
                                   oracle.COMPAT_1_24, is_last=(tail and b == nb - 1))
        got = out_bytes[offs[b]:offs[b + 1]].tobytes()
        if got != want:
            return False
    k = min(64, nb)
    if gzip.decompress(out_bytes[:offs[k]].tobytes()) != slab[:min(k * BLOCK, slab.size)].tobytes():
        return False
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--slab-bytes", type=int, default=SLAB_BYTES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["compress", "inflate"], default="compress",
                    help="compress = the headline metric (default); inflate = the ParDecompress row")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from gzp_amd import _native, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" %
                         (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    if args.workload == "inflate":
        run_inflate(args, torch, dist, world, rank, local_rank, dev)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # One logical stream of world x 550 MiB, sharded at block boundaries (weak scaling: every
    # rank gets 550 MiB +- one block); only the rank that owns the final block emits the tail.
    from gzp_amd import shard
    total = args.slab_bytes * world
    lo, n = shard.shard_bytes(total, BLOCK, world)[rank]
    mode = shard.slab_mode(rank, world, total, BLOCK)
    slab = synth.text_slab(n, seed=20250927 + rank)
    d_in = torch.from_numpy(slab).to(dev)
    ctx = _native.Context(format=_native.FORMAT_BGZF, level=1, buffer_size=BLOCK,
                          compat=_native.COMPAT_1_24, device=local_rank, max_slab_bytes=n)
    cap = ctx.slab_bound(n)
    # two output buffers: with N > 1 the gather of one step's shard (RCCL, asynchronous) runs while
    # the next step compresses into the other buffer, the way ParCompress's lanes overlap
    d_outs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(2 if world > 1 else 1)]
    d_out = d_outs[0]
    gathered = torch.empty(cap * world, dtype=torch.uint8, device=dev) if (world > 1 and rank == 0) else None
    nb = ctx.n_blocks(n)
    block_sizes = np.zeros(nb, dtype=np.uint32)
    ctx.set_profiling(True)
    state = {"i": 0, "pending": None}

    def step():
        buf = d_outs[state["i"] % len(d_outs)]
        state["i"] += 1
        out_len, _ = ctx.compress_slab_device(d_in.data_ptr(), n, buf.data_ptr(), cap, mode,
                                              None, block_sizes)
        if world > 1:
            # in-order write-out: ordered variable-size gather of the shards to rank 0 (RCCL),
            # started now and completed while the next step compresses
            if state["pending"] is not None:  # the previous shard has arrived (it travelled while
                state["pending"].wait()       # this step compressed); `gathered` is free again
            state["pending"] = shard.ordered_gather_start(buf[:out_len], dst=0, out=gathered)
        return out_len

    def drain():
        if state["pending"] is not None:
            state["pending"].wait()
            state["pending"] = None

    stage_acc = {}
    for _ in range(args.warmup):
        step()
    drain()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    out_len = 0
    for _ in range(args.steps):
        out_len = step()
        for k, v in ctx.last_stage_ms().items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
    drain()  # the last gather belongs to the timed region
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ms_per_step = dt / args.steps * 1e3
    total_mib = total / 2**20
    value = total_mib / (dt / args.steps)

    if rank == 0:
        last_buf = d_outs[(state["i"] - 1) % len(d_outs)]
        ok = verify(slab, last_buf[:out_len].cpu().numpy(), block_sizes, tail=(mode == _native.SLAB_LAST))
        stage_ms = {k: v / args.steps for k, v in stage_acc.items()}
        dom = max(stage_ms, key=stage_ms.get)
        alg_bytes = n + out_len  # 1 B read + r B written per input byte
        achieved = alg_bytes / (stage_ms[dom] * 1e-3) / 1e9
        traffic = pmc_traffic(dom)
        res = {
            "metric": "BGZF compress MiB/s at level 1, 550 MiB text",
            "value": round(value, 1),
            "unit": "MiB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "Single MI355X: 64 KiB BGZF blocks, level 1, 550 MiB text slab, "
                            "bit-exact vs libdeflate",
                "slab_bytes": total,
                "shard_bytes": n,
                "block_size": BLOCK,
                "blocks": nb,
                "level": 1,
                "format": "bgzf",
                "ratio": round(out_len / n, 4),
                "parallelism": "block-shard x%d%s" % (world, " + ordered RCCL gather" if world > 1 else ""),
                "verified_bit_exact_sample": bool(ok),
                "device": ctx.device_name(),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dom,
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
            },
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is timed at N = 1 only
            res["cpu_baseline"] = cpu_baseline(slab)
        print(json.dumps(res))
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""bench.py -- BGZF level-1 compression throughput of the MI355X-native ParCompress<Bgzf> path.

Metric (BASELINE.json): "BGZF compress MiB/s at level 1, 550 MiB text".
One step = one pass of the whole hot path (candidates -> match/parse -> Huffman -> CRC -> scan
-> emit) over one 550 MiB synthetic text slab that is already resident in HBM, producing the
complete BGZF stream in HBM.  With N > 1 ranks every rank compresses its own slab (independent
blocks, no data-path collective: weak scaling) and the compressed shards are gathered in rank
order to rank 0 over RCCL inside the timed step (the in-order write-out exchange).

    python bench.py [--gpus N] [--steps K] [--warmup W]        (N > 1 launches itself under
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   torch.distributed.run)

The same JSON line carries, measured in the same run after the headline region (N = 1):
    "e2e"      host-to-host rates: pinned submit/wait pipeline, and the Write API (ParCompress twin)
    "inflate"  the ParDecompress row (BASELINE configs[4]) over the stream just produced
Other workloads: --workload inflate (configs[4] alone), --workload fastq (configs[3]: a 32 GiB
synthetic FASTQ stream generated in HBM, sharded over the ranks).
"""
import argparse
import ctypes
import hashlib
import json
import os
import socket
import subprocess
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
FASTQ_STREAM = 32 << 30   # BASELINE configs[3]
FASTQ_SEED = 20250927


def pmc_traffic(kernel, build_id=None):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_traffic.json) -- only if they were
    collected with THIS build of the library (the file records gzpx_build_id of the run that made it): counters of
    another build's kernels are refused, the line then says `traffic: null` and why."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            doc = json.load(f)
        if build_id is not None and doc.get("build_id") != build_id:
            return None
        return doc["hbm_bytes_per_launch"].get(kernel)
    except Exception:
        return None


def pmc_traffic_bounds(kernel, build_id=None):
    """(low, high, why) where the FETCH_SIZE correction is not calibrated for the kernel's access pattern (k_inflate's
    gathers: tools/summarize_profiles.py), else None -- same build rule as pmc_traffic."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            doc = json.load(f)
        if build_id is not None and doc.get("build_id") != build_id:
            return None
        return doc.get("hbm_bytes_per_launch_bounds", {}).get(kernel)
    except Exception:
        return None


def pmc_traffic_source(build_id):
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            have = json.load(f).get("build_id")
    except Exception:
        return "no profiles/pmc_traffic.json"
    if have != build_id:
        return ("profiles/pmc_traffic.json was collected with build %s, this library is build %s: refused "
                "(tools/pmc_traffic.sh + tools/summarize_profiles.py refresh it)" % (have, build_id))
    return ("profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of this same command with "
            "this same build %s of the library; committed, not collected in this run)" % build_id)


CLOCK_HZ = 2.4e9          # MI355X peak shader clock (MI355X_MICROARCH.md); DVFS runs lower under load
N_SIMD = 256 * 4


def issue_roofline(kernel, kernel_ms, build_id, workload="bench.py"):
    """The second bound beside HBM (round 5): instruction issue, from the committed SQ counters of THIS build
    (profiles/sq_counters.json, tools/pmc_sq.sh).  `issue_frac` = VALU wave-instructions x 4 cycles / (1024 SIMDs x the
    kernel's duration x the 2.4 GHz peak clock): the share of the kernel's time in which its SIMDs issue VALU work.  Four
    cycles per wave-instruction and SIMD is what the counters themselves say for this integer code (one quad-cycle of
    SQ_ACTIVE_INST_VALU per SQ_INSTS_VALU); the 2-cycle rate behind the 157 TFLOP/s vector peak is packed FP32.
    `wave_parked_frac` = SQ_WAIT_ANY / SQ_WAVE_CYCLES: the share of its resident time an average wave spends parked at
    s_waitcnt / s_barrier; `lds_busy_frac` = SQ_LDS_IDX_ACTIVE (cycles) / (256 CUs x the kernel's cycles)."""
    try:
        with open(os.path.join(ROOT, "profiles", "sq_counters.json")) as f:
            doc = json.load(f)
        if doc.get("build_id") != build_id:
            return {"issue_frac": None, "source": "profiles/sq_counters.json belongs to build %s, this library is %s: refused"
                                                  % (doc.get("build_id"), build_id)}
        c = doc["workloads"][workload][kernel]
        cyc = N_SIMD * kernel_ms * 1e-3 * CLOCK_HZ
        return {"issue_frac": round(c["INSTS_VALU"] * 4.0 / cyc, 4),
                "valu_wave_insts": int(c["INSTS_VALU"]), "salu_wave_insts": int(c["INSTS_SALU"]),
                "lds_wave_insts": int(c["INSTS_LDS"]),
                "wave_parked_frac": round(c["WAIT_ANY"] / c["WAVE_CYCLES"], 4),
                "lds_busy_frac": round(c["LDS_IDX_ACTIVE"] / (256 * kernel_ms * 1e-3 * CLOCK_HZ), 4),
                "source": "profiles/sq_counters.json (rocprofv3 --pmc SQ_* passes of this build %s; committed, not "
                          "collected in this run)" % build_id}
    except Exception as e:
        return {"issue_frac": None, "source": "no usable profiles/sq_counters.json (%r)" % (e,)}


def golden_stream(name):
    """The entry `name` of tests/golden/fullsize.json: SHA-256 of a COMPLETE stream (and of its framed
    block sizes) made by the libdeflate binary + gzp's framing (tests/golden/make_fullsize.py)."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "fullsize.json")) as f:
            for e in json.load(f)["streams"]:
                if e["name"] == name:
                    return e
    except Exception:
        pass
    return None


def check_full_stream(name, n, seed, out_host, block_sizes=None):
    """Compare a whole output stream with the committed digest of the libdeflate-made one.  Returns
    (True / False, digest) or (None, digest) when this run is not the golden configuration."""
    sha = hashlib.sha256(out_host).hexdigest()
    g = golden_stream(name)
    if g is None or g["input"].get("n") != n or g["input"].get("seed") != seed:
        return None, sha
    ok = sha == g["sha256"] and int(out_host.size) == g["size"]
    if ok and block_sizes is not None:
        ok = hashlib.sha256(np.ascontiguousarray(block_sizes, dtype="<u4").tobytes()).hexdigest() == g["block_sizes_sha256"]
    return bool(ok), sha


def box_libdeflate():
    """Which libdeflate the GPU box itself carries (none / behaves like v1.10 / like >= v1.1x = the
    pinned 1.24 rule), told by inputs on which the oracle's two compat modes disagree -- the
    fingerprint of tests/test_gpu_box_libdeflate.py.  Outside every timed region."""
    try:
        from gzp_amd import synth
        from oracle import oracle
        L = None
        for path in ("libdeflate.so.0", "/lib/x86_64-linux-gnu/libdeflate.so.0", "/usr/lib/x86_64-linux-gnu/libdeflate.so.0",
                     "/usr/lib64/libdeflate.so.0"):
            try:
                L = ctypes.CDLL(path)
                break
            except OSError:
                continue
        if L is None:
            return None
        L.libdeflate_alloc_compressor.restype = ctypes.c_void_p
        L.libdeflate_alloc_compressor.argtypes = [ctypes.c_int]
        L.libdeflate_deflate_compress.restype = ctypes.c_size_t
        L.libdeflate_deflate_compress.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                                  ctypes.c_size_t]
        comp = L.libdeflate_alloc_compressor(1)
        votes = set()
        for seed in range(1, 40):
            a = synth.make("ascii", 150 + 10 * seed, seed)
            o24 = oracle.deflate_compress(a, 1, oracle.COMPAT_1_24)
            o10 = oracle.deflate_compress(a, 1, oracle.COMPAT_1_10)
            if o24 == o10:
                continue
            out = np.empty(a.size + 256, dtype=np.uint8)
            k = L.libdeflate_deflate_compress(comp, a.ctypes.data, a.size, out.ctypes.data, out.size)
            got = out[:k].tobytes()
            votes.add("1.24-like" if got == o24 else "1.10-like" if got == o10 else "unknown")
            if len(votes) > 1 or "unknown" in votes:
                return "unknown"
        return votes.pop() if votes else None
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


def cpu_baseline(slab, wall_s=3.0):
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


def cpu_baseline_parcompress(slab, want_sha=None, wall_s=3.0):
    """The CPU baseline BASELINE.md 3 / SURVEY 8(d) promise: gzp's ParCompress<Bgzf> itself -- a caller thread that
    write_all()s the slab in 64 KiB chunks (benches/bench.rs:36-45,121) and cuts blocks, num_threads workers behind
    queues bounded at 2 N, one in-order writer thread into an in-memory sink (src/par/compress.rs:248-469, restated
    with pthreads in oracle/cpu_bench.c) -- over the image's libdeflate binary, level 1, num_threads = every usable
    hardware thread, whole passes over the slab for `wall_s` seconds.  The last pass's stream is compared with the
    GPU's (SHA-256)."""
    from oracle import oracle
    oracle.build()
    cores, note = available_cores()
    r = oracle.cpu_bench_parcompress_ref(slab, 1, BLOCK, 65536, threads=cores, wall_s=wall_s)
    if r is None:
        return {"value": None, "unit": "MiB/s", "cores": 0, "kind": "reference", "sample": "no libdeflate.so on this box"}
    nbytes, dt, passes, stream = r
    same = None if want_sha is None else bool(hashlib.sha256(stream).hexdigest() == want_sha)
    return {
        "value": round(nbytes / dt / 2**20, 1),
        "unit": "MiB/s",
        "cores": cores,
        "kind": "reference",
        "MiBps_per_core": round(nbytes / dt / 2**20 / cores, 1),
        "stream_equals_gpu_stream": same,
        "sample": "ParCompress<Bgzf> twin in C (caller thread: 64 KiB write_all calls cut into 65,280-byte blocks; %d "
                  "workers calling the image's libdeflate.so, deflate_compress level 1 + crc32 + BGZF framing, queues "
                  "bounded at 2 N; one in-order writer thread into an in-memory sink): %d whole passes over the same "
                  "550 MiB slab, each a complete spawn / write / finish / join lifetime, in %.2f s%s"
                  % (cores, passes, dt, note),
    }


def cpu_baseline_inflate(comp, offs, sizes, wall_s=3.0):
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


class Env:
    """Where the bench runs: ranks, device, library.  --emulate (tests only) swaps the HIP library for
    the CPU-emulated build and RCCL for gloo, so that the N > 1 control flow can be exercised
    without GPUs; such a run measures nothing."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        from gzp_amd import _native
        self.torch, self.dist, self.native = torch, dist, _native
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.emulate = args.emulate
        if self.emulate:
            sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
            import build_emu
            self.lib = _native.GzpxLib(build_emu.build())
            self.dev = torch.device("cpu")
            self.device_index = 0
        else:
            self.lib = _native.GzpxLib(args.lib) if args.lib else _native.load()
            torch.cuda.set_device(self.local_rank)
            self.dev = torch.device("cuda", self.local_rank)
            self.device_index = self.local_rank
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.emulate:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=self.dev)

    def sync(self):
        if self.world > 1:
            self.dist.barrier()
        if not self.emulate:
            self.torch.cuda.synchronize()

    def max_over_ranks(self, dt):
        if self.world > 1:
            t = self.torch.tensor([dt], dtype=self.torch.float64, device=self.dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def all_ranks(self, x):
        """One float from every rank, as a list on every rank."""
        if self.world == 1:
            return [float(x)]
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.dev)
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def close(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()


def run_inflate(args, env, emit=True, d_stream=None, comp_host=None, slab=None, steps=None):
    """BASELINE.json configs[4] -- ParDecompress<Bgzf> over the output of the compress workload.  One
    step = multi-block inflate + per-block CRC check of the whole BGZF stream (already resident in
    HBM) into HBM.  Also the "inflate" leg of the default line (d_stream / comp_host given)."""
    torch, _native = env.torch, env.native
    n = args.slab_bytes
    steps = steps or args.steps
    device_name = ""
    if comp_host is None:
        from gzp_amd import synth
        slab = synth.text_slab(n, seed=20250927 + env.rank)
        with _native.Context(format=_native.FORMAT_BGZF, level=1, buffer_size=BLOCK, compat=_native.COMPAT_1_24,
                             device=env.device_index, max_slab_bytes=n, lib=env.lib) as c:
            comp_host = np.frombuffer(c.compress_slab(slab, True), dtype=np.uint8).copy()
            device_name = c.device_name()
        d_stream = torch.from_numpy(comp_host).to(env.dev)
    d = _native.DContext(format=_native.FORMAT_BGZF, device=env.device_index, lib=env.lib)
    offs, sizes, used = d.scan_blocks(comp_host)
    d_out = torch.empty(n + 64, dtype=torch.uint8, device=env.dev)

    def step():
        return d.decompress_device(d_stream.data_ptr(), comp_host.size, offs, sizes, d_out.data_ptr(), n + 64)

    for _ in range(min(args.warmup, 2) if not emit else args.warmup):
        step()
    env.sync()
    t0 = time.perf_counter()
    kern_ms = 0.0
    dec_ms = copy_ms = 0.0
    got = 0
    for _ in range(steps):
        got = step()
        kern_ms += d.last_inflate_ms()
        a, b = d.last_inflate_stage_ms()
        dec_ms += a
        copy_ms += b
    env.sync()
    dt = env.max_over_ranks(time.perf_counter() - t0)
    res = None
    if env.rank == 0:
        ok = got == n and bool((d_out[:n].cpu() == torch.from_numpy(slab)).all())
        kern_ms /= steps
        dec_ms /= steps
        copy_ms /= steps
        seg_route = dec_ms > 0
        # the dominant kernel: k_inflate_seg (Huffman decode) on the decode / copy route, k_inflate on the other
        dom, dom_ms = ("k_inflate_seg", dec_ms) if seg_route else ("k_inflate", kern_ms)
        achieved = (comp_host.size + n) / (max(dom_ms, 1e-9) * 1e-3) / 1e9  # reads the stream, writes the text
        res = {
            "metric": "BGZF decompress MiB/s (inflated bytes) of the level-1 550 MiB text stream",
            "value": round(n * env.world / 2**20 / (dt / steps), 1),
            "unit": "MiB/s",
            "n_gpus": env.world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "ParDecompress<Bgzf>: GPU multi-block inflate of output from config 2, CRC "
                            "check, MiB/s vs CPU path",
                "compressed_bytes": int(comp_host.size),
                "inflated_bytes": n,
                "blocks": int(offs.size),
                "parallelism": "block-shard x%d" % env.world,
                "verified_round_trip": bool(ok),
                "route": "k_inflate_seg + k_lzcopy (hand-backs: k_inflate)" if seg_route else "k_inflate",
                "handed_back_members": int(d.last_redo_count()) if seg_route else 0,
                "device": device_name,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dom,
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": pmc_traffic(dom, env.lib.build_id()),
                "traffic_bounds": pmc_traffic_bounds(dom, env.lib.build_id()),
                "issue": issue_roofline(dom, dom_ms, env.lib.build_id(), "bench.py --workload inflate"),
                "kernel_ms": round(dom_ms, 3),
                "inflate_kernels_ms": round(kern_ms, 3),  # every inflate kernel of a step (HIP events around them)
                "pipeline_frac": round((comp_host.size + n) / (max(kern_ms, 1e-9) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            },
        }
        if seg_route:
            res["roofline"]["k_lzcopy_ms"] = round(copy_ms, 3)
        if env.world == 1 and not env.emulate and not args.no_extras:  # (--no-extras: the profiling runs want the step's launches only)
            try:  # what a caller of the Read API sees: host stream in, host bytes out (PCIe both ways; never `value`)
                res["e2e"] = inflate_e2e(env, comp_host, slab)
            except Exception as e:
                res["e2e"] = {"error": repr(e)}
        if not args.no_cpu_baseline and env.world == 1:
            res["cpu_baseline"] = cpu_baseline_inflate(comp_host, offs, sizes, wall_s=3.0 if emit else 2.0)  # (a rate measurement, not a soak)
        if emit:
            print(json.dumps(res))
    d.close()
    return res


def inflate_e2e(env, comp_host, slab):
    """The ParDecompress twin (gzp::ParDecompress, src/par/decompress.rs:112-352) host to host over the bench stream: a
    reader thread that cuts slabs of whole blocks, a device thread with three slabs in flight, the caller draining in
    stream order -- through BufRead's fill_buf / consume (the slab's page-locked bytes, no copy) and through
    `Read::read` into a caller's buffer (one memcpy per byte).  Best of two passes each; every pass CRC-checked by
    the kernels, the bytes of the readinto pass compared with the input."""
    import io
    import zlib
    from gzp_amd import par
    blob = comp_host.tobytes()
    n = slab.size
    out = {}
    buf = bytearray(64 << 20)
    for mode in ("fill_buf", "readinto"):
        best, ok = None, True
        for rep in range(2):
            r = par.ParDecompressBuilder(par.Bgzf, lib=env.lib).device(env.device_index).from_reader(io.BytesIO(blob))
            t0 = time.perf_counter()
            total, crc = 0, 0
            check = mode == "readinto" and rep == 0  # (the first readinto pass carries the comparison, the second the rate)
            while True:
                if mode == "fill_buf":
                    k = len(r.fill_buf())
                    r.consume(k)
                else:
                    k = r.readinto(buf)
                    if k and check:
                        crc = zlib.crc32(memoryview(buf)[:k], crc)
                if not k:
                    break
                total += k
            dt = time.perf_counter() - t0
            r.close()
            ok = ok and total == n and (not check or crc == zlib.crc32(slab))
            if not check:
                best = dt if best is None else min(best, dt)
        out["read_%s_MiBps" % mode] = round(n / 2**20 / best, 1)
        out["read_%s_ok" % mode] = bool(ok)
    out["what"] = ("ParDecompress twin, host stream -> host bytes, default slabs (16 MiB compressed), three in flight; "
                   "fill_buf = the slab's page-locked bytes without a copy, readinto = one memcpy into the caller's buffer")
    return out


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


def level_legs(env, d_in, n):
    """The same device-resident slab at gzp's default level (3: greedy parser), through the lazy (6)
    and lazy2 (9 = Compression::best()) parsers measured after the
    headline region: four timed slabs each (two at level 9); every output is inflated and CRC-checked on
    the GPU and compared with the input."""
    torch, _native = env.torch, env.native
    out = {}
    for level in (3, 6, 9):
        # (level 12 -- the near-optimal parser, libdeflate 1.10's, seconds per slab -- left the default line in round 5:
        # `--workload bgzf3 --level 12` still times it)
        try:
            out["level_%d" % level] = _level_leg(env, d_in, n, level)
        except Exception as e:  # (one level's failure does not take the others' numbers with it)
            out["level_%d" % level] = {"error": repr(e)}
    return out


def mgzip3_leg(env, n=4 << 30):
    """BASELINE configs[2] inside the default line: Mgzip, 1 MiB blocks, level 3, 4 GiB of printable-ASCII noise
    generated in HBM (the generator and seed of `--workload mgzip3`); the whole stream is compared with the
    libdeflate-made digest in tests/golden/fullsize.json, inflated and CRC-checked on the GPU, and the inflation of
    this very stream -- 4,096 members of 1 MiB -- is timed too."""
    torch, _native = env.torch, env.native
    fmt, bs = _native.FORMAT_MGZIP, 1 << 20
    d_in = torch.empty(n + 64, dtype=torch.uint8, device=env.dev)
    _native.synth_ascii_device(d_in.data_ptr(), 0, n, 8, lib=env.lib)
    ctx = _native.Context(format=fmt, level=3, buffer_size=bs, compat=_native.COMPAT_1_24,
                          device=env.device_index, max_slab_bytes=n, lib=env.lib)
    cap = ctx.slab_bound(n)
    d_out = torch.empty(cap, dtype=torch.uint8, device=env.dev)
    ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
    env.sync()
    ctx.set_profiling(True)
    steps = 3
    acc = {}
    t0 = time.perf_counter()
    for _ in range(steps):
        out_len, _ = ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
        for k, v in ctx.last_stage_ms().items():
            acc[k] = acc.get(k, 0.0) + v / steps
    env.sync()
    dt = (time.perf_counter() - t0) / steps
    host = d_out[:out_len].cpu().numpy()
    full_ok, sha = check_full_stream("config3_ascii_%dGiB_mgzip_l3" % (n >> 30), n, 8, host)
    d = _native.DContext(format=fmt, device=env.device_index, lib=env.lib)
    offs, sizes, used = d.scan_blocks(host)
    d_back = torch.empty(n + 64, dtype=torch.uint8, device=env.dev)
    got = d.decompress_device(d_out.data_ptr(), used, offs, sizes, d_back.data_ptr(), n + 64)
    ok = got == n and bool(torch.equal(d_back[:n], d_in[:n]))
    env.sync()
    t1 = time.perf_counter()
    kms = 0.0
    for _ in range(3):
        d.decompress_device(d_out.data_ptr(), used, offs, sizes, d_back.data_ptr(), n + 64)
        kms += d.last_inflate_ms() / 3
    env.sync()
    inflate_leg = {"MiBps": round(n / 2**20 / ((time.perf_counter() - t1) / 3), 1), "k_inflate_ms": round(kms, 3),
                   "members": int(offs.size),
                   "roofline": {"bound": "hbm", "kernel": "k_inflate_seg + k_lzcopy (every inflate kernel of the step)", "achieved": round((n + out_len) / (kms * 1e-3) / 1e9, 2),
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round((n + out_len) / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}}
    d.close()
    ctx.close()
    dom = max(acc, key=acc.get)
    achieved = (n + out_len) / (max(acc[dom], 1e-9) * 1e-3) / 1e9
    res = {"workload": "Single MI355X: Mgzip 1 MiB blocks, level 3, %d GiB /dev/urandom-seeded ASCII" % (n >> 30),
           "MiBps": round(n / 2**20 / dt, 1), "ms_per_step": round(dt * 1e3, 3), "steps": steps, "slab_bytes": n,
           "blocks": int(offs.size), "ratio": round(out_len / n, 4), "gpu_inflate_crc_roundtrip_ok": bool(ok),
           "stream_sha256": sha, "verified_bit_exact_full": full_ok,
           "roofline": {"bound": "hbm", "kernel": dom, "kernel_ms": round(acc[dom], 3), "achieved": round(achieved, 2),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                        "pipeline_frac": round((n + out_len) / dt / 1e9 / HBM_PEAK_GBS, 5),
                        "stage_ms": {k: round(v, 3) for k, v in acc.items()}},
           "inflate_of_output": inflate_leg}
    del d_in, d_out, d_back
    return res


def _level_leg(env, d_in, n, level):
    torch, _native = env.torch, env.native
    ctx = _native.Context(format=_native.FORMAT_BGZF, level=level, buffer_size=BLOCK, compat=_native.COMPAT_1_24,
                          device=env.device_index, max_slab_bytes=n, lib=env.lib)
    cap = ctx.slab_bound(n)
    d_out = torch.empty(cap, dtype=torch.uint8, device=env.dev)
    if level < 10:
        ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
    env.sync()
    ctx.set_profiling(True)  # HIP events around every launch group, as in the headline region
    steps = 4 if level < 9 else 2 if level < 10 else 1
    stage_acc = {}
    t0 = time.perf_counter()
    for _ in range(steps):
        out_len, _ = ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
        for k, v in ctx.last_stage_ms().items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v / steps
    env.sync()
    dt = (time.perf_counter() - t0) / steps
    host = d_out[:out_len].cpu().numpy()
    full_ok, sha = check_full_stream("text_550MiB_bgzf_l%d" % level, n, 20250927, host)
    d = _native.DContext(format=_native.FORMAT_BGZF, device=env.device_index, lib=env.lib)
    offs, sizes, used = d.scan_blocks(host)
    d_back = torch.empty(n + 64, dtype=torch.uint8, device=env.dev)
    got = d.decompress_device(d_out.data_ptr(), used, offs, sizes, d_back.data_ptr(), n + 64)
    ok = got == n and bool(torch.equal(d_back[:n], d_in[:n]))
    d.close()
    compat_name = "libdeflate 1.10" if ctx.active_compat() == _native.COMPAT_1_10 else "libdeflate >= 1.1x (1.24)"
    ctx.close()
    dom = max(stage_acc, key=stage_acc.get)
    achieved = (n + out_len) / (max(stage_acc[dom], 1e-9) * 1e-3) / 1e9
    # the library's stage label covers a launch GROUP; at levels 3-4 the matcher inside it is the compaction kernel
    launches = {"k_match_hc+k_parse_hc": ("k_hc_init, k_match_hc_sparse, k_hc_orphan, k_parse_hc, k_match_hc_stale, k_parse_hc x2"
                                          if level in (3, 4) else "k_hc_init, k_match_hc, k_hc_orphan, k_parse_hc x3")}
    res = {"MiBps": round(n / 2**20 / dt, 1), "ms_per_step": round(dt * 1e3, 3), "steps": steps,
                               "compat_in_force": compat_name,  # (levels 10-12: the 1.10 rules whatever was asked for)
                               "ratio": round(out_len / n, 4), "gpu_inflate_crc_roundtrip_ok": bool(ok),
                               "stream_sha256": sha, "verified_bit_exact_full": full_ok,
                               "roofline": {"bound": "hbm", "kernel": dom, "launches": launches.get(dom, dom),
                                            "kernel_ms": round(stage_acc[dom], 3),
                                            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": round(achieved / HBM_PEAK_GBS, 5),
                                            "pipeline_frac": round((n + out_len) / dt / 1e9 / HBM_PEAK_GBS, 5)}}
    del d_out, d_back
    return res


def e2e_legs(env, slab, want_sha):
    """Host-to-host rates of the same slab, measured after the headline region (SURVEY 8(d) timings
    ii and iii):
      device_pinned  page-locked host buffers through gzpx_compress_slab_submit / _wait, 8 slabs,
                     up to three in flight: copy-in, kernels and copy-out overlap on three streams
      api_write      the Write API: ParCompress twin (C ABI gzpx_par_*), ordinary pageable input,
                     one write_all of the whole slab + finish(), the writer callback is a sink
      api_write_64k  the same through 64 KiB write() calls, the shape of benches/bench.rs:36-45
                     (called from Python; api_write_64k_native: the same calls looped in C)
    Every output is checked against the device-resident result (SHA-256)."""
    _native = env.native
    L = env.lib.L
    n = slab.size
    out = {}
    nb_total = -(-n // BLOCK)
    per = -(-nb_total // 8)
    cuts = [min(n, i * per * BLOCK) for i in range(9)]
    pieces = [(cuts[i], cuts[i + 1]) for i in range(8) if cuts[i + 1] > cuts[i]]
    ctx = _native.Context(format=_native.FORMAT_BGZF, level=1, buffer_size=BLOCK, compat=_native.COMPAT_1_24,
                          device=env.device_index, max_slab_bytes=per * BLOCK, lib=env.lib)
    cap = ctx.slab_bound(n) + 8 * 4096
    p_in = L.gzpx_host_alloc(n)
    p_out = L.gzpx_host_alloc(cap)
    if not p_in or not p_out:
        return {"error": "gzpx_host_alloc failed"}
    ctypes.memmove(p_in, slab.ctypes.data, n)
    best = None
    digest = None
    for rep in range(3):  # first repetition warms the staging buffers
        t0 = time.perf_counter()
        tickets, pos, total = [], 0, 0
        sizes = []
        for i, (lo, hi) in enumerate(pieces):
            mode = _native.SLAB_LAST if i == len(pieces) - 1 else _native.SLAB_FULL_BLOCKS
            bound = ctx.slab_bound(hi - lo)
            if len(tickets) == 3:
                t, o = tickets.pop(0)
                sizes.append((o, ctx.wait(t)[0]))
            tk = ctx.submit(p_in + lo, hi - lo, p_out + pos, bound, mode)
            tickets.append((tk, pos))
            pos += bound
        for t, o in tickets:
            sizes.append((o, ctx.wait(t)[0]))
        dt = time.perf_counter() - t0
        if rep and (best is None or dt < best):
            best = dt
        if rep == 2:
            h = hashlib.sha256()
            buf = (ctypes.c_uint8 * cap).from_address(p_out)
            view = np.frombuffer(buf, dtype=np.uint8)
            for o, s in sizes:
                h.update(view[o:o + s])
            digest = h.hexdigest()
    out["device_pinned_MiBps"] = round(n / 2**20 / best, 1)
    out["device_pinned_ok"] = digest == want_sha
    ctx.close()
    L.gzpx_host_free(p_in)
    L.gzpx_host_free(p_out)

    # the Write API
    state = {"h": None, "n": 0}

    def sink(user, data, nbytes):
        state["n"] += nbytes
        if state["h"] is not None:
            state["h"].update(ctypes.string_at(data, nbytes))
        return 0

    cb = _native.WRITE_FN(sink)
    for name, chunk in (("api_write_MiBps", n), ("api_write_64k_MiBps", 65536), ("api_write_64k_native_MiBps", -65536)):
        best = None
        for rep in range(3):
            state["h"] = hashlib.sha256() if rep == 2 else None
            state["n"] = 0
            cfg = _native.GzpxParConfig(_native.FORMAT_BGZF, 1, _native.COMPAT_1_24, env.device_index, BLOCK, 8, 1024)
            h = ctypes.c_void_p()
            env.lib.check(L.gzpx_par_create(ctypes.byref(cfg), cb, None, ctypes.byref(h)))
            base = slab.ctypes.data
            t0 = time.perf_counter()
            if chunk < 0:  # the same 64 KiB write() calls looped on the native side (no ctypes cost per call)
                rc = L.gzpx_par_write_chunked(h, base, n, -chunk)
            else:
                for lo in range(0, n, chunk):
                    rc = L.gzpx_par_write(h, base + lo, min(chunk, n - lo))
                    if rc:
                        break
            rc = rc or L.gzpx_par_finish(h)
            dt = time.perf_counter() - t0
            L.gzpx_par_destroy(h)
            if rc:
                return dict(out, error="gzpx_par rc %d" % rc)
            if rep == 1:  # (the hashing repetition is not a timing)
                best = dt
            elif rep == 0:
                pass
        out[name] = round(n / 2**20 / best, 1)
        out[name.replace("MiBps", "ok")] = state["h"].hexdigest() == want_sha
    return out


def run_fastq(args, env):
    """BASELINE.json configs[3]: BGZF level 1 on a 32 GiB synthetic FASTQ stream, sharded over the
    ranks at block boundaries (strong scaling: the stream is fixed), in-order gather to rank 0."""
    torch, dist, _native = env.torch, env.dist, env.native
    from gzp_amd import shard
    total = args.stream_bytes
    lo, n = shard.shard_bytes(total, BLOCK, env.world)[env.rank]
    mode = shard.slab_mode(env.rank, env.world, total, BLOCK)
    d_in = torch.empty(n + 64, dtype=torch.uint8, device=env.dev)
    _native.synth_fastq_device(d_in.data_ptr(), lo, n, FASTQ_SEED, lib=env.lib)  # generated in HBM
    ctx = _native.Context(format=_native.FORMAT_BGZF, level=1, buffer_size=BLOCK, compat=_native.COMPAT_1_24,
                          device=env.device_index, max_slab_bytes=n, lib=env.lib)
    cap = ctx.slab_bound(n)
    d_out = torch.empty(cap, dtype=torch.uint8, device=env.dev)
    nb = ctx.n_blocks(n)
    block_sizes = np.zeros(nb, dtype=np.uint32)
    gathered = None
    out_len = 0

    def step():
        ol, _ = ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, mode, None, block_sizes)
        g = None
        if env.world > 1:
            g = shard.ordered_gather(d_out[:ol], dst=0)
        return ol, g

    for _ in range(args.warmup):
        step()
    env.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_len, gathered = step()
    env.sync()
    dt = env.max_over_ranks(time.perf_counter() - t0)
    # ---- verification, outside the timed region
    info = {}
    if env.world > 1:
        stream = gathered  # rank 0: the whole stream, in order
    else:
        stream = d_out[:out_len]
    if env.rank == 0:
        host = stream.cpu().numpy()
        info["stream_bytes_out"] = int(host.size)
        info["stream_sha256"] = hashlib.sha256(host).hexdigest()
        # the stream's integrity check on the GPU: every block inflated and CRC-checked (what gzip -t does)
        d = _native.DContext(format=_native.FORMAT_BGZF, device=env.device_index, lib=env.lib)
        offs, sizes, used = d.scan_blocks(host)
        ok = used == host.size
        piece = 1 << 30  # compressed bytes per call: whole blocks
        b0 = 0
        d_chk = torch.empty(min(total, 4 << 30) + BLOCK, dtype=torch.uint8, device=env.dev)
        d_ref = torch.empty_like(d_chk)
        upos = 0
        while ok and b0 < offs.size:
            b1 = b0
            while b1 < offs.size and int(offs[b1]) + int(sizes[b1]) - int(offs[b0]) <= piece and (b1 - b0) * BLOCK < (4 << 30):
                b1 += 1
            lo_c, hi_c = int(offs[b0]), int(offs[b1 - 1]) + int(sizes[b1 - 1])
            got = d.decompress_device(stream.data_ptr() + lo_c, hi_c - lo_c, offs[b0:b1] - np.uint64(lo_c),
                                      sizes[b0:b1], d_chk.data_ptr(), d_chk.numel())
            _native.synth_fastq_device(d_ref.data_ptr(), upos, got, FASTQ_SEED, lib=env.lib)
            if not env.emulate:
                torch.cuda.synchronize()
            ok = ok and bool((d_chk[:got] == d_ref[:got]).all())
            upos += got
            b0 = b1
        info["gpu_inflate_crc_roundtrip_ok"] = bool(ok and upos == total)
        d.close()
        # gzip -t on the two ends of the stream (a prefix of whole blocks; the tail incl. the EOF marker)
        k = int(np.searchsorted(offs, 64 << 20))
        pre = host[:int(offs[k])] if k < offs.size else host
        k2 = int(np.searchsorted(offs, max(0, host.size - (64 << 20))))
        suf = host[int(offs[min(k2, offs.size - 1)]):]
        try:
            rc = [subprocess.run(["gzip", "-t"], input=x.tobytes(), capture_output=True).returncode for x in (pre, suf)]
            info["gzip_t_prefix_suffix_rc"] = rc
        except FileNotFoundError:
            info["gzip_t_prefix_suffix_rc"] = None
        res = {
            "metric": "BGZF compress MiB/s at level 1, 32 GiB synthetic FASTQ (configs[3])",
            "value": round(total / 2**20 / (dt / args.steps), 1),
            "unit": "MiB/s",
            "n_gpus": env.world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": dict({
                "workload": "8xMI355X input shard + RCCL gather: BGZF level 1 on 32 GiB synthetic FASTQ, in-order "
                            "concat verified by gzip -t",
                "stream_bytes": total, "shard_bytes": n, "block_size": BLOCK,
                "blocks": -(-total // BLOCK),
                "parallelism": "block-shard x%d%s" % (env.world, " + ordered RCCL gather" if env.world > 1 else ""),
                "ratio": round(info["stream_bytes_out"] / total, 4),
                "device": ctx.device_name(),
            }, **info),
        }
        print(json.dumps(res))
    ctx.close()


def run_config(args, env, fmt, level, bs, kind, n, label):
    """Any other (format, level, block size, input) on one GPU per rank, device-resident: BASELINE
    configs[2] (--workload mgzip3: Mgzip, 1 MiB blocks, level 3, 4 GiB of printable-ASCII noise generated
    in HBM) and the text slab at gzp's default level (--workload bgzf3)."""
    torch, _native = env.torch, env.native
    from gzp_amd import synth
    if kind == "ascii":
        d_in = torch.empty(n + 64, dtype=torch.uint8, device=env.dev)
        _native.synth_ascii_device(d_in.data_ptr(), 0, n, 8 + env.rank, lib=env.lib)
    else:
        d_in = torch.from_numpy(synth.text_slab(n, seed=20250927 + env.rank)).to(env.dev)
    ctx = _native.Context(format=fmt, level=level, buffer_size=bs, compat=_native.COMPAT_1_24,
                          device=env.device_index, max_slab_bytes=n, lib=env.lib)
    cap = ctx.slab_bound(n)
    d_out = torch.empty(cap, dtype=torch.uint8, device=env.dev)
    ctx.set_profiling(True)
    if getattr(args, "debug_flags", 0):  # (experiments: Config.debug, as in the level-1 workload)
        ctx.debug_set_flags(args.debug_flags)
    acc = {}
    out_len = 0
    for _ in range(args.warmup):
        ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
    env.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_len, nb = ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
        for k, v in ctx.last_stage_ms().items():
            acc[k] = acc.get(k, 0.0) + v / args.steps
    env.sync()
    dt = env.max_over_ranks(time.perf_counter() - t0)
    if env.rank == 0:
        # integrity: every block inflated and CRC-checked on the GPU, compared with the input on the device
        host = d_out[:out_len].cpu().numpy()
        d = _native.DContext(format=fmt, device=env.device_index, lib=env.lib)
        offs, sizes, used = d.scan_blocks(host)
        d_back = torch.empty(n + 64, dtype=torch.uint8, device=env.dev)
        got = d.decompress_device(d_out.data_ptr(), used, offs, sizes, d_back.data_ptr(), n + 64)
        ok = got == n and bool(torch.equal(d_back[:n], d_in[:n]))
        # ... and the ParDecompress side on THIS stream (configs[2]'s 1 MiB members leave the one-wave-
        # per-member kernel 4,096 members for 5,120 wave slots): two timed inflations of the output
        env.sync()
        t1 = time.perf_counter()
        kms = 0.0
        for _ in range(2):
            d.decompress_device(d_out.data_ptr(), used, offs, sizes, d_back.data_ptr(), n + 64)
            kms += d.last_inflate_ms() / 2
        env.sync()
        inflate_leg = {"MiBps": round(n / 2**20 / ((time.perf_counter() - t1) / 2), 1), "k_inflate_ms": round(kms, 3),
                       "members": int(offs.size)}
        d.close()
        dom = max(acc, key=acc.get)
        achieved = (n + out_len) / (max(acc[dom], 1e-9) * 1e-3) / 1e9
        print(json.dumps({
            "metric": label, "value": round(n * env.world / 2**20 / (dt / args.steps), 1), "unit": "MiB/s",
            "n_gpus": env.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": label, "slab_bytes": n, "block_size": bs, "level": level,
                       "format": "bgzf" if fmt == _native.FORMAT_BGZF else "mgzip", "ratio": round(out_len / n, 4),
                       "gpu_inflate_crc_roundtrip_ok": bool(ok), "stream_sha256": hashlib.sha256(host).hexdigest(),
                       "inflate_of_output": inflate_leg,
                       "device": ctx.device_name()},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "stage_ms": {k: round(v, 3) for k, v in acc.items()}}}))
    ctx.close()


class PeerWatchdog:
    """Bounds the peer-window leg of an N > 1 run (main()): after `seconds` rank 0 prints the line it already has, with
    `writeouts.peer_error`, and every rank ends with exit code 0 -- a write-out that hangs on some box must not take the
    measured RCCL figures with it."""

    def __init__(self, seconds, rank, res):
        # (the line as it stands when the peer leg begins: the timer thread prints this snapshot, never the live dict the
        # main thread may be adding the peer figures to)
        self.seconds, self.rank = seconds, rank
        self.res = json.loads(json.dumps(res)) if res is not None else None
        self.lock, self.done = threading.Lock(), False
        self.timer = threading.Timer(seconds, self._fire)
        self.timer.daemon = True
        self.timer.start()

    def _fire(self):
        with self.lock:
            if self.done:
                return
            self.done = True
        if self.res is not None:
            self.res.setdefault("writeouts", {})["peer_error"] = (
                "the peer-window write-out did not finish within %.0f s (watchdog; GZPX_BENCH_PEER_TIMEOUT): the line "
                "carries the other write-outs only" % self.seconds)
            self.res["writeouts"]["peer_leg"] = "timeout"  # (the exit code stays 0: the measured line is complete)
            print(json.dumps(self.res), flush=True)
        else:
            time.sleep(2.0)  # (rank 0 writes first)
        os._exit(0)

    def stop(self):
        """True when the leg ended before the watchdog fired."""
        with self.lock:
            mine = not self.done
            self.done = True
        self.timer.cancel()
        return mine


def strong_leg(args, env, ctx, cap, gathered):
    """See main(): the metric's own slab at N GPUs.  Uses the rank's existing context (its batch holds a whole slab) and
    the weak region's gather buffer; W warm-up + K timed steps between barriers, max over ranks, the write-out of step i
    overlapping the compression of step i + 1 exactly as in the weak regions."""
    torch, _native = env.torch, env.native
    from gzp_amd import shard, synth
    world, rank = env.world, env.rank
    total = args.slab_bytes
    slab0 = synth.text_slab(total, seed=20250927)
    lo, n = shard.shard_bytes(total, BLOCK, world)[rank]
    mode = shard.slab_mode(rank, world, total, BLOCK)
    d_in = torch.from_numpy(slab0[lo:lo + n].copy()).to(env.dev)
    bufs = [torch.empty(max(ctx.slab_bound(max(n, 1)), 64), dtype=torch.uint8, device=env.dev) for _ in range(2)]
    st = {"i": 0, "pending": None, "view": None}

    def wait_pending():
        if st["pending"] is not None:
            v = st["pending"].wait()
            st["pending"] = None
            if v is not None:
                st["view"] = v

    def step():
        buf = bufs[st["i"] % 2]
        st["i"] += 1
        out_len = 0
        if mode is not None:
            out_len, _ = ctx.compress_slab_device(d_in.data_ptr(), n, buf.data_ptr(), buf.numel(), mode)
        wait_pending()
        st["pending"] = shard.ordered_gather_start(buf[:out_len], dst=0, out=gathered)

    for _ in range(args.warmup):
        step()
    wait_pending()
    env.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    wait_pending()
    env.sync()
    per_rank = env.all_ranks((time.perf_counter() - t0) / args.steps * 1e3)
    if rank != 0:
        return None
    host = st["view"].cpu().numpy()
    sha = hashlib.sha256(host).hexdigest()
    full_ok = None
    if not env.emulate:
        full_ok, _ = check_full_stream("config2_text_550MiB_bgzf_l1", total, 20250927, host)
    elif total <= (4 << 20):  # (the CPU dry run: the oracle's single-process stream of the same slab)
        from oracle import oracle
        full_ok = host.tobytes() == oracle.compress_stream(slab0, oracle.FMT_BGZF, 1, oracle.COMPAT_1_24, BLOCK)
    ms = max(per_rank)
    return {"MiBps": round(total / 2**20 / (ms * 1e-3), 1), "ms_per_step": round(ms, 3),
            "rank_ms_per_step": [round(x, 3) for x in per_rank], "slab_bytes": total, "shard_bytes_rank0": n,
            "writeout": "rccl", "stream_sha256": sha, "verified_bit_exact_full": full_ok,
            "what": "THE %d-byte slab of the metric cut into %d contiguous block ranges, one per rank, ordered RCCL gather "
                    "of the compressed shards to rank 0: strong scaling, the whole gathered stream against the "
                    "libdeflate-made digest" % (total, world)}


def self_launch(args):
    """`python bench.py --gpus N` from a bare shell: re-execute under torch.distributed.run, one rank
    per GPU (the driver may also launch it that way itself; then WORLD_SIZE is already set)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--slab-bytes", type=int, default=SLAB_BYTES)
    ap.add_argument("--stream-bytes", type=int, default=FASTQ_STREAM, help="--workload fastq: the whole stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the headline region (no e2e / inflate legs)")
    ap.add_argument("--writeout", choices=["rccl", "offsets"], default="rccl",
                    help="N > 1 in-order write-out: rccl = ordered gather of the compressed shards to rank 0 over "
                         "xGMI (north_star); offsets = all_gather of the shard sizes only, every rank copies its "
                         "shard to its own page-locked buffer at its stream offset")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak (default; the driver's contract) = every rank compresses its own 550 MiB, value = "
                         "N x 550 MiB / step; strong = THE 550 MiB slab of the metric sharded over the N ranks and gathered in "
                         "order, value = 550 MiB / step.  The weak line carries the strong figure as `strong_550MiB` either way")
    ap.add_argument("--workload", choices=["compress", "inflate", "fastq", "mgzip3", "bgzf3"], default="compress",
                    help="compress = the headline metric (default); inflate = the ParDecompress row; fastq = configs[3]; "
                         "mgzip3 = configs[2] (Mgzip 1 MiB blocks, level 3, 4 GiB ASCII); bgzf3 = the text slab at level 3")
    ap.add_argument("--level", type=int, default=3, help="--workload bgzf3: any built level (0-12) instead of 3")
    ap.add_argument("--emulate", action="store_true", help=argparse.SUPPRESS)  # tests: CPU emulator + gloo, no timing value
    # development: A/B runs (gzpx_debug_set_flags: 2 = level 1 through the dense k_match / k_parse pair) and
    # experiment builds of the library; the driver's line uses neither
    ap.add_argument("--debug-flags", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--lib", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1 and "RANK" not in os.environ:
        self_launch(args)  # does not return
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))

    env = Env(args)
    torch, dist, _native = env.torch, env.dist, env.native
    from gzp_amd import synth
    world, rank = env.world, env.rank
    if args.workload == "inflate":
        run_inflate(args, env)
        env.close()
        return
    if args.workload == "fastq":
        run_fastq(args, env)
        env.close()
        return
    if args.workload == "mgzip3":
        run_config(args, env, _native.FORMAT_MGZIP, 3, 1 << 20, "ascii",
                   (4 << 30) if args.slab_bytes == SLAB_BYTES else args.slab_bytes,
                   "Single MI355X: Mgzip 1 MiB blocks, level 3, 4 GiB /dev/urandom-seeded ASCII")
        env.close()
        return
    if args.workload == "bgzf3":
        run_config(args, env, _native.FORMAT_BGZF, args.level, BLOCK, "text", args.slab_bytes,
                   "BGZF compress MiB/s at level 3 (gzp's default level), 550 MiB text" if args.level == 3 else
                   "BGZF compress MiB/s at level %d, 550 MiB text" % args.level)
        env.close()
        return

    # One logical stream of world x 550 MiB, sharded at block boundaries (weak scaling: every
    # rank gets 550 MiB +- one block); only the rank that owns the final block emits the tail.
    from gzp_amd import shard
    total = args.slab_bytes * world
    lo, n = shard.shard_bytes(total, BLOCK, world)[rank]
    mode = shard.slab_mode(rank, world, total, BLOCK)
    slab = synth.text_slab(n, seed=20250927 + rank)
    d_in = torch.from_numpy(slab).to(env.dev)
    ctx = _native.Context(format=_native.FORMAT_BGZF, level=1, buffer_size=BLOCK,
                          compat=_native.COMPAT_1_24, device=env.device_index, max_slab_bytes=n, lib=env.lib)
    cap = ctx.slab_bound(n)
    # two output buffers: with N > 1 the write-out of one step's shard (asynchronous) runs while the
    # next step compresses into the other buffer, the way ParCompress's slabs overlap
    d_outs = [torch.empty(cap, dtype=torch.uint8, device=env.dev) for _ in range(2 if world > 1 else 1)]
    # N > 1: BOTH in-order write-outs are timed in this one run, back to back -- first the one --writeout
    # names (default rccl: north_star's ordered gather), then the other; the line's `value` is the faster one.
    modes = [None] if world == 1 else [args.writeout] + [m for m in ("rccl", "offsets") if m != args.writeout]
    gathered = torch.empty(cap * world, dtype=torch.uint8, device=env.dev) if (world > 1 and rank == 0) else None
    # (a third one -- the writer's buffer IPC-mapped into every rank -- is timed LAST, behind everything else the line
    # carries and under a watchdog: see peer_leg below)
    peer_box = {"win": None}
    host_out = None
    if world > 1 and not env.emulate:
        host_out = [torch.empty(cap, dtype=torch.uint8).pin_memory() for _ in range(2)]
        copy_stream = torch.cuda.Stream()
    nb = ctx.n_blocks(n)
    block_sizes = np.zeros(nb, dtype=np.uint32)
    # the timed region carries HIP events around the dominant kernel only (two markers per step; its duration is
    # the roofline figure); the other stages are timed in a few steps of their own behind it
    ctx.set_profiling(2)
    if args.debug_flags:
        ctx.debug_set_flags(args.debug_flags)
    state = {"i": 0, "pending": None, "offsets": None, "writeout": modes[0], "wait_s": 0.0, "views": {}}

    def wait_pending():
        if state["pending"] is not None:
            t = time.perf_counter()
            view = state["pending"].wait()  # (rank 0, rccl / peer: the whole stream in order, a view of the writer's buffer)
            state["wait_s"] += time.perf_counter() - t  # host time spent waiting for a shard to leave
            state["pending"] = None
            if view is not None:
                state["views"][state["writeout"]] = view

    def step():
        k = state["i"] % len(d_outs)
        buf = d_outs[k]
        state["i"] += 1
        out_len, _ = ctx.compress_slab_device(d_in.data_ptr(), n, buf.data_ptr(), cap, mode,
                                              None, block_sizes)
        if world > 1:
            wait_pending()  # the previous shard has left (it travelled while this step compressed)
            if state["writeout"] == "peer":
                state["pending"] = peer_box["win"].gather_start(buf[:out_len])
            elif state["writeout"] == "rccl":
                # in-order write-out: ordered variable-size gather of the shards to rank 0 (RCCL),
                # started now and completed while the next step compresses
                state["pending"] = shard.ordered_gather_start(buf[:out_len], dst=0, out=gathered)
            else:
                # only the 8-byte sizes cross the fabric: their exclusive scan is every rank's offset
                # in the output stream, and each rank moves its own shard to the host side (where the
                # writer would pwrite it at that offset) over its own PCIe link
                state["offsets"] = shard.stream_offsets(out_len, env.dev)
                if host_out is not None:
                    copy_stream.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(copy_stream):
                        host_out[k][:out_len].copy_(buf[:out_len], non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(copy_stream)
                    state["pending"] = shard.EventHandle(ev)
        return out_len

    def timed_region(writeout):
        """W warm-up steps, then exactly K timed steps between barriers; max over ranks."""
        state["writeout"] = writeout
        acc = {}
        for _ in range(args.warmup):
            step()
        wait_pending()
        state["wait_s"] = 0.0
        env.sync()
        t0 = time.perf_counter()
        out_len = 0
        for _ in range(args.steps):
            out_len = step()
            for k, v in ctx.last_stage_ms().items():
                acc[k] = acc.get(k, 0.0) + v
        wait_pending()  # the last write-out belongs to the timed region
        env.sync()
        mine = time.perf_counter() - t0
        per_rank = env.all_ranks(mine / args.steps * 1e3)
        waits = env.all_ranks(state["wait_s"] / args.steps * 1e3)
        return {"dt": max(per_rank) * args.steps / 1e3 if world > 1 else mine, "out_len": out_len, "stage_acc": acc,
                "rank_ms_per_step": [round(x, 3) for x in per_rank], "rank_writeout_wait_ms": [round(x, 3) for x in waits]}

    regions = {}
    for wmode in modes:
        regions[wmode] = timed_region(wmode)
    # N > 1: the line's `value` is north_star's write-out -- the ordered RCCL gather of the compressed shards to rank 0 over
    # xGMI (round 5; rounds 3-4 reported the fastest of the write-outs, one of which moves no payload between GPUs).  The
    # others stay beside it: `writeouts` carries each with its per-rank times, `value_rccl` / `value_offsets` / `value_peer`
    # repeat them at the top level.
    value_mode = None if world == 1 else "rccl"
    head = regions[value_mode]
    dt, out_len, stage_acc = head["dt"], head["out_len"], head["stage_acc"]
    # every stage, outside the timed region (thirteen event markers per step cost the step 0.03 ms) -- under the
    # write-out `value` was measured with
    ctx.set_profiling(1)
    state["writeout"] = value_mode
    other_steps = min(3, args.steps)
    stage_other = {}
    for _ in range(other_steps):
        step()
        for k, v in ctx.last_stage_ms().items():
            stage_other[k] = stage_other.get(k, 0.0) + v / other_steps
    wait_pending()
    env.sync()
    ctx.set_profiling(2)

    # N > 1, strong scaling: THE 550 MiB slab of the metric (seed 20250927) cut into N contiguous block ranges, every rank
    # compresses its range, the ordered RCCL gather puts the stream together on rank 0 -- 550 MiB per step whatever N is.
    # The gathered stream is compared with the libdeflate-made digest of the whole stream (tests/golden/fullsize.json).
    strong = None
    rccl_stream = None
    if world > 1:
        if rank == 0 and "rccl" in state["views"]:  # (the strong leg gathers into the same buffer: keep this stream for the
            rccl_stream = state["views"]["rccl"].clone()  # comparison with the peer window's)
        strong = strong_leg(args, env, ctx, cap, gathered)

    ms_per_step = dt / args.steps * 1e3
    total_mib = total / 2**20
    value = total_mib / (dt / args.steps)
    scaling = "weak"
    if world > 1 and args.scaling == "strong" and rank == 0:  # (asked for: the strong figure is the line's value)
        value, ms_per_step, scaling = strong["MiBps"], strong["ms_per_step"], "strong"

    if rank == 0:
        last_buf = d_outs[(state["i"] - 1) % len(d_outs)]
        out_host = last_buf[:out_len].cpu().numpy()
        ok = verify(slab, out_host, block_sizes, tail=(mode == _native.SLAB_LAST))
        # every block, not a sample: the whole stream against the digest of the libdeflate-made one
        full_ok, stream_sha = (None, hashlib.sha256(out_host).hexdigest())
        if world == 1 and not env.emulate:
            full_ok, stream_sha = check_full_stream("config2_text_550MiB_bgzf_l1", n, 20250927, out_host, block_sizes)
        live = {k: v / args.steps for k, v in stage_acc.items() if v > 0}  # the timed region's own events: the dominant kernel
        stage_ms = dict(stage_other)
        stage_ms.update(live)
        dom = max(stage_ms, key=stage_ms.get)
        if dom not in live:  # (not expected: the match stage dominates at every level)
            live = {dom: stage_other[dom]}
        alg_bytes = n + out_len  # 1 B read + r B written per input byte
        achieved = alg_bytes / (max(stage_ms[dom], 1e-9) * 1e-3) / 1e9
        build_id = env.lib.build_id()
        traffic = pmc_traffic(dom, build_id)
        traffic_all = pmc_traffic("pipeline", build_id)
        issue = issue_roofline(dom, stage_ms[dom], build_id)
        res = {
            "metric": "BGZF compress MiB/s at level 1, 550 MiB text",
            "value": round(value, 1),
            "unit": "MiB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic" if not env.emulate else "synthetic (CPU emulator dry run: not a measurement)",
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
                "parallelism": "block-shard x%d%s" % (
                    world, "" if world == 1 else
                    " + ordered RCCL gather of the compressed shards to rank 0"),
                "verified_bit_exact_sample": bool(ok),
                "verified_bit_exact_full": full_ok,  # all blocks: SHA-256 of the stream and of the framed sizes == tests/golden/fullsize.json
                # parity, machine-readable: the rules in force in THIS run, and the libdeflate binary every golden digest
                # and vector is pinned on (the 1.24 gzp locks is not in the image; the two rules that differ are written
                # from SURVEY A.7 and never fire on this input -- tests/test_gpu_fullstream.py compares the two modes)
                "compat_in_force": "libdeflate 1.10" if ctx.active_compat() == _native.COMPAT_1_10 else "libdeflate >= 1.1x (1.24)",
                "compat_pinned": "1.10",
                "compat": "libdeflate >= 1.1x rule (the pinned 1.24); the golden digest is the v1.10 binary's, whose "
                          "stream is the same on this input",
                "box_libdeflate": box_libdeflate() if not env.emulate else None,
                "blocks_handed_back_to_dense_kernels": ctx.debug_redo_count(),
                "stream_sha256": stream_sha,
                "device": ctx.device_name(),
            },
            "compat_pinned": "1.10",  # the libdeflate binary every golden vector / digest comes from (config.compat_in_force: this run's rules)
            "roofline": {
                "bound": "hbm",
                "kernel": dom,
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "traffic_ratio": round(traffic / alg_bytes, 2) if traffic else None,  # this kernel's HBM bytes / the path's algorithmic bytes
                # what the kernel really moves over its own duration, as a fraction of the peak (it is issue-bound)
                "own_traffic_frac": round(traffic / (max(stage_ms[dom], 1e-9) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None,
                "traffic_pipeline": traffic_all,
                "traffic_ratio_pipeline": round(traffic_all / alg_bytes, 2) if traffic_all else None,
                "traffic_source": pmc_traffic_source(build_id),
                "issue_frac": issue.get("issue_frac"),  # VALU-busy share of the kernel's time (see issue_roofline)
                "issue": issue,
                "library_build_id": build_id,
                # the whole timed step (kernels, launch gaps, the side stream's join), not a sum of stages
                "pipeline_frac": round(alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "hbm_read_frac": round(n / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),  # input bytes only (north_star's wording)
                "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
                "stage_ms_source": "%s: HIP events inside the timed region (%d steps); the other stages: %d steps behind it"
                                   % (dom, args.steps, other_steps),
            },
        }
        if world > 1:
            res["writeouts"] = {
                m: {"MiBps": round(total_mib / (r["dt"] / args.steps), 1), "ms_per_step": round(r["dt"] / args.steps * 1e3, 3),
                    "rank_ms_per_step": r["rank_ms_per_step"], "rank_writeout_wait_ms": r["rank_writeout_wait_ms"]}
                for m, r in regions.items()}
            res["writeouts"]["value_is"] = value_mode if scaling == "weak" else "strong_550MiB (rccl)"
            res["writeouts"]["fastest"] = min(regions, key=lambda m: regions[m]["dt"])
            res["strong_550MiB"] = strong
            for m, r in regions.items():
                res["value_" + m] = round(total_mib / (r["dt"] / args.steps), 1)
        if world == 1 and not args.no_extras:
            ctx.set_profiling(False)
            try:
                res["e2e"] = e2e_legs(env, slab, res["config"]["stream_sha256"])
            except Exception as e:  # the headline line must survive a failing extra
                res["e2e"] = {"error": repr(e)}
            try:
                inf = run_inflate(args, env, emit=False, d_stream=last_buf, comp_host=out_host, slab=slab,
                                  steps=max(2, min(args.steps, 5)))
                res["inflate"] = {k: inf[k] for k in ("metric", "value", "unit", "ms_per_step", "roofline")}
                res["inflate"]["verified_round_trip"] = inf["config"]["verified_round_trip"]
                if "e2e" in inf:
                    res["inflate"]["e2e"] = inf["e2e"]
                if "cpu_baseline" in inf:
                    res["inflate"]["cpu_baseline"] = inf["cpu_baseline"]
            except Exception as e:
                res["inflate"] = {"error": repr(e)}
            if not env.emulate:  # (the emulator would take minutes per level: covered by tests/test_emu_levels.py)
                try:
                    res["levels"] = level_legs(env, d_in, n)
                except Exception as e:
                    res["levels"] = {"error": repr(e)}
                try:  # BASELINE configs[2] (the bench slab's buffers are released first)
                    del d_in, last_buf
                    d_outs.clear()
                    torch.cuda.empty_cache()
                    res["mgzip3"] = mgzip3_leg(env)
                except Exception as e:
                    res["mgzip3"] = {"error": repr(e)}
        if world == 1:
            # how much of this run the GPU was at work: the timed steps of every leg x their step time (kernels and launch
            # gaps; data generation, digests and the CPU legs are the rest of the run)
            act = (args.steps + args.warmup + other_steps) * ms_per_step * 1e-3
            inf = res.get("inflate", {})
            if "ms_per_step" in inf:
                act += (max(2, min(args.steps, 5)) + 2) * inf["ms_per_step"] * 1e-3
            for leg in res.get("levels", {}).values():
                if isinstance(leg, dict) and "ms_per_step" in leg:
                    act += (leg.get("steps", 1) + 1) * leg["ms_per_step"] * 1e-3
            mg = res.get("mgzip3", {})
            if "ms_per_step" in mg:
                act += (mg.get("steps", 1) + 1) * mg["ms_per_step"] * 1e-3 + 4 * mg.get("inflate_of_output", {}).get("k_inflate_ms", 0.0) * 1e-3
            res["gpu_active_s"] = round(act, 3)
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is timed at N = 1 only
            res["cpu_baseline"] = cpu_baseline(slab)
            try:  # ... and gzp's own orchestration around the same library (queues, 64 KiB writes, in-order writer)
                res["cpu_baseline_parcompress"] = cpu_baseline_parcompress(slab, res["config"]["stream_sha256"])
            except Exception as e:
                res["cpu_baseline_parcompress"] = {"error": repr(e)}
    # N > 1, the third in-order write-out, where the devices can map each other's memory: the writer's buffer IPC-mapped
    # into every rank, every rank copies its shard to its stream offset itself (copy engines over xGMI: no RCCL kernel on
    # any CU).  It runs LAST and under a watchdog: the line above is complete without it, and this is the one leg that no
    # box with a single GPU can rehearse -- if it does not finish, rank 0 prints the line it has (with `peer_error`) and
    # every rank leaves.
    hang_test = bool(os.environ.get("GZPX_BENCH_TEST_PEER_HANG")) and env.emulate  # (tests/test_bench_dryrun.py)
    if world > 1 and (not env.emulate or hang_test):
        dist.barrier()  # (rank 0 has just spent seconds checking its stream: every rank's clock starts here)
        dog = PeerWatchdog(float(os.environ.get("GZPX_BENCH_PEER_TIMEOUT", "90")), rank, res if rank == 0 else None)
        peer_err = None
        try:
            if hang_test:
                time.sleep(3600)
            peer_box["win"] = shard.PeerWindow(cap * world, env.dev, dst=0)
            ctx.set_profiling(2)
            r = timed_region("peer")
            if rank == 0:
                mib = round(total_mib / (r["dt"] / args.steps), 1)
                regions["peer"] = r
                res["writeouts"]["peer"] = {"MiBps": mib, "ms_per_step": round(r["dt"] / args.steps * 1e3, 3),
                                            "rank_ms_per_step": r["rank_ms_per_step"],
                                            "rank_writeout_wait_ms": r["rank_writeout_wait_ms"]}
                res["value_peer"] = mib
                res["writeouts"]["fastest"] = min(regions, key=lambda m: regions[m]["dt"])
                if rccl_stream is not None and "peer" in state["views"]:  # the window holds the stream RCCL delivered
                    a_ = state["views"]["peer"]
                    res["writeouts"]["peer"]["window_equals_rccl_stream"] = bool(a_.numel() == rccl_stream.numel() and
                                                                                 torch.equal(a_, rccl_stream))
        except Exception as e:  # (PeerWindow's set-up fails on every rank or on none)
            peer_err = repr(e)
        if not dog.stop():  # the watchdog has fired and is writing the line: it ends the process
            time.sleep(30)
        if peer_err is not None:
            if rank == 0:
                res["writeouts"]["peer_error"] = peer_err
                res["writeouts"]["peer_leg"] = "failed"
                print(json.dumps(res), flush=True)
            os._exit(0)  # (a rank that failed alone leaves the others inside a collective: nobody waits for a clean close)
    if rank == 0:
        print(json.dumps(res))
    ctx.close()
    env.close()


if __name__ == "__main__":
    main()

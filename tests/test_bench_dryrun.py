"""bench.py's control flow without GPUs: `python bench.py --gpus 2` from a bare shell must launch
itself under torch.distributed.run (the driver's 8-GPU run starts it exactly like that), shard the
stream, gather it in order and print ONE JSON line.  --emulate swaps the HIP library for the
CPU-emulated build and RCCL for gloo; nothing is measured."""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _run(*extra, **more_env):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(more_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--emulate", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline", "--no-extras"] + list(extra), capture_output=True, text=True, timeout=900,
                       env=env, cwd="/tmp")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_two_ranks_self_launch_ordered_gather():
    d = _run("--gpus", "2", "--slab-bytes", "150000")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["slab_bytes"] == 300000
    assert d["config"]["verified_bit_exact_sample"] is True
    for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype",
              "data", "roofline"):
        assert k in d
    # one run tells the whole story: both in-order write-outs timed back to back, per-rank times in the line
    w = d["writeouts"]
    assert w["value_is"] == "rccl" and w["fastest"] in ("rccl", "offsets")
    assert set(w) == {"rccl", "offsets", "value_is", "fastest"}  # (no IPC window on CPU)
    for m in ("rccl", "offsets"):
        assert len(w[m]["rank_ms_per_step"]) == 2 and len(w[m]["rank_writeout_wait_ms"]) == 2 and w[m]["MiBps"] > 0
        assert d["value_" + m] == w[m]["MiBps"]  # both write-outs at the top level as well
    # the line's value is north_star's write-out, the ordered RCCL gather -- whichever of them is faster (round 5)
    assert abs(w["rccl"]["ms_per_step"] - d["ms_per_step"]) < 1e-6 and d["value"] == d["value_rccl"]
    assert d["config"]["parallelism"].endswith("ordered RCCL gather of the compressed shards to rank 0")
    # ... and the metric's own slab at N ranks: strong scaling, the gathered stream equal to the single-process one
    st = d["strong_550MiB"]
    assert st["slab_bytes"] == 150000 and st["verified_bit_exact_full"] is True and len(st["rank_ms_per_step"]) == 2
    assert d["compat_pinned"] == "1.10" and d["config"]["compat_in_force"].startswith("libdeflate >= 1.1x")


def test_two_ranks_strong_scaling_is_the_value_when_asked_for():
    d = _run("--gpus", "2", "--slab-bytes", "150000", "--scaling", "strong")
    assert d["scaling"] == "strong" and d["value"] == d["strong_550MiB"]["MiBps"]
    assert d["strong_550MiB"]["verified_bit_exact_full"] is True and d["ms_per_step"] == d["strong_550MiB"]["ms_per_step"]


def test_a_hanging_peer_window_leg_does_not_take_the_line_with_it():
    """The IPC-window write-out is the one leg no single-GPU box can rehearse: it runs last and under a watchdog.  Here
    the leg is made to hang on every rank: rank 0 still prints the complete line (RCCL value, strong leg) with
    `peer_error`, every rank ends with exit code 0."""
    d = _run("--gpus", "2", "--slab-bytes", "150000", GZPX_BENCH_TEST_PEER_HANG="1", GZPX_BENCH_PEER_TIMEOUT="3")
    assert "did not finish within 3 s" in d["writeouts"]["peer_error"]
    assert d["value"] == d["value_rccl"] and d["strong_550MiB"]["verified_bit_exact_full"] is True
    assert "peer" not in d["writeouts"] and "value_peer" not in d


def test_eight_ranks_three_blocks():
    """configs[3]'s world size on the CPU: 8 ranks, a slab of three blocks -- five ranks own no block at all and take part
    in every gather with nothing; the strong-scaling leg's gathered stream is still the single-process stream."""
    d = _run("--gpus", "8", "--slab-bytes", "150000")
    assert d["n_gpus"] == 8 and d["value"] == d["value_rccl"] and len(d["writeouts"]["rccl"]["rank_ms_per_step"]) == 8
    st = d["strong_550MiB"]
    assert st["verified_bit_exact_full"] is True and st["shard_bytes_rank0"] == 65280 and len(st["rank_ms_per_step"]) == 8


def test_config4_workload_two_ranks():
    d = _run("--gpus", "2", "--workload", "fastq", "--stream-bytes", "400000")
    c = d["config"]
    assert d["scaling"] == "strong" and c["gpu_inflate_crc_roundtrip_ok"] is True
    assert c["gzip_t_prefix_suffix_rc"] in ([0, 0], None) and c["blocks"] == 7


def test_single_rank_line_has_e2e_and_inflate_legs(capsys, monkeypatch):
    # in-process (no second interpreter): the default line's extra legs
    sys.path.insert(0, ROOT)
    import bench
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--emulate", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                                      "--slab-bytes", "140000"])
    bench.main()
    d = json.loads([l for l in capsys.readouterr().out.splitlines() if l.startswith("{")][0])
    assert d["e2e"]["device_pinned_ok"] and d["e2e"]["api_write_ok"] and d["e2e"]["api_write_64k_ok"]
    assert d["inflate"]["verified_round_trip"] is True and d["inflate"]["roofline"]["kernel"] == "k_inflate_seg"

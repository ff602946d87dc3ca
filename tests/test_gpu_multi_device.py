"""The N > 1 paths on DISTINCT devices -- only a box with at least two GPUs can run these (the round-end GPU box
has one: they skip there; the first multi-GPU box runs them).  What they pin: hipDeviceEnablePeerAccess /
hipMemcpyPeerAsync of gzpx_multi_compress_slab_device across real device ordinals, the host-buffer multi-device
call on several devices, and `bench.py --gpus 2` exactly as the driver's scaling run starts it (self-launch under
torch.distributed.run, one rank per GPU over RCCL, both in-order write-outs, ONE JSON line).  The in-order property
they preserve is the writer loop's, /root/reference/src/par/compress.rs:305-310 (cited, not read at run time)."""
import json
import os
import subprocess
import sys

import pytest
import torch

import twin_cases as tc

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _n_gpus():
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


needs_two = pytest.mark.skipif(_n_gpus() < 2, reason="needs at least two GPUs (device_count() = %d)" % _n_gpus())


@needs_two
def test_multi_device_calls_on_distinct_devices(hip_lib, oracle):
    tc.multi_device_distinct(hip_lib, oracle, n_physical=min(_n_gpus(), 8))


@needs_two
def test_bench_two_gpus_as_the_driver_launches_it():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["verified_bit_exact_sample"] is True
    w = d["writeouts"]
    timed = [m for m in ("rccl", "offsets", "peer") if m in w]
    assert d["value"] == d["value_rccl"] and w["value_is"] == "rccl" and w["fastest"] in timed
    assert d["strong_550MiB"]["verified_bit_exact_full"] is True  # THE 550 MiB slab over two GPUs == libdeflate's stream
    assert w.get("peer", {}).get("window_equals_rccl_stream", True) is True
    assert "peer" in w or "peer_error" in w  # the copy-engine write-out ran, or the line says why not
    for m in timed:
        assert len(w[m]["rank_ms_per_step"]) == 2 and w[m]["MiBps"] > 0

"""The lines the driver's runs produce, checked on the GPU box by launching bench.py the way the driver
does (a subprocess from a bare shell): the default headline line with its self-proving keys, and
configs[3]'s workload (`--workload fastq`) on a 4 GiB share of the stream."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _bench(*args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_headline_line_proves_itself():
    d = _bench("--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu-baseline")
    c, r = d["config"], d["roofline"]
    assert d["metric"].startswith("BGZF compress MiB/s at level 1") and d["unit"] == "MiB/s" and d["n_gpus"] == 1
    assert c["verified_bit_exact_full"] is True and c["verified_bit_exact_sample"] is True  # all 8,835 blocks vs libdeflate's
    assert c["blocks"] == 8835 and c["blocks_handed_back_to_dense_kernels"] == 0
    assert r["kernel"] in r["stage_ms"] and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert 0 < r["pipeline_frac"] < r["frac"] and 0 < r["hbm_read_frac"] < r["pipeline_frac"]
    assert c["box_libdeflate"] in (None, "1.10-like", "1.24-like", "unknown")
    # the traffic figure is this build's or none at all (profiles/pmc_traffic.json records the build it belongs to)
    assert "issue_frac" in r and "source" in r["issue"]
    assert r["library_build_id"] and ((r["traffic"] is None) == ("refused" in r["traffic_source"] or "no profiles" in r["traffic_source"]))


def test_default_line_carries_the_other_configurations():
    """Everything the default `python bench.py` line promises beside the headline: the levels legs, configs[2] (Mgzip 1 MiB
    blocks, level 3, 4 GiB of ASCII noise) with its digest and the inflation of its own stream, configs[4] (`inflate`),
    the host-to-host legs and the CPU baseline -- every stream checked against the libdeflate-made digest inside the run."""
    d = _bench("--steps", "3", "--warmup", "1", timeout=1500)
    for lv in ("level_3", "level_6", "level_9"):
        leg = d["levels"][lv]
        assert leg["verified_bit_exact_full"] is True and leg["gpu_inflate_crc_roundtrip_ok"] is True and leg["MiBps"] > 0
    assert "level_12" not in d["levels"]  # (round 5: the near-optimal parser left the default line; --workload bgzf3 --level 12)
    assert d["compat_pinned"] == "1.10" and d["config"]["compat_in_force"].startswith("libdeflate >= 1.1x")
    m = d["mgzip3"]
    assert m["verified_bit_exact_full"] is True and m["gpu_inflate_crc_roundtrip_ok"] is True and m["blocks"] == 4096
    assert m["inflate_of_output"]["MiBps"] > 0 and m["roofline"]["kernel"] in m["roofline"]["stage_ms"]
    assert d["inflate"]["verified_round_trip"] is True and d["e2e"]["api_write_ok"] and d["e2e"]["device_pinned_ok"]
    assert d["cpu_baseline"]["kind"] in ("reference", "port") and d["cpu_baseline"]["cores"] >= 1
    pc = d["cpu_baseline_parcompress"]  # gzp's own orchestration over the box's libdeflate: its stream is the GPU's stream
    assert pc["value"] is None or (pc["value"] > 0 and pc["stream_equals_gpu_stream"] in (True, False))
    if pc["value"] is not None and d["config"]["box_libdeflate"] == "1.10-like":
        assert pc["stream_equals_gpu_stream"] is True


def test_config4_fastq_share_of_the_stream():
    d = _bench("--workload", "fastq", "--stream-bytes", str(4 << 30), "--steps", "1", "--warmup", "1", "--no-cpu-baseline")
    c = d["config"]
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["value"] > 0
    assert c["gpu_inflate_crc_roundtrip_ok"] is True
    assert c["gzip_t_prefix_suffix_rc"] in ([0, 0], None)

# round 5: the one-pass hash4 candidates kernel (k_candidates_h4, ds_mskor) against the two half passes (--debug-flags 64)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5_h4}; mkdir -p $O
for R in 1 2; do
for A in "--workload bgzf3 --level 3 --debug-flags 64" "--workload bgzf3 --level 3" "--workload bgzf3 --level 6 --debug-flags 64" "--workload bgzf3 --level 6" "--workload mgzip3 --debug-flags 64" "--workload mgzip3"; do
  timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras $A > $O/ab.json 2> $O/ab.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/ab.json").read().strip().splitlines()[-1])
    sm = d.get("roofline", {}).get("stage_ms") or {}
    print("run $R $A |", d["value"], d["ms_per_step"], {k: v for k, v in sm.items() if "match" in k or "cand" in k}, d["config"].get("gpu_inflate_crc_roundtrip_ok"), d["config"].get("stream_sha256", "")[:12])
except Exception as e:
    print("$A", "FAILED", e, open("$O/ab.err").read()[-600:])
PY
done
done
timeout 900 python -m pytest tests/test_gpu_levels.py tests/test_gpu_fullstream.py tests/test_gpu_fuzz_slice.py tests/test_gpu_orphan.py tests/test_gpu_fullsize.py -x -q -k "not near_optimal and not level_12 and not config4" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log

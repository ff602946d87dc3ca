# round 5: k_match_hc_sparse against the dense kernel -- phase clocks (experiment build), A/B of the product build, parity
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5_sparse}; mkdir -p $O
timeout 600 python tools/exp_hc_sparse2.py ${2:-3,4} > $O/exp.log 2>&1; tail -12 $O/exp.log
for A in "--workload bgzf3 --level 3 --debug-flags 16" "--workload bgzf3 --level 3" "--workload bgzf3 --level 2 --debug-flags 16" "--workload bgzf3 --level 2" "--workload bgzf3 --level 4 --debug-flags 16" "--workload bgzf3 --level 4" "--workload mgzip3 --debug-flags 16" "--workload mgzip3"; do
  timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras $A > $O/ab.json 2> $O/ab.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/ab.json").read().strip().splitlines()[-1])
    sm = d.get("roofline", {}).get("stage_ms") or {}
    print("$A |", d["value"], d["ms_per_step"], {k: v for k, v in sm.items() if "match" in k or "cand" in k}, d["config"].get("gpu_inflate_crc_roundtrip_ok"), d["config"].get("stream_sha256", "")[:12])
except Exception as e:
    print("$A", "FAILED", e, open("$O/ab.err").read()[-600:])
PY
done
timeout 900 python -m pytest tests/test_gpu_levels.py tests/test_gpu_fullstream.py tests/test_gpu_fuzz_slice.py tests/test_gpu_orphan.py tests/test_gpu_fullsize.py -x -q -k "not near_optimal and not level_12 and not config4" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log

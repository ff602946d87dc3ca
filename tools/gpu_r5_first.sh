# round 5, first call: the new tests first (so a failure shows early), then the whole suite, smoke and the default line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_first; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullstream.py::test_config4_all_eight_shares_and_the_whole_32gib_stream tests/test_gpu_peer_window.py tests/test_gpu_fuzz_slice.py tests/test_gpu_twin.py -x -q --durations=15 > $O/new.log 2>&1; echo "new rc=$?"; tail -25 $O/new.log
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -20 $O/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(json.dumps({k: d[k] for k in ("value","ms_per_step","compat_pinned")}), json.dumps(d["roofline"])[:1500])
for k in ("e2e","inflate","levels","mgzip3","cpu_baseline","cpu_baseline_parcompress"):
    print(k, json.dumps(d.get(k))[:700])
PY

# the round-end check the driver runs, plus the default bench line: tools/gpu_validate.sh <outdir under gpurun_out>
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-validate}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(json.dumps({k: d[k] for k in ("value","ms_per_step")}), json.dumps(d["roofline"]), json.dumps(d["config"]))
for k in ("e2e","inflate","levels","cpu_baseline"):
    print(k, json.dumps(d.get(k))[:900])
PY

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
python bench.py > gpurun_out/r2_bench_final.log 2>&1; tail -1 gpurun_out/r2_bench_final.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['stage_ms']); print(d['e2e']); print(d['inflate']['value']); print(d['levels']); print(d['cpu_baseline']['value'])"

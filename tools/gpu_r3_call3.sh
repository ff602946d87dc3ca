cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c3; mkdir -p $O
timeout 300 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_fused.json 2> $O/bench_fused.err; echo "fused rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3c3/bench_fused.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["stage_ms"], d["config"].get("blocks_handed_back_to_dense_kernels"), d["config"]["stream_sha256"][:12])
PY
timeout 600 python tools/exp_mparse.py > $O/exp_mparse.log 2>&1; echo rc=$?; tail -5 $O/exp_mparse.log

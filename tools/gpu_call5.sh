cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_twin.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(d['e2e'])"

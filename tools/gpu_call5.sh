cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
for i in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('l1', d['value'], d['ms_per_step'], d['roofline']['stage_ms'])"; done

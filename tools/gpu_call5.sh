cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_levels.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
for w in "bgzf3" "mgzip3 --steps 2"; do python bench.py --workload $w --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:40], d['value'], d['ms_per_step'], d['roofline']['stage_ms']['k_match'], d['config']['gpu_inflate_crc_roundtrip_ok'])"; done

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_levels.py tests/test_gpu_fullstream.py tests/test_gpu_parity.py -x -q 2>&1 | tail -2
python bench.py --workload bgzf3 --steps 5 --warmup 2 > gpurun_out/r2_bgzf3.log 2>&1; tail -1 gpurun_out/r2_bgzf3.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bgzf3', d['value'], d['ms_per_step'], d['roofline']['stage_ms'], d['config']['gpu_inflate_crc_roundtrip_ok'])"
python bench.py --workload mgzip3 --steps 3 --warmup 1 > gpurun_out/r2_mgzip3.log 2>&1; tail -1 gpurun_out/r2_mgzip3.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mgzip3', d['value'], d['ms_per_step'], d['roofline']['stage_ms'], d['config']['gpu_inflate_crc_roundtrip_ok'])"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('l1', d['value'], d['ms_per_step'], d['roofline']['stage_ms'])"

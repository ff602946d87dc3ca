cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_levels.py -x -q 2>&1 | tail -4
timeout 600 python tools/exp_levels.py 128 2>&1 | tee gpurun_out/r2_levels.log | tail -12

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_levels.py -x -q 2>&1 | tail -3
timeout 600 python tools/exp_levels.py 512 1,2,3,4,5,6,7,8,9 2>&1 | grep level | tee gpurun_out/r2_levels512b.log

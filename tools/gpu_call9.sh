cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python tools/gpu_fuzz.py 300 777 > gpurun_out/r2_soak_levels.log 2>&1
tail -3 gpurun_out/r2_soak_levels.log

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python tools/gpu_fuzz.py 100 99001 > gpurun_out/r2_soak_final.log 2>&1
tail -1 gpurun_out/r2_soak_final.log
timeout 120 python tools/gpu_fuzz_twin.py 50 99002 > gpurun_out/r2_soak_twin_final.log 2>&1
tail -1 gpurun_out/r2_soak_twin_final.log

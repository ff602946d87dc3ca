cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python tools/gpu_fuzz_twin.py 120 4242 > gpurun_out/r2_soak_twin.log 2>&1
tail -2 gpurun_out/r2_soak_twin.log
timeout 200 python tools/gpu_fuzz.py 120 31337 > gpurun_out/r2_soak_levels2.log 2>&1
tail -1 gpurun_out/r2_soak_levels2.log

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c2
timeout 600 python tools/exp_mparse.py > gpurun_out/r3c2/exp_mparse.log 2>&1; echo rc=$?; cat gpurun_out/r3c2/exp_mparse.log | tail -8

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 bash tools/profile_round.sh r02 2>&1 | tail -12

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/exp_hc_bounds.py 2>&1 | grep level | tee gpurun_out/r2_hc_bounds.log

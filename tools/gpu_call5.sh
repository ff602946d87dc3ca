cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time python bench.py ) > gpurun_out/r2_bench_default.log 2>&1
tail -5 gpurun_out/r2_bench_default.log | cut -c1-3000

set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time python -m pytest tests -m gpu -x -q --durations=15) > gpurun_out/r2_pytest1.log 2>&1
tail -30 gpurun_out/r2_pytest1.log
(time python bench.py --steps 10 --warmup 3) > gpurun_out/r2_bench1.log 2>&1
tail -2 gpurun_out/r2_bench1.log | cut -c1-3000
(time python bench.py --workload fastq --stream-bytes 4294967296 --steps 3 --warmup 1) > gpurun_out/r2_fastq1.log 2>&1
tail -2 gpurun_out/r2_fastq1.log | cut -c1-1500

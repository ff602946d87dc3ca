cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time python -m pytest tests -m gpu -x -q --durations=8) > gpurun_out/r2_pytest2.log 2>&1
tail -14 gpurun_out/r2_pytest2.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
(time python bench.py --steps 20 --warmup 5) > gpurun_out/r2_bench2.log 2>&1
tail -4 gpurun_out/r2_bench2.log | cut -c1-6000
python bench.py --workload bgzf3 --steps 5 --warmup 2 > gpurun_out/r2_bgzf3.log 2>&1; tail -1 gpurun_out/r2_bgzf3.log | cut -c1-1200
python bench.py --workload mgzip3 --steps 3 --warmup 1 > gpurun_out/r2_mgzip3.log 2>&1; tail -1 gpurun_out/r2_mgzip3.log | cut -c1-1200
bash tools/profile_round.sh r02 2>&1 | tail -12

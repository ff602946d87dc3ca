cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_r02
rocprofv3 --kernel-trace --stats -d $O/prof_r02 -o r02 --output-format csv -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $O/prof_r02.log 2>&1
f=$(find $O/prof_r02 -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then head -9 "$f" | cut -d, -f1-4 | cut -c1-40,150-; fi

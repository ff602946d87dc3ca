cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for L in 6 9; do
rm -rf $R/gpurun_out/prof_x$L
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_x$L -o x --output-format csv -- python $R/bench.py --workload bgzf3 --level $L --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_x$L.log 2>&1
f=$(find $R/gpurun_out/prof_x$L -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then head -4 "$f" | cut -d, -f1-4 | cut -c1-60,200-; fi
grep -v "^[WIE]2026" $R/gpurun_out/prof_x$L.log | tail -1 | cut -c1-200
done

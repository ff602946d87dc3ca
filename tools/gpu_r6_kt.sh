# round 6: kernel trace of the inflate workload (decode / copy pair)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6_kt}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o inf --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload inflate --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/prof/inf_kernel_stats.csv")))
for r in rows[:8]:
    print("%-60s calls %3s avg %.3f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
tail -1 $O/prof.log | cut -c1-160

# (round 5) kernel durations of the mixed compositions by both routes: tools/gpu_r5_mixed_kt.sh <outdir>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
for W in thirds blocks; do
for F in 16 0; do
  T=${W}_$F
  timeout 120 rocprofv3 --kernel-trace --stats -d $O/$T -o x --output-format csv -- python $R/tools/exp_mixed_trace.py $W $F > $O/$T.log 2>&1
  tail -1 $O/$T.log
  python3 - <<PY
import csv, glob
for row in csv.DictReader(open(glob.glob("$O/$T/**/x_kernel_stats.csv", recursive=True)[0])):
    if "gzpx" in row["Name"] and float(row["AverageNs"]) * int(row["Calls"]) > 4e4:
        print("  $T", row["Name"].replace("void ", "").replace("gzpx::", "")[:24], row["Calls"], "avg %.3f min %.3f max %.3f ms" % (float(row["AverageNs"]) / 1e6, float(row["MinNs"]) / 1e6, float(row["MaxNs"]) / 1e6))
PY
  rm -rf $O/$T
done
done

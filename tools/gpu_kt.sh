# kernel durations (rocprofv3 --kernel-trace --stats) of library builds at one level: tools/gpu_kt.sh <outdir> <level> <lib> [<lib> ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; LV=$2; shift; shift; mkdir -p $O
for L in "$@"; do
  T=$(basename $L .so)
  rocprofv3 --kernel-trace --stats -d $O/$T -o x --output-format csv -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --workload bgzf3 --level $LV --lib $R/$L > $O/$T.log 2>&1
  python3 - <<PY
import csv, glob
for row in csv.DictReader(open(glob.glob("$O/$T/**/x_kernel_stats.csv", recursive=True)[0])):
    if any(k in row["Name"] for k in ("k_match_hc", "k_parse_hc", "k_parse_lazy")):
        print("$T level $LV", row["Name"].replace("void ", "").replace("gzpx::", "")[:22], row["Calls"], "avg %.3f ms" % (float(row["AverageNs"]) / 1e6))
PY
done

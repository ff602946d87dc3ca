# per-call kernel durations of one configs[2] step (1 GiB of it): which of k_parse_hc's rounds cost what
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_trace_mgzip; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O -o mg --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload mgzip3 --slab-bytes 1073741824 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/mg_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last occurrence of k_emit marks the end of the timed step; print the 40 kernels before it
last = max(i for i, n in enumerate(names) if "k_emit" in n)
t0 = int(rows[max(0, last - 22)]["Start_Timestamp"])
for r in rows[max(0, last - 22):last + 1]:
    print("%8.3f ms +%7.3f ms  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Kernel_Name"][:70]))
PY

# round 6: the decode / LZ-copy pair (k_inflate_seg + k_lzcopy) against k_inflate: suites, the bench stream, configs[2]'s
# stream, phase clocks, kernel trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6_inflate}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_decompress.py -x -q > $O/pytest_dec.log 2>&1; echo "pytest decompress rc=$?"; tail -3 $O/pytest_dec.log
for R in seg wave; do
  GZPX_INFLATE_ROUTE=$R timeout 300 python bench.py --workload inflate --steps 10 --warmup 2 --no-cpu-baseline > $O/inf_$R.json 2> $O/inf_$R.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/inf_$R.json").read().strip().splitlines()[-1])
    print("route=$R text stream:", d["value"], "MiB/s", d["ms_per_step"], "ms; inflate kernels", d["roofline"]["kernel_ms"], "ms; round trip", d["config"]["verified_round_trip"])
except Exception as e:
    print("route=$R FAILED", e, open("$O/inf_$R.err").read()[-800:])
PY
done
timeout 300 python tools/exp_inflate_seg.py > $O/phases.txt 2>&1; cat $O/phases.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o inf --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload inflate --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_profiles.py $O/prof 2>/dev/null | head -30 || find $O/prof -name "*stats*" | head
for R in seg wave; do
  GZPX_INFLATE_ROUTE=$R timeout 300 python bench.py --workload mgzip3 --slab-bytes 1073741824 --steps 2 --warmup 1 --no-cpu-baseline > $O/mg_$R.json 2> $O/mg_$R.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/mg_$R.json").read().strip().splitlines()[-1])
    print("route=$R configs[2]'s stream (1 GiB):", d["config"]["inflate_of_output"], d["config"]["gpu_inflate_crc_roundtrip_ok"])
except Exception as e:
    print("route=$R mgzip FAILED", e, open("$O/mg_$R.err").read()[-800:])
PY
done

# round 3, call 1: parity of the fused level-1 kernel on hardware + A/B against the dense pair + kernel stats
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c1
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" ; tail -3 $O/pytest.log
timeout 300 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_fused.json 2> $O/bench_fused.err; echo "fused rc=$?"
timeout 300 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --debug-flags 2 > $O/bench_dense.json 2> $O/bench_dense.err; echo "dense rc=$?"
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/r3c1"
for n in ("fused","dense"):
    try:
        d=json.loads(open(O+"/bench_%s.json"%n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["roofline"]["stage_ms"], d["config"].get("blocks_handed_back_to_dense_kernels"), d["config"]["stream_sha256"][:12])
    except Exception as e:
        print(n, "ERR", e)
PY
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r03 --output-format csv -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $O/prof.log 2>&1
f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp $f $O/kernel_stats.csv; head -12 "$f" | cut -d, -f1-4 | cut -c1-60; fi
find $O/prof -name '*.csv' ! -name '*kernel_stats.csv' -size +1M -delete

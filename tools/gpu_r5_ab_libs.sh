# A/B of library builds at levels 3 and 4 (550 MiB text): tools/gpu_r5_ab_libs.sh <outdir> <lib> [<lib> ...]
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
for R in 1 2; do
for L in "$@"; do
  for LV in 3 4; do
    timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --workload bgzf3 --level $LV --lib $L > $O/ab.json 2> $O/ab.err
    python - <<PY
import json
try:
    d = json.loads(open("$O/ab.json").read().strip().splitlines()[-1])
    sm = d["roofline"]["stage_ms"]
    print("run $R", "$L".split("/")[-1], "level $LV:", d["ms_per_step"], "ms; match+parse", sm.get("k_match_hc+k_parse_hc"), d["config"]["stream_sha256"][:10])
except Exception as e:
    print("$L level $LV FAILED", e, open("$O/ab.err").read()[-300:])
PY
  done
done
done

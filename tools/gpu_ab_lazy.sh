# A/B of library builds at the lazy levels (6 and 9), every stage time: tools/gpu_ab_lazy.sh <outdir> <lib> [<lib> ...]
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
for L in "$@"; do
  T=$(basename $L .so)
  for LV in ${LEVELS:-6 9}; do
    timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --workload bgzf3 --level $LV --lib $L > $O/$T.$LV.json 2> $O/$T.$LV.err
    python - <<PY
import json
try:
    d = json.loads(open("$O/$T.$LV.json").read().strip().splitlines()[-1])
    print("$T level $LV", d["value"], d["ms_per_step"], d.get("roofline", {}).get("stage_ms"), d["config"].get("stream_sha256_matches_libdeflate"))
except Exception as e:
    print("$T level $LV", "FAILED", e, open("$O/$T.$LV.err").read()[-400:])
PY
  done
done

# A/B of library builds at levels 3 / 6 / 9 (+ configs[2]): tools/gpu_ab_levels.sh <outdir> <lib> [<lib> ...]
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
for L in "$@"; do
  T=$(basename $L .so)
  for W in "bgzf3 --level 3" "bgzf3 --level 6" "bgzf3 --level 9" "mgzip3"; do
    N=$(echo $W | tr -d ' -')
    timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --workload $W --lib $L > $O/$T.$N.json 2> $O/$T.$N.err
    python - <<PY
import json
try:
    d = json.loads(open("$O/$T.$N.json").read().strip().splitlines()[-1])
    sm = d.get("roofline", {}).get("stage_ms") or {}
    print("$T $W", d["value"], d["ms_per_step"], {k: v for k, v in sm.items() if "match" in k or "cand" in k})
except Exception as e:
    print("$T $W", "FAILED", e, open("$O/$T.$N.err").read()[-400:])
PY
  done
done

# A/B of builds and switches in one call: tools/gpu_ab.sh <outdir> "<lib>|<bench args>" ["<lib>|<bench args>" ...]
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
i=0
for V in "$@"; do
  L=${V%%|*}; A=${V#*|}; i=$((i+1))
  timeout 600 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline $A --lib $L > $O/$i.json 2> $O/$i.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$i.json").read().strip().splitlines()[-1])
    print("$V", d["value"], d["ms_per_step"], d.get("roofline",{}).get("stage_ms"), d["config"].get("verified_bit_exact_full"))
except Exception as e:
    print("$V", "FAILED", e, open("$O/$i.err").read()[-400:])
PY
done

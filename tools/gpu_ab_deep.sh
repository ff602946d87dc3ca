# A/B of library builds at the deep lazy levels 7 / 8 / 9: tools/gpu_ab_deep.sh <outdir> <lib> [<lib> ...]
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
for L in "$@"; do
  T=$(basename $L .so)
  for LV in 7 8 9; do
    timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --workload bgzf3 --level $LV --lib $L > $O/$T.l$LV.json 2> $O/$T.l$LV.err
    python - <<PY
import json
try:
    d = json.loads(open("$O/$T.l$LV.json").read().strip().splitlines()[-1])
    sm = d.get("roofline", {}).get("stage_ms") or {}
    print("$T level $LV", d["value"], d["ms_per_step"], {k: v for k, v in sm.items() if "match" in k}, d["config"].get("gpu_inflate_crc_roundtrip_ok"))
except Exception as e:
    print("$T level $LV", "FAILED", e, open("$O/$T.l$LV.err").read()[-300:])
PY
  done
done

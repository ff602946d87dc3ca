# A/B of experiment builds of the library: tools/gpu_r3_variants.sh <outdir> "<bench args>" <lib> [<lib> ...]
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; A="$1"; shift; mkdir -p $O
for L in "$@"; do
  timeout 600 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline $A --lib $L > $O/$(basename $L).json 2> $O/$(basename $L).err
  python - <<PY
import json
d=json.loads(open("$O/$(basename $L).json").read().strip().splitlines()[-1])
print("$L", d["value"], d["ms_per_step"], d.get("roofline",{}).get("stage_ms"), d["config"].get("verified_bit_exact_full"), d["config"].get("verified_round_trip"))
PY
done

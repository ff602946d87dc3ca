# A/B of library builds on the two inflate streams (the bench text's BGZF stream; configs[2]'s Mgzip stream):
#   tools/gpu_inflate_ab.sh <outdir> <lib> [<lib> ...]
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
for L in "$@"; do
  T=$(basename $L .so)
  timeout 300 python bench.py --workload inflate --steps 5 --warmup 1 --no-cpu-baseline --lib $L > $O/$T.inflate.json 2> $O/$T.inflate.err
  timeout 400 python bench.py --workload mgzip3 --steps 2 --warmup 1 --no-cpu-baseline --lib $L > $O/$T.mgzip3.json 2> $O/$T.mgzip3.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/$T.inflate.json").read().strip().splitlines()[-1])
    print("$T text stream: inflate", d["value"], "MiB/s", d["ms_per_step"], "ms; k_inflate", d["roofline"].get("kernel_ms"), "ms", d["config"].get("verified_round_trip"))
except Exception as e:
    print("$T inflate FAILED", e, open("$O/$T.inflate.err").read()[-300:])
try:
    d = json.loads(open("$O/$T.mgzip3.json").read().strip().splitlines()[-1])
    print("$T configs[2] stream: compress", d["value"], "inflate_of_output", d["config"]["inflate_of_output"], d["config"]["gpu_inflate_crc_roundtrip_ok"])
except Exception as e:
    print("$T mgzip3 FAILED", e, open("$O/$T.mgzip3.err").read()[-300:])
PY
done

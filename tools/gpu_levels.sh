# Levels 3 / 6 / 9 on the bench text and configs[2], stage times in one call: tools/gpu_levels.sh <outdir> [levels...]
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
LV=${@:-3 6 9}
for L in $LV; do
  timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --workload bgzf3 --level $L > $O/bgzf$L.json 2> $O/bgzf$L.err
done
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --workload mgzip3 > $O/mgzip3.json 2> $O/mgzip3.err
python - <<PY
import json
for name in [*("bgzf%s" % l for l in "$LV".split()), "mgzip3"]:
    try:
        d = json.loads(open("$O/%s.json" % name).read().strip().splitlines()[-1])
        print(name, d["value"], d["ms_per_step"], d.get("roofline", {}).get("stage_ms"), d["config"].get("verified_bit_exact_full"), d["config"].get("stream_sha256_matches_libdeflate"))
    except Exception as e:
        print(name, "FAILED", e, open("$O/%s.err" % name).read()[-600:])
PY

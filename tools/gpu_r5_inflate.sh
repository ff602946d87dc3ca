# round 5: k_inflate's symbol loops side by side (GZPX_INFLATE_SEG = 0 / 64 / 128): the bench stream, configs[2]'s own stream,
# and the decompression suites with each
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5_inflate}; mkdir -p $O
for S in ${2:-0 64 128}; do  # (the 64 / 128 variants live in commit ea48e3c only: measured slower, taken out again)
  GZPX_INFLATE_SEG=$S timeout 300 python bench.py --workload inflate --steps 5 --warmup 2 --no-cpu-baseline > $O/inf_$S.json 2> $O/inf_$S.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/inf_$S.json").read().strip().splitlines()[-1])
    print("SEG=$S text stream:", d["value"], "MiB/s", d["ms_per_step"], "ms; k_inflate", d["roofline"]["kernel_ms"], "ms; round trip", d["config"]["verified_round_trip"])
except Exception as e:
    print("SEG=$S FAILED", e, open("$O/inf_$S.err").read()[-500:])
PY
  GZPX_INFLATE_SEG=$S timeout 300 python bench.py --workload mgzip3 --slab-bytes 1073741824 --steps 2 --warmup 1 --no-cpu-baseline > $O/mg_$S.json 2> $O/mg_$S.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/mg_$S.json").read().strip().splitlines()[-1])
    print("SEG=$S configs[2]'s stream (1 GiB):", d["config"]["inflate_of_output"], d["config"]["gpu_inflate_crc_roundtrip_ok"])
except Exception as e:
    print("SEG=$S mgzip FAILED", e, open("$O/mg_$S.err").read()[-500:])
PY
done
for S in ${3:-64 128}; do
  GZPX_INFLATE_SEG=$S timeout 600 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_twin.py tests/test_gpu_par.py -x -q > $O/pytest_$S.log 2>&1; echo "SEG=$S pytest rc=$?"; tail -2 $O/pytest_$S.log
  GZPX_INFLATE_SEG=$S timeout 200 python tools/gpu_fuzz_inflate.py 40 $((5 + S)) > $O/fuzz_$S.log 2>&1; tail -1 $O/fuzz_$S.log
done

# (round 5) k_match_hc_stale dealing stale blocks out in pieces: A/B against the build before (libgzpx_prev.so) on the
# compositions of tools/exp_mixed_hc.py, the parity suites of levels 2-4, the stale-path hunt on the GPU.
#   tools/gpu_r5_stale.sh <outdir>
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
for LV in 3 4; do
  timeout 120 python tools/exp_mixed_hc.py $LV gzp_amd/lib/libgzpx_prev.so > $O/mixed_l$LV.txt 2>&1; echo "mixed l$LV rc=$?"
done
grep -A2 "^mixed\|^blocks\|^text\|^random" $O/mixed_l3.txt | grep "level\|k_match_hc"
timeout 400 python -m pytest tests/test_gpu_levels.py tests/test_gpu_fuzz_slice.py tests/test_gpu_orphan.py -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
timeout 100 python tools/emu_fuzz_stale.py 70 901 --gpu > $O/stale_fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -2 $O/stale_fuzz.txt
timeout 100 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --workload bgzf3 --level 3 > $O/l3.json 2> $O/l3.err; echo "bgzf3 rc=$?"
python - <<PY
import json
d = json.loads(open("$O/l3.json").read().strip().splitlines()[-1])
print("level 3:", d["ms_per_step"], "ms", d["value"], d["config"].get("verified_bit_exact_full"), d["roofline"]["stage_ms"])
PY

# round 5, the closing calls.  tools/gpu_r5_final.sh soak | validate | profile
cd $GRAFT_REPO_ROOT
case "$1" in
soak)  # the new level 3-4 matcher under the random soak: default routing, and k_match_hc_sparse forced on every block
  O=gpurun_out/r5_soak; mkdir -p $O
  for F in 0 32; do
  python - <<PY > $O/l24_$F.log 2>&1
import sys
sys.path.insert(0, ".")
from tools import gpu_fuzz
from gzp_amd import _native
from oracle import oracle
c, bad = gpu_fuzz.fuzz(_native.load(), oracle, seed=20260928 + $F, secs=${2:-90}, min_level=2, max_level=4, verbose=True, debug_flags=$F)
print("levels 2-4 soak, debug flags $F: %d cases, %d failures" % (c, len(bad)), bad[:3])
PY
  tail -1 $O/l24_$F.log
  done
  python tools/gpu_soak_levels.py ${2:-90} 20260931 > $O/l29.log 2>&1; tail -1 $O/l29.log
  python tools/gpu_fuzz_twin.py 45 99 > $O/twin.log 2>&1; tail -1 $O/twin.log
  ;;
validate)  # what the driver runs at round end, plus the default line kept as profiles/r05_bench_line.json
  bash tools/gpu_validate.sh r5_validate
  ;;
profile)
  bash tools/profile_round.sh r05 > gpurun_out/profile_r05.log 2>&1; tail -8 gpurun_out/profile_r05.log
  bash tools/pmc_sq.sh pmc_sq_l1 > gpurun_out/pmc_sq_l1.log 2>&1; grep -E "k_mparse|k_candidates" gpurun_out/pmc_sq_l1.log | head -4
  bash tools/pmc_sq.sh pmc_sq_l3 "--workload bgzf3" > gpurun_out/pmc_sq_l3.log 2>&1; grep -E "k_match_hc" gpurun_out/pmc_sq_l3.log | head -4
  bash tools/pmc_sq.sh pmc_sq_inflate "--workload inflate" > gpurun_out/pmc_sq_inflate.log 2>&1; grep -E "k_inflate" gpurun_out/pmc_sq_inflate.log | head -2
  ;;
esac

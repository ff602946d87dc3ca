# round 6, the closing calls.  tools/gpu_r6_final.sh validate | profile
cd $GRAFT_REPO_ROOT
case "$1" in
validate)  # what the driver runs at round end, plus the default line kept as profiles/r06_bench_line.json
  bash tools/gpu_validate.sh r6_validate
  ;;
profile)
  rm -rf gpurun_out/prof_r06* gpurun_out/pmc_sq_l1 gpurun_out/pmc_sq_l3 gpurun_out/pmc_sq_inflate
  bash tools/profile_round.sh r06 > gpurun_out/profile_r06.log 2>&1; tail -8 gpurun_out/profile_r06.log
  bash tools/pmc_sq.sh pmc_sq_l1 > gpurun_out/pmc_sq_l1.log 2>&1; grep -E "k_mparse|k_candidates" gpurun_out/pmc_sq_l1.log | head -4
  bash tools/pmc_sq.sh pmc_sq_l3 "--workload bgzf3" > gpurun_out/pmc_sq_l3.log 2>&1; grep -E "k_match_hc" gpurun_out/pmc_sq_l3.log | head -4
  bash tools/pmc_sq.sh pmc_sq_inflate "--workload inflate" > gpurun_out/pmc_sq_inflate.log 2>&1; grep -E "k_inflate|k_lzcopy" gpurun_out/pmc_sq_inflate.log | head -4
  ;;
esac

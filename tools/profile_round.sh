#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>
# kernel-trace stats of the bench workloads (compress = headline, inflate, bgzf3 / mgzip3 for the hc
# kernels, the text slab at levels 6 and 9 for the lazy parsers) and the two PMC passes (FETCH_SIZE / WRITE_SIZE, each in its own run, never combined with
# other trace domains).  Summaries: tools/summarize_profiles.py.
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o $tag --output-format csv -- $B > $O/prof_$tag.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_inflate -o ${tag}_inflate --output-format csv -- $B --workload inflate > $O/prof_${tag}_inflate.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_bgzf3 -o ${tag}_bgzf3 --output-format csv -- $B --workload bgzf3 > $O/prof_${tag}_bgzf3.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_mgzip3 -o ${tag}_mgzip3 --output-format csv -- $B --workload mgzip3 --steps 2 > $O/prof_${tag}_mgzip3.log 2>&1
GZPX_INFLATE_ROUTE=wave rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_inflate_wave -o ${tag}_inflate_wave --output-format csv -- $B --workload inflate > $O/prof_${tag}_inflate_wave.log 2>&1
for L in 6 9 ${PROFILE_L12:+12}; do
  rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_bgzf$L -o ${tag}_bgzf$L --output-format csv -- $B --workload bgzf3 --level $L --steps 2 > $O/prof_${tag}_bgzf$L.log 2>&1
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- $B > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_inflate -o f --output-format csv -- $B --workload inflate > $O/pmc_fetch_inflate.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_inflate -o w --output-format csv -- $B --workload inflate > $O/pmc_write_inflate.log 2>&1
find $O -name "*_kernel_stats.csv" -newer $R/bench.py -o -name "*counter_collection.csv" -newer $R/bench.py | sort
for w in "" _inflate _bgzf3 _mgzip3 _bgzf6 _bgzf9; do tail -1 $O/prof_${tag}$w.log | cut -c1-400; done

#!/bin/bash
# the four HBM-traffic passes (FETCH_SIZE / WRITE_SIZE, compress and inflate workloads, each in its own run) -- what
# tools/summarize_profiles.py needs to refresh profiles/pmc_traffic.json for the current build of the library
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_fetch_inflate $O/pmc_write_inflate
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- $B > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_inflate -o f --output-format csv -- $B --workload inflate > $O/pmc_fetch_inflate.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_inflate -o w --output-format csv -- $B --workload inflate > $O/pmc_write_inflate.log 2>&1
tail -1 $O/pmc_write_inflate.log | cut -c1-200

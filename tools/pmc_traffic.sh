#!/bin/bash
# the two HBM-traffic passes of the compress workload only (FETCH_SIZE / WRITE_SIZE, separate runs)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- $B > $O/pmc_write.log 2>&1
tail -1 $O/pmc_write.log | cut -c1-300

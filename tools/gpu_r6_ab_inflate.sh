# A/B of library builds on the inflate workload (bench stream): tools/gpu_r6_ab_inflate.sh <lib> [<lib> ...]
cd $GRAFT_REPO_ROOT
for R in 1 2; do
for L in "$@"; do
  timeout -s KILL 120 python bench.py --workload inflate --steps 20 --warmup 3 --no-cpu-baseline --lib $L 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('run $R', '$L'.split('/')[-1], d['value'], 'MiB/s', d['ms_per_step'], 'ms; decode', r['kernel_ms'], 'copy', r.get('k_lzcopy_ms'), d['config']['verified_round_trip'], 'handed back', d['config'].get('handed_back_members'))"
done
done

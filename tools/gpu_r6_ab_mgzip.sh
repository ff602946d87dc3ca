# A/B of library builds on configs[2] (1 GiB of it) and level 3 of the text slab: tools/gpu_r6_ab_mgzip.sh <lib> [<lib> ...]
cd $GRAFT_REPO_ROOT
for L in "$@"; do
  timeout -s KILL 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --workload mgzip3 --slab-bytes 1073741824 --lib $L 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); sm = d['roofline']['stage_ms']
print('$L'.split('/')[-1], 'mgzip3 1GiB:', d['ms_per_step'], 'ms', {k: round(v, 2) for k, v in sm.items()}, d['config'].get('stream_sha256', '')[:10], d['config'].get('verified_bit_exact_full'))"
  timeout -s KILL 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --workload bgzf3 --level 3 --lib $L 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); sm = d['roofline']['stage_ms']
print('$L'.split('/')[-1], 'bgzf3 l3:', d['ms_per_step'], 'ms', {k: round(v, 2) for k, v in sm.items()}, d['config'].get('stream_sha256', '')[:10])"
done

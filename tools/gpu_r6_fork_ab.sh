cd $GRAFT_REPO_ROOT
for F in 0 256 0 256; do
  timeout -s KILL 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --workload mgzip3 --debug-flags $F 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); sm = d['roofline']['stage_ms']
print('mgzip3 4GiB flags $F', d['ms_per_step'], 'ms', {k: round(v, 2) for k, v in sm.items()}, d['config'].get('stream_sha256', '')[:10])"
done
for L in 3 4 2 6 9; do for F in 0 256; do
  timeout -s KILL 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --workload bgzf3 --level $L --debug-flags $F 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); sm = d['roofline']['stage_ms']
print('text level $L flags $F', d['ms_per_step'], 'ms', {k: round(v, 2) for k, v in sm.items() if v > 0.1})"
done; done

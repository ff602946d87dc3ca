cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/r2_bench_final.log 2>&1; tail -1 gpurun_out/r2_bench_final.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['stage_ms']); print(d['e2e']); print(d['inflate']['value']); print({k:(v['MiBps'],v['ms_per_step']) for k,v in d['levels'].items()}); print(d['cpu_baseline']['value'])"

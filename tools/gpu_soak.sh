# randomised parity soaks on the GPU box (not part of the pytest suites): tools/gpu_soak.sh <seconds per tool>
cd $GRAFT_REPO_ROOT
S=${1:-120}
mkdir -p gpurun_out/soak
python - <<PY > gpurun_out/soak/l1.log 2>&1
# level-1 heavy: the match-on-demand kernel and its hand-back route on every input class
import sys, time
sys.path.insert(0, ".")
from tools import gpu_fuzz
from gzp_amd import _native
from oracle import oracle
c, bad = gpu_fuzz.fuzz(_native.load(), oracle, seed=777, secs=$S, max_level=1, verbose=True)
print("level<=1 soak: %d cases, %d failures" % (c, len(bad)))
PY
tail -2 gpurun_out/soak/l1.log
python tools/gpu_fuzz.py $S 4242 > gpurun_out/soak/all.log 2>&1; tail -1 gpurun_out/soak/all.log
python tools/gpu_fuzz_twin.py $S 99 > gpurun_out/soak/twin.log 2>&1; tail -1 gpurun_out/soak/twin.log
python tools/gpu_fuzz_inflate.py $S 5 > gpurun_out/soak/inflate.log 2>&1; tail -1 gpurun_out/soak/inflate.log
python - <<PY > gpurun_out/soak/l1012.log 2>&1
# levels 10-12: the near-optimal parser (one lane per block: sizes kept below 400,000 bytes)
import sys
sys.path.insert(0, ".")
from tools import gpu_fuzz
from gzp_amd import _native
from oracle import oracle
c, bad = gpu_fuzz.fuzz(_native.load(), oracle, seed=1012, secs=$S, min_level=10, max_level=12, max_n=400000, verbose=True)
print("levels 10-12 soak: %d cases, %d failures" % (c, len(bad)))
PY
tail -2 gpurun_out/soak/l1012.log

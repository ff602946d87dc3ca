cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys
sys.path.insert(0, "tools")
import gpu_fuzz
from gzp_amd import _native
from oracle import oracle
oracle.build()
cases, bad = gpu_fuzz.fuzz(_native.load(), oracle, seed=20250929, max_cases=900, max_n=600_000, verbose=True)
print("cases", cases, "bad", bad)
PY

import sys
sys.path.insert(0, ".")
from tools import gpu_fuzz
from gzp_amd import _native
from oracle import oracle
secs = float(sys.argv[1]); seed = int(sys.argv[2])
c, bad = gpu_fuzz.fuzz(_native.load(), oracle, seed=seed, secs=secs, min_level=2, max_level=9, verbose=False)
print("levels 2-9 soak (seed %d): %d cases, %d failures" % (seed, c, len(bad)), bad[:3])

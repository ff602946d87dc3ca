"""Host-side probe: how many hardware threads the container may really use (cgroup cpu.max) and how
the oracle port scales with worker threads.  Used to size bench.py's cpu_baseline."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
os.system("lscpu | egrep 'Model name|Socket|Core|Thread|MHz' | head -8")
from oracle import oracle
from gzp_amd import synth
oracle.build()
a = synth.text_slab(256<<20, seed=1)
for th in (1, 8, 32, 64, 128, 256):
    nb, dt, used = oracle.cpu_bench_compress(a, threads=th, wall_s=2.0)
    print("compress", used, "threads", round(nb/dt/2**20,1), "MiB/s")

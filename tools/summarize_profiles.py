"""Turn the rocprofv3 CSVs of a gpurun call (gpurun_out/) into the committed summaries under
profiles/: the --kernel-trace --stats table and the per-kernel HBM traffic from the two PMC
passes (FETCH_SIZE, WRITE_SIZE; separate runs, counters in KiB).

gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of a coalesced
streaming read; calibrated here on k_candidates, whose only HBM read is the input slab exactly
once (known byte count) -- the factor that makes that kernel read `slab_bytes` is applied to all.
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
slab_bytes = 576_716_800

os.makedirs(P, exist_ok=True)
stats = os.path.join(G, "prof_%s" % tag, "%s_kernel_stats.csv" % tag)
shutil.copy(stats, os.path.join(P, "%s_kernel_stats.csv" % tag))


def counters(path):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"].split("(")[0].replace("gzpx::", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items() if k.startswith("k_")}


fetch = counters(os.path.join(G, "pmc_fetch", "f_counter_collection.csv"))
write = counters(os.path.join(G, "pmc_write", "w_counter_collection.csv"))
cal = slab_bytes / (fetch["k_candidates"] * 1024.0)
doc = {
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py slab",
    "units": "bytes per launch (counter KiB x 1024)",
    "fetch_calibration_factor": round(cal, 3),
    "fetch_calibration": "k_candidates reads the %d-byte slab exactly once" % slab_bytes,
    "raw_fetch_kib": fetch,
    "raw_write_kib": write,
    "hbm_bytes_per_launch": {k: int(fetch.get(k, 0) * 1024 * cal + write.get(k, 0) * 1024) for k in fetch},
}
with open(os.path.join(P, "pmc_traffic.json"), "w") as f:
    json.dump(doc, f, indent=1)
print(json.dumps(doc["hbm_bytes_per_launch"], indent=1))
print("calibration factor", cal)

#!/bin/bash
# SQ counters of the bench workloads (two passes of <= 8 SQ counters each; counters only with --kernel-trace)
#   tools/pmc_sq.sh <outdir-under-gpurun_out> ["<extra bench args>"]
# prints the per-launch table and writes <outdir>/sq_counters.json (workload -> kernel -> counter, with the build id
# of the library that ran); tools/summarize_profiles.py merges those into profiles/sq_counters.json, which
# bench.py's `roofline.issue` reads.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pmc_sq}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras $2"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/a -o a --output-format csv -- $B > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/b -o b --output-format csv -- $B > $O/b.log 2>&1
python3 - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, "$R")
from gzp_amd import build as _gbuild
doc = {"workload": "bench.py $2".strip(), "build_id": _gbuild.source_id(), "kernels": {}}
for tag in ("a","b"):
    fs = glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs: print(tag, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for row in csv.DictReader(open(fs[0])):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("gzpx::", "")[:32]
        acc[k][row["Counter_Name"].replace("SQ_", "")] += float(row["Counter_Value"])
        disp[k].add(row["Dispatch_Id"])
    for k, d in acc.items():  # per launch (the run holds len(disp[k]) launches of the kernel)
        print("%-28s launches=%-3d " % (k, len(disp[k])) + "  ".join("%s=%.3g" % (c, v / len(disp[k])) for c, v in sorted(d.items())))
        name = k if k.startswith("k_candidates") else k.split("<")[0]
        doc["kernels"].setdefault(name, {}).update({c: v / len(disp[k]) for c, v in d.items()})
json.dump(doc, open("$O/sq_counters.json", "w"), indent=1)
PY

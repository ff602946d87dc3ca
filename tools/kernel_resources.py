"""profiles/<tag>_kernel_resources.txt: registers, LDS, scratch and spills of every kernel of the product
library, from hipcc's -Rpass-analysis=kernel-resource-usage remarks (no GPU needed)."""
import os
import re
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
rows = []
for src in ("gzpx_kernels.hip", "gzpx_nearopt.hip", "gzpx_synth.hip", "gzpx_check.hip"):
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "gzp_amd", "csrc", src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True)
    cur = None
    for line in p.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(.*", "", name).replace("gzpx::", "").replace("void ", "")}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" [")[0]] = int(m.group(2))
out = os.path.join(ROOT, "profiles", "%s_kernel_resources.txt" % tag)
with open(out, "w") as f:
    f.write("# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage (tools/kernel_resources.py)\n")
    f.write("%-34s %5s %5s %8s %7s %6s %6s %9s\n" % ("kernel", "VGPR", "SGPR", "scratch", "spillV", "spillS", "occ", "LDS"))
    for r in rows:
        f.write("%-34s %5d %5d %8d %7d %6d %6d %9d\n" % (r["name"][:34], r.get("VGPRs", 0), r.get("TotalSGPRs", 0), r.get("ScratchSize", 0),
                                                           r.get("VGPRs Spill", 0), r.get("SGPRs Spill", 0), r.get("Occupancy", 0), r.get("LDS Size", 0)))
print(open(out).read())

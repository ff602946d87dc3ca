"""k_match_hc_sparse (round 5) next to the dense k_match_hc, and where its time goes.  -DGZPX_EXPERIMENT build (never the
product): thread 0's clock per phase, summed over the blocks of the launches; 550 MiB of text (BGZF, levels 2-4) and 1 GiB
of configs[2]'s ASCII noise (Mgzip, 1 MiB blocks, level 3); every stream compared with the dense route's.
    python tools/exp_hc_sparse2.py [levels, e.g. 3 or 2,3,4]"""
import ctypes
import hashlib
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gzp_amd import _native, build, synth

exp = os.path.join(build.LIB_DIR, "libgzpx_exp.so")
srcs = [os.path.join(build.CSRC, s) for s in build.SOURCES]
deps = srcs + [os.path.join(build.CSRC, f) for f in os.listdir(build.CSRC) if f.endswith((".h", ".hpp"))]
if not os.path.exists(exp) or os.path.getmtime(exp) < max(os.path.getmtime(d) for d in deps):  # (built in the container, it travels with the snapshot: no GPU minutes spent compiling)
    subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
                           "-DGZPX_EXPERIMENT", "-I", build.INCLUDE] + srcs + ["-o", exp], stderr=subprocess.DEVNULL)
if "--build-only" in sys.argv:
    sys.exit(0)
import torch  # noqa: E402  (behind --build-only: the container has no GPU)

lib = _native.GzpxLib(exp)
cyc = (ctypes.c_ulonglong * 8)()
names = ["window", "first nodes", "walks", "lists", "searches"]
levels = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 and sys.argv[1][0].isdigit() else [3]
for what, level in [("text", lv) for lv in levels] + [("ascii", 3)]:
    if what == "text":
        n, fmt, bs = 576_716_800, 0, 65280
        d_in = torch.from_numpy(synth.text_slab(n, seed=20250927)).cuda()
    else:
        n, fmt, bs = 1 << 30, 1, 1 << 20
        d_in = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
        _native.synth_ascii_device(d_in.data_ptr(), 0, n, 8, lib=lib)
    sha = {}
    for route, flags in (("dense", 16), ("sparse", 0)):
        ctx = _native.Context(format=fmt, level=level, buffer_size=bs, lib=lib, max_slab_bytes=n)
        ctx.debug_set_flags(flags)
        cap = ctx.slab_bound(n)
        d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        ctx.set_profiling(True)
        nb = ctx.n_blocks(n)
        acc = {}
        for it in range(4):
            if it == 1:
                lib.L.gzpx_exp_sparse(cyc, 1)
            out_len, _ = ctx.compress_slab_device(d_in.data_ptr(), n, d_out.data_ptr(), cap, True)
            if it:
                for k, v in ctx.last_stage_ms().items():
                    acc[k] = acc.get(k, 0.0) + v / 3
        lib.L.gzpx_exp_sparse(cyc, 1)
        sha[route] = hashlib.sha256(d_out[:out_len].cpu().numpy()).hexdigest()
        c = [x / 3 / nb for x in cyc]
        tot = sum(c[:5])
        line = "%s level %d %s: match+parse %.2f ms (experiment build: the clocks cost)" % (what, level, route, acc["k_match_hc+k_parse_hc"])
        if route == "sparse" and tot:
            line += "; cycles per block %.0f: " % tot + ", ".join("%s %.0f (%.0f%%)" % (names[k], c[k], 100 * c[k] / tot) for k in range(5))
            line += "; per block: %.1f rounds, %.0f listed searches (%.1f%% of the positions), %.1f tiles" % (
                c[5], c[6], 100.0 * c[6] * nb / n, c[7])
        print(line, flush=True)
        ctx.close()
        del d_out
    print("   streams equal:", sha["dense"] == sha["sparse"], flush=True)
    del d_in

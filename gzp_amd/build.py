"""Build gzp_amd/lib/libgzpx.so with hipcc for gfx950 (in-tree, so it travels with the repo)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIB_DIR, "libgzpx.so")
INCLUDE = os.path.abspath(os.path.join(_HERE, "..", "include"))
SOURCES = ["gzpx_kernels.hip", "gzpx_nearopt.hip", "gzpx_synth.hip", "gzpx_check.hip", "gzpx_api.cpp", "gzpx_par.cpp"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def source_id():
    """SHA-256 (16 hex digits) over the sources the library is built from, in a fixed order: the library carries it
    (gzpx_build_id) and profiles/pmc_traffic.json records the one it was collected with, so that bench.py can tell a
    committed traffic figure of another build from one of this build."""
    import hashlib
    h = hashlib.sha256()
    names = sorted(os.listdir(CSRC))
    for path in [os.path.join(CSRC, f) for f in names if f.endswith((".hip", ".cpp", ".h", ".hpp"))] + [os.path.join(INCLUDE, "gzpx.h")]:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    deps.append(os.path.join(INCLUDE, "gzpx.h"))
    if (not force and os.path.exists(LIB)
            and os.path.getmtime(LIB) >= max(os.path.getmtime(d) for d in deps)):
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    # (one builder at a time, into a temporary name renamed when complete: concurrent importers never load half a file)
    import fcntl
    with open(LIB + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if (not force and os.path.exists(LIB)
                and os.path.getmtime(LIB) >= max(os.path.getmtime(d) for d in deps)):
            return LIB
        tmp = LIB + ".tmp.%d" % os.getpid()
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
               "-DGZPX_BUILD_ID=\"%s\"" % source_id(), "-I", INCLUDE] + srcs + ["-o", tmp]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))

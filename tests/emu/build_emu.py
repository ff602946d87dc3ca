"""Build tests/emu/libgzpx_emu.so: the product sources compiled with g++ against the CPU SIMT
emulator (tests/emu/hip/hip_runtime.h).  TEST INFRASTRUCTURE ONLY -- gzp_amd never loads it."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
CSRC = os.path.join(ROOT, "gzp_amd", "csrc")
OUT = os.path.join(HERE, "libgzpx_emu.so")
SOURCES = [os.path.join(CSRC, "gzpx_kernels.hip"), os.path.join(CSRC, "gzpx_nearopt.hip"), os.path.join(CSRC, "gzpx_synth.hip"), os.path.join(CSRC, "gzpx_check.hip"),
           os.path.join(CSRC, "gzpx_api.cpp"),
           os.path.join(CSRC, "gzpx_par.cpp"), os.path.join(HERE, "emu_runtime.cpp")]


def build(force=False):
    srcs = [s for s in SOURCES if os.path.exists(s)]
    deps = srcs + [os.path.join(CSRC, "gzpx_device.h"), os.path.join(ROOT, "include", "gzpx.h"),
                   os.path.join(HERE, "hip", "hip_runtime.h")]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    if (not force and os.path.exists(OUT)
            and os.path.getmtime(OUT) >= max(os.path.getmtime(d) for d in deps if os.path.exists(d))):
        return OUT
    # Several processes may want the library at once (the two ranks of `bench.py --emulate --gpus 2`): one builds, under a
    # file lock and into a temporary name that is renamed when complete; the others wait and find it up to date.
    import fcntl
    with open(OUT + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if (not force and os.path.exists(OUT)
                and os.path.getmtime(OUT) >= max(os.path.getmtime(d) for d in deps if os.path.exists(d))):
            return OUT
        tmp = OUT + ".tmp.%d" % os.getpid()
        cmd = ["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas",
               "-I", HERE, "-I", os.path.join(ROOT, "include"), "-x", "c++"] + srcs + ["-o", tmp]
        subprocess.check_call(cmd)
        os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="-f" in sys.argv))

"""Build a variant of the library with extra -D switches: python tools/build_variant.py <name> -DX=1 ...  ->  gzp_amd/lib/libgzpx_<name>.so
(measurement only; the product library is gzp_amd/lib/libgzpx.so)."""
import os, subprocess, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gzp_amd import build
name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(build.LIB_DIR, "libgzpx_%s.so" % name)
srcs = [os.path.join(build.CSRC, s) for s in build.SOURCES]
subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
                       "-DGZPX_BUILD_ID=\"%s\"" % build.source_id(), "-I", build.INCLUDE] + flags + srcs + ["-o", out], stderr=subprocess.DEVNULL)
print(out)

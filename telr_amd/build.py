"""Build telr_amd/libtelrhip.so (hipcc, gfx950 only) in-tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libtelrhip.so")
SOURCES = ["telr_engine.hip"]
DEPS = sorted(f for f in os.listdir(SRC) if f.endswith((".hip", ".h"))) + ["../../include/telr_hip.h"]          # every source and header under csrc/


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(os.path.join(SRC, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", SO] + \
          os.environ.get("TELR_EXTRA_FLAGS", "").split() + [os.path.join(SRC, s) for s in SOURCES] + ["-lz"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(SO)

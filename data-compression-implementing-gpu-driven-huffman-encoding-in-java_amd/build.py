"""Builds libdczhip.so (gfx950 code object + C ABI) in-tree with hipcc.  No GPU needed to build."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libdczhip.so")
SOURCES = ["dcz_api.hip", "k1_histogram.hip", "k2_codebuild.hip", "k3_encode.hip", "k4_decode.hip", "gen.hip"]
HEADERS = ["dcz_internal.h", os.path.join("..", "..", "include", "dcz.h")]


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    for f in SOURCES + HEADERS:
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    return False


def build(force=False, verbose=False):
    if not force and not _stale():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-o", SO] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(SO)

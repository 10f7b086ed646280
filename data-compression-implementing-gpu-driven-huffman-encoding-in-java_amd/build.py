"""Builds libdczhip.so (gfx950 code object + C ABI) in-tree with hipcc.  No GPU needed to build."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libdczhip.so")
SOURCES = ["dcz_api.hip", "k1_histogram.hip", "k2_codebuild.hip", "k3_encode.hip", "k4_decode.hip", "k5_sha256.hip", "gen.hip"]
HEADERS = ["dcz_internal.h", os.path.join("..", "..", "include", "dcz.h")]


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    for f in SOURCES + HEADERS:
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    return False


HOST_SOURCES = [os.path.join("host", "dcz_service.cpp"), os.path.join("host", "dcz_cli.cpp")]
CLI = os.path.join(HERE, "dczcli")


def build_host(force=False, verbose=False):
    """C++ host mirror of the reference's service seam + CLI (links libdczhip.so by rpath)."""
    srcs = [os.path.join(CSRC, f) for f in HOST_SOURCES]
    hdr = os.path.join(CSRC, "host", "dcz_service.h")
    if not force and os.path.exists(CLI) and all(os.path.getmtime(CLI) >= os.path.getmtime(f) for f in srcs + [hdr, SO]):
        return CLI
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc, "-O2", "-std=c++17", "-Wall", "-pthread", "-o", CLI] + srcs + ["-L" + HERE, "-ldczhip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return CLI


def build(force=False, verbose=False):
    if not force and not _stale():
        build_host(False, verbose)
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-o", SO] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    build_host(True, verbose)
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(SO)

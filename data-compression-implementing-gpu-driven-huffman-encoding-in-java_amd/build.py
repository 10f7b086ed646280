"""Builds libdczhip.so (gfx950 code object + C ABI) in-tree with hipcc.  No GPU needed to build."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libdczhip.so")
SOURCES = ["dcz_api.hip", "k1_histogram.hip", "k2_codebuild.hip", "k3_encode.hip", "k4_decode.hip", "k4_fixed.hip", "k4_split.hip", "k4_dfa.hip", "k5_sha256.hip", "gen.hip"]
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
    cmd = [hipcc, "-O2", "-std=c++17", "-Wall", "-pthread", "-o", CLI + ".part"] + srcs + ["-L" + HERE, "-ldczhip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(CLI + ".part", CLI)
    return CLI


def build(force=False, verbose=False, extra_flags=(), so=None, objdir=None):
    """Compile every .hip source to an object (in parallel, only the stale ones) and link libdczhip.so."""
    from concurrent.futures import ThreadPoolExecutor
    so = so or SO
    main = so == SO
    if main and not force and not _stale():
        build_host(False, verbose)
        return so
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    objdir = objdir or os.path.join(HERE, "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-Wall", "-Wno-unused-function"] + list(extra_flags)

    def compile_one(f):
        src = os.path.join(CSRC, f)
        obj = os.path.join(objdir, f.replace(".hip", ".o"))
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_t):
            return obj
        # (written under another name and moved into place: a build that is killed half-way must not leave a file that the
        #  time stamps call fresh -- a library left behind by an interrupted session took GPU exceptions in round 3)
        cmd = [hipcc] + flags + ["-c", src, "-o", obj + ".part.o"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        os.replace(obj + ".part.o", obj)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so + ".part"] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(so + ".part", so)
    if main:
        build_host(True, verbose)
    return so


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(SO)

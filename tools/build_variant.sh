#!/bin/bash
# usage: build_variant.sh <name> [-D...]   -> variants/lib_<name>.so (A/B runs through DCZ_LIB=...; variants/ is git-ignored)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift; mkdir -p "$R/variants"
python3 - "$R" "$N" "$@" <<'PY'
import sys, importlib.util, os
root, name, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
spec = importlib.util.spec_from_file_location("b", os.path.join(root, "data-compression-implementing-gpu-driven-huffman-encoding-in-java_amd", "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
print(b.build(force=True, extra_flags=flags, so=os.path.join(root, "variants", "lib_%s.so" % name), objdir=os.path.join(root, "variants", "obj_" + name)))
PY

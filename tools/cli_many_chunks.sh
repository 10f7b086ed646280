#!/bin/bash
# one-off check on the GPU box: the C++ CLI with >= 512 chunks per batch (K5 digests) against the Python mirror
set -e
cd $GRAFT_REPO_ROOT
P=data-compression-implementing-gpu-driven-huffman-encoding-in-java_amd
python - <<'PY'
import numpy as np
np.random.default_rng(5).integers(0, 200, size=1500 * (1 << 20) + 12345, dtype=np.uint8).tofile("/tmp/many.bin")
PY
$P/dczcli compress /tmp/many.bin /tmp/many.dcz 1 | tr "\r" "\n" | grep -v Progress
$P/dczcli verify /tmp/many.dcz
$P/dczcli decompress /tmp/many.dcz /tmp/many.out | tr "\r" "\n" | grep -v Progress
cmp /tmp/many.bin /tmp/many.out && echo "CLI round trip OK"
python - <<'PY'
import sys, hashlib
sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
blob = open("/tmp/many.dcz", "rb").read()
h, start = pkg.container.locate_header(blob)
data = open("/tmp/many.bin", "rb").read()
bad = sum(1 for c in h.chunks if c.sha256 != hashlib.sha256(data[c.original_offset:c.original_offset + c.original_size]).digest())
print("chunks", len(h.chunks), "digest mismatches vs hashlib", bad)
PY

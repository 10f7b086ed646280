#!/bin/bash
set -e
cd $GRAFT_REPO_ROOT
P=data-compression-implementing-gpu-driven-huffman-encoding-in-java_amd
$P/dczcli compress /tmp/many.bin /tmp/many32.dcz | tr "\r" "\n" | grep -v Progress | grep -E "Throughput|Checksum|Encoding|File"
$P/dczcli decompress /tmp/many32.dcz /tmp/many32.out | tr "\r" "\n" | grep -v Progress | grep -E "Throughput|Checksum|Decoding|File"
cmp /tmp/many.bin /tmp/many32.out && echo "CLI 32MB-chunk round trip OK"

#!/bin/bash
# HBM traffic of every dcz kernel: separate --pmc passes for FETCH_SIZE and WRITE_SIZE (TCC slots do not fit both)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; TAG=$1; shift
cd /tmp
for P in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/trf_${TAG}_$P -- python3 $R/bench.py --no-secondary "$@" --steps 3 --warmup 1 --cpu-sample-mib 0 --no-verify > $R/gpurun_out/trf_${TAG}_$P.json 2> $R/gpurun_out/trf_${TAG}_$P.err || echo "pass $P failed"
done
cd $R && python3 tools/pmcsum.py gpurun_out/trf_${TAG}_FETCH_SIZE gpurun_out/trf_${TAG}_WRITE_SIZE

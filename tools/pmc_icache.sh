#!/bin/bash
# usage: pmc_icache.sh <tag> <bench args...> -- instruction-cache and issue counters per kernel
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; TAG=$1; shift
cd /tmp
i=0
for P in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/pmci_${TAG}_$i -- python3 $R/bench.py --no-secondary "$@" --steps 2 --warmup 1 --cpu-sample-mib 0 --no-verify > /dev/null 2> $R/gpurun_out/pmci_${TAG}_$i.err || echo "pass $i failed"
done
cd $R && python3 tools/pmcsum.py gpurun_out/pmci_${TAG}_*/

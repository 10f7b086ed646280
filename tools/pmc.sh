#!/bin/bash
# usage: pmc.sh <tag> <bench args...>   -- two PMC passes, prints per-kernel means
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; TAG=$1; shift
cd /tmp
i=0
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/bench.py --no-secondary "$@" --steps 2 --warmup 1 --cpu-sample-mib 0 --no-verify > /dev/null 2> $R/gpurun_out/pmc_${TAG}_$i.err || echo "pass $i failed"
done
cd $R && python3 tools/pmcsum.py gpurun_out/pmc_${TAG}_*/

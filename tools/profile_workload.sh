#!/bin/bash
# usage: profile_workload.sh <tag> <bench args...>
# rocprofv3 evidence for one bench workload (run on the GPU box from the repo root): kernel-trace stats, HBM traffic
# (FETCH_SIZE / WRITE_SIZE in separate --pmc passes) and SQ counters.  Summaries land in gpurun_out/prof_<tag>_*.
set -o pipefail
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; TAG=$1; shift
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/st_$TAG -- python3 $R/bench.py --no-secondary "$@" --steps 20 --warmup 5 --cpu-sample-mib 0 > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_${TAG}_stats.err || echo "stats failed"
cp $(ls $R/gpurun_out/st_$TAG/*/*_kernel_stats.csv | head -1) $R/gpurun_out/prof_${TAG}_kernel_stats.csv || echo "no stats csv"
for P in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/trf_${TAG}_$P -- python3 $R/bench.py --no-secondary "$@" --steps 3 --warmup 1 --cpu-sample-mib 0 --no-verify > $R/gpurun_out/trf_${TAG}_$P.json 2> $R/gpurun_out/trf_${TAG}_$P.err || echo "pass $P failed"
done
i=0
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/bench.py --no-secondary "$@" --steps 2 --warmup 1 --cpu-sample-mib 0 --no-verify > /dev/null 2> $R/gpurun_out/pmc_${TAG}_$i.err || echo "sq pass $i failed"
done
cd $R
python3 tools/pmcsum.py gpurun_out/trf_${TAG}_FETCH_SIZE gpurun_out/trf_${TAG}_WRITE_SIZE > gpurun_out/prof_${TAG}_traffic.txt 2>&1
python3 tools/pmcsum.py gpurun_out/pmc_${TAG}_1 gpurun_out/pmc_${TAG}_2 > gpurun_out/prof_${TAG}_sq.txt 2>&1
# keep only the summaries (the raw per-dispatch csv files are large)
rm -rf gpurun_out/st_$TAG gpurun_out/trf_${TAG}_FETCH_SIZE gpurun_out/trf_${TAG}_WRITE_SIZE gpurun_out/pmc_${TAG}_1 gpurun_out/pmc_${TAG}_2
echo "profile $TAG done"

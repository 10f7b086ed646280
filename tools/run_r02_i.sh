#!/bin/bash
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
for w in lowentropy text; do
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/st_i_$w -- python3 $R/bench.py --workload $w --bytes-per-gpu 1073741824 --chunk-bytes 16777216 --steps 3 --warmup 1 --cpu-sample-mib 0 > $R/gpurun_out/i_$w.json 2> $R/gpurun_out/i_$w.err || echo "stats failed"
cp $(ls $R/gpurun_out/st_i_$w/*/*_kernel_stats.csv | head -1) $R/gpurun_out/i_${w}_kernel_stats.csv; rm -rf $R/gpurun_out/st_i_$w
done

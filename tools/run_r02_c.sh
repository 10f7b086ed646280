#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/c_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/c_tests.log
export TMPDIR=/tmp; cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/st_c -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-sample-mib 0 > $R/gpurun_out/c_random8g.json 2> $R/gpurun_out/c_stats.err || echo "stats failed"
cd $R; cp $(ls gpurun_out/st_c/*/*_kernel_stats.csv | head -1) gpurun_out/c_kernel_stats.csv; rm -rf gpurun_out/st_c
python tools/summ.py gpurun_out/c_random8g.json
cut -c1-150 gpurun_out/c_kernel_stats.csv | head -20

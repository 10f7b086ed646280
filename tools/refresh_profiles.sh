#!/bin/bash
# Regenerates everything under profiles/ for one version tag (run on the GPU box from the repo root).
set -o pipefail
V=$1; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_$V -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-sample-mib 0 > $R/gpurun_out/stats_$V.json 2> $R/gpurun_out/stats_$V.err || echo "stats failed"
cd $R
cp $(ls gpurun_out/stats_$V/*/*_kernel_stats.csv | head -1) gpurun_out/kernel_stats_$V.csv || echo "no stats csv"
bash tools/traffic.sh $V > gpurun_out/traffic_$V.txt 2>&1 && python3 tools/mk_traffic.py $V random8g 8589934592 1048576 "build $V" > gpurun_out/traffic_json_$V.txt
bash tools/pmc.sh $V > gpurun_out/pmc_sq_$V.txt 2>&1
timeout -k 10 600 python bench.py > gpurun_out/bench_random8g_$V.json 2> gpurun_out/bench_random8g_$V.err || echo "bench default failed"
for w in random256m text lowentropy; do timeout -k 10 300 python bench.py --workload $w --cpu-sample-mib 0 > gpurun_out/bench_${w}_$V.json 2>/dev/null || echo "bench $w failed"; done
timeout -k 10 300 python bench.py --workload text --bytes-per-gpu 8589934592 --cpu-sample-mib 0 > gpurun_out/bench_text8g_$V.json 2>/dev/null || echo "bench text8g failed"
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --backend gloo --single-device --bytes-per-gpu 2147483648 --cpu-sample-mib 0 > gpurun_out/bench_2rank_$V.json 2> gpurun_out/bench_2rank_$V.err || echo "2rank failed"
python tools/summ.py gpurun_out/bench_random8g_$V.json gpurun_out/bench_random256m_$V.json gpurun_out/bench_text_$V.json gpurun_out/bench_text8g_$V.json gpurun_out/bench_lowentropy_$V.json

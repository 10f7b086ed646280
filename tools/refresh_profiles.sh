#!/bin/bash
# Regenerates the round's evidence under gpurun_out/ for one tag (run on the GPU box from the repo root); copy what is to
# be judged into profiles/ afterwards (tools/collect_profiles.sh <tag> <round>).
#   per workload: rocprofv3 kernel stats, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes), SQ counters
#   bench lines for every workload (hipEvent kernel times, roofline, cpu_baseline on the default one), 2-rank gloo rehearsal
set -o pipefail
V=$1; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
# HBM traffic of the default workload first: bench.py quotes it as roofline.traffic when the shapes match
export TMPDIR=/tmp
( cd /tmp; for P in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/trf_${V}_$P -- python3 $R/bench.py --no-secondary --steps 3 --warmup 1 --cpu-sample-mib 0 --no-verify > $R/gpurun_out/trf_${V}_$P.json 2> $R/gpurun_out/trf_${V}_$P.err || echo "pass $P failed"
  done )
python3 tools/mk_traffic.py $V random8g 8589934592 1048576 "build $V" > gpurun_out/traffic_json_$V.txt 2>&1 || echo "mk_traffic failed"
rm -rf gpurun_out/trf_${V}_FETCH_SIZE gpurun_out/trf_${V}_WRITE_SIZE
bash tools/profile_workload.sh ${V}_random8g
bash tools/profile_workload.sh ${V}_text8g --workload text8g
bash tools/profile_workload.sh ${V}_lowentropy --workload lowentropy
bash tools/profile_workload.sh ${V}_text32m --workload text_32m
bash tools/profile_workload.sh ${V}_text256 --workload text
bash tools/profile_workload.sh ${V}_random256m --workload random256m
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_random8g_$V.json 2> gpurun_out/bench_random8g_$V.err || echo "bench default failed"

HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --no-secondary --gpus 2 --backend gloo --single-device --bytes-per-gpu 2147483648 --cpu-sample-mib 0 > gpurun_out/bench_2rank_$V.json 2> gpurun_out/bench_2rank_$V.err || echo "2rank failed"
python tools/summ.py gpurun_out/bench_random8g_$V.json

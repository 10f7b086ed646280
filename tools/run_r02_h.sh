#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for w in text lowentropy; do
for cb in 33554432 16777216; do
  timeout -k 10 300 python bench.py --workload $w --bytes-per-gpu 1073741824 --chunk-bytes $cb --cpu-sample-mib 0 --steps 3 > gpurun_out/h_${w}_${cb}_split.json 2> gpurun_out/h.err || echo "failed"
  DCZ_K4_SPLIT_BELOW=0 timeout -k 10 300 python bench.py --workload $w --bytes-per-gpu 1073741824 --chunk-bytes $cb --cpu-sample-mib 0 --steps 3 > gpurun_out/h_${w}_${cb}_nosplit.json 2> gpurun_out/h.err || echo "failed"
done; done
timeout -k 10 300 python bench.py --workload text --bytes-per-gpu 33554432 --chunk-bytes 33554432 --cpu-sample-mib 0 --steps 5 > gpurun_out/h_text_1x32m_split.json 2> gpurun_out/h.err || echo "failed"
DCZ_K4_SPLIT_BELOW=0 timeout -k 10 300 python bench.py --workload text --bytes-per-gpu 33554432 --chunk-bytes 33554432 --cpu-sample-mib 0 --steps 5 > gpurun_out/h_text_1x32m_nosplit.json 2> gpurun_out/h.err || echo "failed"
timeout -k 10 300 python bench.py --workload text --cpu-sample-mib 0 --steps 3 > gpurun_out/h_text_1g_4m.json 2> gpurun_out/h.err
python tools/summ.py gpurun_out/h_*.json

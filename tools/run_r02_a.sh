#!/bin/bash
# round 2, GPU call A: multi-symbol class on text (variants) + baseline rocprof evidence for text8g / lowentropy / random256m
R=$GRAFT_REPO_ROOT; cd $R
for v in base m11 m12 m13; do
  DCZ_LIB=$R/variants/lib_$v.so timeout -k 10 300 python bench.py --workload text --bytes-per-gpu 8589934592 --cpu-sample-mib 0 --steps 3 > gpurun_out/a_text8g_$v.json 2> gpurun_out/a_text8g_$v.err || echo "variant $v failed"
  DCZ_LIB=$R/variants/lib_$v.so timeout -k 10 300 python bench.py --workload text --cpu-sample-mib 0 --steps 3 > gpurun_out/a_text1g_$v.json 2> gpurun_out/a_text1g_$v.err || echo "variant $v 1g failed"
done
python tools/summ.py gpurun_out/a_text8g_*.json gpurun_out/a_text1g_*.json
bash tools/profile_workload.sh r02base_text8g --workload text --bytes-per-gpu 8589934592
bash tools/profile_workload.sh r02base_lowentropy --workload lowentropy
bash tools/profile_workload.sh r02base_random256m --workload random256m

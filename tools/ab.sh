#!/bin/bash
# A/B of the main build against variants/lib_<v>.so on the given workloads: gpu_r3_ab.sh "<workloads>" "<variants>"
R=$GRAFT_REPO_ROOT; cd $R
for w in $1; do
  timeout -k 10 300 python bench.py --no-secondary --workload $w --cpu-sample-mib 0 --steps 5 --warmup 2 > gpurun_out/ab_main_$w.json 2> gpurun_out/ab_main_$w.err || echo "main $w failed: $(tail -1 gpurun_out/ab_main_$w.err)"
  python tools/summ.py gpurun_out/ab_main_$w.json
  for v in $2; do
    DCZ_LIB=$R/variants/lib_$v.so timeout -k 10 300 python bench.py --no-secondary --workload $w --cpu-sample-mib 0 --steps 5 --warmup 2 > gpurun_out/ab_${v}_$w.json 2> gpurun_out/ab_${v}_$w.err || echo "$v $w failed: $(tail -1 gpurun_out/ab_${v}_$w.err)"
    python tools/summ.py gpurun_out/ab_${v}_$w.json
  done
done

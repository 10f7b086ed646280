#!/bin/bash
# A/B of environment knobs: ab_env.sh "<workloads>" "<VAR=val ...>" (each setting against the default, 20 steps)
cd $GRAFT_REPO_ROOT
for w in $1; do
  timeout -k 10 300 python bench.py --no-secondary --workload $w --cpu-sample-mib 0 --steps 20 --warmup 5 > gpurun_out/abe_${w}_default.json 2> gpurun_out/abe.err || echo "failed $w default"
  echo -n "default: "; python tools/summ.py gpurun_out/abe_${w}_default.json
  for kv in $2; do
    env $kv timeout -k 10 300 python bench.py --no-secondary --workload $w --cpu-sample-mib 0 --steps 20 --warmup 5 > gpurun_out/abe_${w}_${kv}.json 2> gpurun_out/abe.err || echo "failed $w $kv"
    echo -n "$kv: "; python tools/summ.py gpurun_out/abe_${w}_${kv}.json
  done
done

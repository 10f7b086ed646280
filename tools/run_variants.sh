#!/bin/bash
# usage: run_variants.sh <workload args ...> -- v1 v2 ...   : bench each variants/lib_<v>.so, print one summary line each
R=$GRAFT_REPO_ROOT; cd $R
ARGS=(); while [ "$1" != "--" ]; do ARGS+=("$1"); shift; done; shift
for v in "$@"; do
  DCZ_LIB=$R/variants/lib_$v.so timeout -k 10 300 python bench.py --no-secondary "${ARGS[@]}" --cpu-sample-mib 0 --steps 3 > gpurun_out/v_$v.json 2> gpurun_out/v_$v.err || echo "variant $v failed"
done
for v in "$@"; do python tools/summ.py gpurun_out/v_$v.json; done

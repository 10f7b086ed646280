#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
bash tools/run_variants.sh --workload text8g -- cur g0 g0p88 g0p92
for v in cur g0; do echo "== $v"; DCZ_LIB=$R/variants/lib_$v.so timeout -k 10 300 python tools/bench_dist.py 2>&1 | grep -v amdgpu; done

#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
DCZ_LIB=$R/variants/lib_m11prof.so timeout -k 10 300 python tools/k4prof.py > gpurun_out/k4prof_m11.txt 2>&1; grep -v Warning gpurun_out/k4prof_m11.txt | tail -6
export DCZ_LIB=$R/variants/lib_m11.so
bash tools/profile_workload.sh r02m11_text8g --workload text --bytes-per-gpu 8589934592
grep -A2 "true, 1, 11, 0" gpurun_out/prof_r02m11_text8g_sq.txt
